#!/usr/bin/env python3
"""Where do a kernel's scratch (spill) instructions sit?  Compiles the given csrc/*.hip files to gfx950 assembly (device only, the library's own flags) and reports,
per kernel, the scratch loads / stores, FLAT accesses and MFMAs INSIDE loop blocks that contain MFMAs (the K loops) against the whole kernel.  The build-time
resource report (profiles/r4_kernel_resources.json) gives the totals; this says whether any of them is paid per K step.
usage: python scripts/asm_loop_report.py [out.json] [file.hip ...]   (default: every hot file)"""
import collections, json, os, re, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from theatergen_amd import build as tg_build  # noqa: E402

HOT = ["tg_conv_slab.hip", "tg_conv_halo.hip", "tg_gemm.hip", "tg_gemm_ln.hip", "tg_gemm_bt.hip", "tg_attention.hip", "tg_rowchain.hip"]


def demangle(names):
    out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.split("\n")
    return dict(zip(names, out))


def analyse(path):
    lines = open(path).read().split("\n")
    starts = [(i, l.split(":")[0]) for i, l in enumerate(lines) if re.match(r"^_Z\w+:\s+; @", l)]
    res = {}
    for i, name in starts:
        end = next(j for j in range(i, len(lines)) if lines[j].strip().startswith(".Lfunc_end"))
        blocks, cur = collections.OrderedDict(), ("entry", False)
        for l in lines[i:end]:
            t = l.strip()
            if t.startswith(".LBB"):
                cur = (t.split(":")[0], "Loop" in t)
            elif t and not t.startswith(";") and not t.startswith("."):
                blocks.setdefault(cur, []).append(t.split()[0])
        tot = collections.Counter()
        inl = collections.Counter()
        for (lbl, loop), ins in blocks.items():
            c = collections.Counter()
            for op in ins:
                if op.startswith("scratch_load"): c["scratch_load"] += 1
                elif op.startswith("scratch_store"): c["scratch_store"] += 1
                elif op.startswith("flat_"): c["flat"] += 1
                elif op.startswith("v_mfma"): c["mfma"] += 1
            tot.update(c)
            if loop and c["mfma"] > 0:
                inl.update(c)
        if tot["mfma"] == 0:
            continue
        res[name] = dict(total=dict(tot), in_mfma_loops=dict(inl))
    return res


def main():
    out = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "profiles", "r4_scratch_in_mfma_loops.json")
    files = sys.argv[2:] or HOT
    report = {}
    with tempfile.TemporaryDirectory() as td:
        for f in files:
            src = os.path.join(ROOT, "theatergen_amd", "csrc", f)
            asm = os.path.join(td, f + ".s")
            cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-result", *tg_build.EXTRA_FLAGS.get(f, []),
                   "-S", "--cuda-device-only", "-I", os.path.join(ROOT, "include"), "-I", os.path.join(ROOT, "theatergen_amd", "csrc"), src, "-o", asm]
            subprocess.run(cmd, check=True, capture_output=True)
            r = analyse(asm)
            dm = demangle(list(r))
            for k, v in r.items():
                name = re.sub(r"\(anonymous namespace\)::", "", dm.get(k, k)).split("(")[0].replace("void ", "")
                if "IDF16_" in k:
                    continue            # the fp16 twins mirror the bf16 instances (c++filt here does not know DF16b / DF16_: names stay mangled)
                report[name] = dict(file=f, **v)
    worst = {k: v for k, v in report.items() if v["in_mfma_loops"].get("scratch_load", 0) + v["in_mfma_loops"].get("scratch_store", 0) > 0}
    json.dump(dict(note="scratch / FLAT instructions inside loop blocks that contain MFMAs vs. the whole kernel (static counts from hipcc -S, gfx950)",
                   kernels_with_scratch_in_mfma_loops=sorted(worst), kernels=report), open(out, "w"), indent=1)
    print(len(report), "kernels;", len(worst), "with scratch inside an MFMA loop:", sorted(worst)[:8])


if __name__ == "__main__":
    main()
