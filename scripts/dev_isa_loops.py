#!/usr/bin/env python
"""dev helper (not part of the product): what the register allocator did to a kernel's MFMA loop.

    python scripts/dev_isa_loops.py theatergen_amd/csrc/tg_conv_slab_pp.hip [substring of the kernel symbol] [-D...]

Compiles the file to gfx950 assembly (hipcc -S --cuda-device-only) and reports, per kernel with MFMAs: the instructions between the first inner-loop header in front of the first MFMA
and the last MFMA of that loop, the scratch instructions among them (dword / wider = an accumulator or fragment tuple bouncing through scratch), and the MFMAs whose destination tuple
differs from their accumulator input (the allocator renaming accumulators: harmless by itself, the symptom of a fragmented file).  Round 6: the two-wave slab conv carried six
`scratch_load_dword; s_waitcnt vmcnt(0)` per chunk (or, one unrelated edit later, a whole accumulator tile) inside its K loop; profiles/r6_slab_findings.md."""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    src = sys.argv[1]
    filt = sys.argv[2] if len(sys.argv) > 2 and not sys.argv[2].startswith("-") else ""
    extra = [a for a in sys.argv[2:] if a.startswith("-")]
    out = os.path.join(tempfile.mkdtemp(prefix="isa_"), "k.s")
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-result", "-I" + os.path.join(ROOT, "include"),
           "-S", "--cuda-device-only", *extra, src, "-o", out]
    subprocess.run(cmd, check=True, stderr=subprocess.DEVNULL)
    lines = open(out).read().split("\n")
    starts = [i for i, l in enumerate(lines) if re.match(r"^_Z\w+:", l)]
    for s in starts:
        name = lines[s].split(":")[0]
        if filt not in name:
            continue
        e = next((j for j in range(s, len(lines)) if "s_endpgm" in lines[j]), len(lines))
        body = lines[s:e]
        mf = [j for j, l in enumerate(body) if "v_mfma" in l]
        if not mf:
            continue
        hdr = [j for j, l in enumerate(body) if "Loop Header" in l]
        h0 = max([h for h in hdr if h < mf[0]] or [0])
        h1 = min([h for h in hdr if h > mf[0]] or [len(body)])
        last = max(j for j in mf if j < h1)
        loop = body[h0:last + 1]
        sc = [l for l in loop if "scratch_" in l]
        wide = [l for l in sc if "dwordx" in l]
        ren = 0
        for j in mf:
            ops = [o.strip() for o in body[j].split(";")[0].split(None, 1)[1].split(",")]
            if len(ops) >= 4 and ops[3] != "0" and ops[0] != ops[3]:
                ren += 1
        # accumulator registers = destinations of the loop's MFMAs; any OTHER instruction of the loop that names one of them (a copy at the back edge, a spill) is
        # a hazard when the MFMAs are inline asm (no wait states inserted), and a slow loop either way
        def regs(tok):
            m = re.match(r"v\[(\d+):(\d+)\]", tok)
            if m:
                return set(range(int(m.group(1)), int(m.group(2)) + 1))
            m = re.match(r"v(\d+)$", tok)
            return {int(m.group(1))} if m else set()
        accs = set()
        for l in loop:
            if "v_mfma" in l:
                accs |= regs(l.split(";")[0].split(None, 1)[1].split(",")[0].strip())
        touched = 0
        for l in loop:
            t = l.split(";")[0].strip()
            if not t or t.startswith((".", ";")) or "v_mfma" in t or ":" in t.split()[0]:
                continue
            ops = re.findall(r"v\[\d+:\d+\]|v\d+", t)
            if any(regs(o) & accs for o in ops):
                touched += 1
        ss = next((l.split(":")[1].strip() for l in lines[e:e + 400] if "ScratchSize" in l), "?")
        print(f"{name[:100]:100s} loop {len(loop):5d} lines, mfma {len(mf):4d} (renamed {ren:3d}), scratch in loop {len(sc):3d} (wide {len(wide):2d}), accumulators touched outside MFMAs {touched:3d}, kernel scratch {ss} B")


if __name__ == "__main__":
    main()
