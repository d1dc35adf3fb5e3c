#!/bin/bash
# Shader clock / package power of the GPU while the graph-replayed bench runs (rocm-smi sampled every ~0.5 s): the MFMA peak the bench
# line divides by assumes 2.4 GHz; what the firmware grants under this workload is recorded in profiles/r3_clock_power.json.
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r3clk; mkdir -p $O
python $R/bench.py --steps 20 --warmup 1 --no-cpu-baseline --no-roofline --no-other-configs > $O/bench.log 2>&1 &
BP=$!
sleep 14          # model build + warm-up story
: > $O/samples.txt
while kill -0 $BP 2>/dev/null; do
  /opt/rocm/bin/rocm-smi --showclocks --showpower --json 2>/dev/null >> $O/samples.txt; echo >> $O/samples.txt
  sleep 0.4
done
python - <<PY
import json, re, statistics
sclk, pwr = [], []
for line in open("$O/samples.txt"):
    line = line.strip()
    if not line.startswith("{"): continue
    try: d = json.loads(line)
    except ValueError: continue
    c = d.get("card0", {})
    for k, v in c.items():
        if "sclk" in k.lower():
            m = re.search(r"(\d+)\s*Mhz", str(v), re.I)
            if m: sclk.append(int(m.group(1)))
        if "power" in k.lower() and "socket" in k.lower() or "Average Graphics Package Power" in k or "Current Socket Graphics Package Power" in k:
            try: pwr.append(float(v))
            except ValueError: pass
val = None
for l in open("$O/bench.log"):
    if l.startswith("{"): val = json.loads(l)["value"]
out = {"command": "bench.py --steps 20 --warmup 1 (graph replay), rocm-smi --showclocks --showpower sampled every ~0.5 s after the warm-up story",
       "samples": len(sclk), "images_per_s": val,
       "sclk_mhz": {"min": min(sclk) if sclk else None, "median": statistics.median(sclk) if sclk else None, "max": max(sclk) if sclk else None},
       "package_power_w": {"min": min(pwr) if pwr else None, "median": statistics.median(pwr) if pwr else None, "max": max(pwr) if pwr else None}}
json.dump(out, open("$O/r3_clock_power.json", "w"), indent=1)
print(json.dumps(out)); print(open("$O/samples.txt").read()[:1500])
PY
