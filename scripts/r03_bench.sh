#!/bin/bash
# the driver's round-end sequence: smoke(), then the bench command as the driver runs it
set -u
mkdir -p gpurun_out
timeout -k 10 300 python3 -u -c "import __graft_entry__ as g; g.smoke()" || { echo "smoke failed"; exit 1; }
timeout -k 10 900 python3 -u bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r03_bench.json 2> gpurun_out/r03_bench.err || { echo "bench failed"; tail -20 gpurun_out/r03_bench.err; exit 1; }
grep "timed step\|lavapipe" gpurun_out/r03_bench.err | tail -12
python3 - <<'PY'
import json
d = json.load(open("gpurun_out/r03_bench.json"))
def show(c):
    r = c.get("roofline", {})
    print(c.get("name", c["config"]["name"]), f"{c['value']:.0f} Mrays/s {c['ms_per_step']:.1f} ms", "verified", c.get("frame_verified"), "props", c.get("frame_properties_ok"),
          "band", c.get("cpu_baseline", {}).get("gpu_band_bit_exact"), "roof", r.get("bound"), f"{r.get('frac', 0):.3f}", r.get("kernel"), "traffic", r.get("traffic"), "cpu", f"{c.get('cpu_baseline', {}).get('value', 0):.1f}")
show(d)
for c in d.get("configs", []):
    if "skipped" in c: print(c)
    else: show(c)
for k in ("fast_arith", "bvh_wide4"):
    print(k, {a: b for a, b in d.get(k, {}).items() if a != "note"})
print("cpu_baseline", {a: b for a, b in d["cpu_baseline"].items() if a != "sample"})
PY
