#!/bin/bash
# round 4, GPU call 4: sub-batching of the last two decoder levels (Infinity-Cache residency of dx / V between launches)
export TMPDIR=/tmp
O=gpurun_out/r04d
mkdir -p $O
timeout 300 python -m pytest tests -m gpu -x -q -k "structure_switches" > $O/pytest_sub.log 2>&1; tail -2 $O/pytest_sub.log
for sub in 0 1 2 4 8; do
  I2V_DEC_SUB=$sub timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 > $O/bench_bair64_sub$sub.json
  I2V_DEC_SUB=$sub timeout 300 python bench.py --config land128 --steps 10 --warmup 3 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 > $O/bench_land128_sub$sub.json
done
python3 - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r04d/bench_*.json")):
    try:
        d = json.load(open(f)); print(f.split("/")[-1], round(d["ms_per_step"], 3), "ms", round(d["roofline"]["ms_per_step"], 2), "dom ms")
    except Exception as e: print(f, "ERR", e)
PY
