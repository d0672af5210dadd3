#!/bin/bash
# round 4, GPU call 10: cache-policy hints (non-temporal V stream / output stores / operand writer), A/B through alternate builds
export TMPDIR=/tmp
O=gpurun_out/r04j
mkdir -p $O
for v in base vnt outnt modnt allnt; do
  L=""; [ $v != base ] && L=tools/_tl/libi2v_hip_$v.so
  I2V_LIB_PATH=$L timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 > $O/bench_bair64_$v.json
  I2V_LIB_PATH=$L timeout 300 python bench.py --config land128 --steps 10 --warmup 3 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 > $O/bench_land128_$v.json
done
I2V_LIB_PATH=tools/_tl/libi2v_hip_allnt.so timeout 300 python -m pytest tests -m gpu -x -q -k "nf8_bair or full_width_bair or full_width_128" 2>&1 | tail -2
python3 - <<'PY'
import json
for v in ("base", "vnt", "outnt", "modnt", "allnt"):
    for f in ("bench_bair64", "bench_land128"):
        try:
            d = json.load(open(f"gpurun_out/r04j/{f}_{v}.json")); print(f, v, round(d["ms_per_step"], 3), "ms; F(4,3)", round(d["roofline"]["ms_per_step"], 2))
        except Exception as e: print(f, v, "ERR", e)
PY
