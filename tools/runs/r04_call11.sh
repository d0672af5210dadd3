#!/bin/bash
# round 4, GPU call 11: i2v_dec_prepare (SPADE branches ahead of the forward, underneath the cINN pass): tests + single-call latency
export TMPDIR=/tmp
O=gpurun_out/r04k
mkdir -p $O
timeout 600 python -m pytest tests -m gpu -x -q -k "prepare or model_ or generate_ or sample_prior or pipelined or smoke" 2>&1 | tail -3
for b in 64 8; do
  timeout 300 python bench.py --batch $b --steps 10 --warmup 3 --no-cpu-baseline --sustain 0 --no-exact 2>/dev/null | tail -1 > $O/bench_bair64_b$b.json
  timeout 300 python bench.py --batch $b --steps 10 --warmup 3 --no-cpu-baseline --sustain 0 --no-exact --pipeline 0 2>/dev/null | tail -1 > $O/bench_bair64_b${b}_serial.json
done
timeout 300 python bench.py --config land128 --steps 10 --warmup 3 --no-cpu-baseline --sustain 0 --no-exact 2>/dev/null | tail -1 > $O/bench_land128.json
python3 - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r04k/bench_*.json")):
    d = json.load(open(f)); print(f.split("/")[-1], "step", round(d["ms_per_step"], 3), "single call", round(d["single_call"]["ms"], 3))
PY
