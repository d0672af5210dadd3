#!/bin/bash
# round 4, GPU call 13: start skew of the persistent F(4,3) workgroups
export TMPDIR=/tmp
O=gpurun_out/r04m
mkdir -p $O
for cfg in "0 0" "2 0" "2 2" "2 4" "2 8"; do
  set -- $cfg
  I2V_W4_PIPE=$1 I2V_W4_SKEW=$2 timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 > $O/bench_bair64_p$1_s$2.json
  I2V_W4_PIPE=$1 I2V_W4_SKEW=$2 timeout 300 python bench.py --config land128 --steps 10 --warmup 3 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 > $O/bench_land128_p$1_s$2.json
done
python3 - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r04m/bench_*.json")):
    d = json.load(open(f)); print(f.split("/")[-1], round(d["ms_per_step"], 3), "ms; F(4,3)", round(d["roofline"]["ms_per_step"], 2))
PY
