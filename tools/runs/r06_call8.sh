#!/bin/bash
# round 6, GPU call 8: conversion-rounding probe, the whole GPU suite on the current build, I2V_DEC_GEN=0/1/2 on Landscape, per-layer tables
# after the split-K-aware channel-tile choice of the direct kernel (head_0)
export TMPDIR=/tmp
O=gpurun_out/r06_8; mkdir -p $O
timeout 60 tools/cvt_tie_test > $O/cvt_tie_test.txt 2>&1; cat $O/cvt_tie_test.txt
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -8 > $O/gpu_tests.txt; cat $O/gpu_tests.txt
for g in 0 2 1 0 2; do
  I2V_DEC_GEN=$g timeout 300 python bench.py --config land128 --steps 20 --warmup 3 --lean 2>/dev/null | tail -1 > $O/bench_land128_gen$g.$RANDOM.json
done
timeout 300 python bench.py --steps 20 --warmup 3 --lean --per-layer $O/per_layer_bair64.csv 2>/dev/null | tail -1 > $O/bench_bair64.json
timeout 300 python bench.py --batch 8 --steps 40 --warmup 5 --lean --per-layer $O/per_layer_bair8.csv 2>/dev/null | tail -1 > $O/bench_bair8.json
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r06_8/bench_*.json')):
    try:
        r=json.load(open(f)); print(f.split('/')[-1], 'ms/step %.3f'%r['ms_per_step'], 'single', (r.get('single_call') or {}).get('ms'))
    except Exception as e: print(f, 'ERR', e)
PY
head -5 $O/per_layer_bair64.csv; head -5 $O/per_layer_bair8.csv
