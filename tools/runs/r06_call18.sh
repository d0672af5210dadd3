#!/bin/bash
# round 6, GPU call 18: direct split-fp16 conv: weight pipeline over virtual stages (prefetch across chunk boundaries) vs per chunk (x0 binary),
# TSK wave permutation on / off, clipped staging on / off; checksums must agree everywhere
export TMPDIR=/tmp
O=gpurun_out/r06_18; mkdir -p $O
{
for rep in 1 2; do
  for shape in "64 2 8 8 1024 1024 0 1 20" "64 2 8 8 1024 1024 1 1 20" "8 2 8 8 1024 1024 0 1 40" "8 2 8 8 1024 1024 1 1 40" "64 1 4 4 1024 1024 0 1 40" "8 1 4 4 1024 1024 0 1 40" "8 1 64 64 128 256 0 0 20" "3 2 8 8 256 256 0 1 40" "2 4 16 16 64 64 0 0 40"; do
    echo "-- $shape"
    tools/conv16_bench_x0 $shape
    tools/conv16_bench $shape
    I2V_C16_PERM=0 tools/conv16_bench $shape
    I2V_C16_CLIP=0 tools/conv16_bench $shape
  done
done
} > $O/conv16_xchunk_ab.txt 2>&1
cat $O/conv16_xchunk_ab.txt | cut -c1-200
timeout 1200 python -m pytest tests -m gpu -x -q -k "decoder or dec_ or model or shard or block or embed or encoder or determinism" > $O/tests_decoder.txt 2>&1; tail -3 $O/tests_decoder.txt
timeout 300 python bench.py --steps 20 --warmup 3 --lean --per-layer $O/per_layer_bair64.csv 2>/dev/null | tail -1 > $O/bench_bair64.json
timeout 300 python bench.py --batch 8 --steps 40 --warmup 5 --lean --per-layer $O/per_layer_bair8.csv 2>/dev/null | tail -1 > $O/bench_bair8.json
timeout 300 python bench.py --config land128 --steps 20 --warmup 3 --lean 2>/dev/null | tail -1 > $O/bench_land128.json
python - <<'PY'
import json
for f in ('bench_bair64','bench_bair8','bench_land128'):
    r=json.load(open(f'gpurun_out/r06_18/{f}.json')); print(f, r['ms_per_step'], (r.get('single_call') or {}).get('ms'))
PY
head -5 $O/per_layer_bair64.csv $O/per_layer_bair8.csv
