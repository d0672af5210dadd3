#!/bin/bash
# round 6, GPU call 23: direct conv with the operand reads interleaved between the MFMAs (vs the burst form), ablations of the stage loop (fixed build),
# the decoder tests, the pipeline probe with alternating prefetch priorities
export TMPDIR=/tmp
O=gpurun_out/r06_23; mkdir -p $O
{
for rep in 1 2; do
for shape in "64 2 8 8 1024 1024 0 1 20" "64 2 8 8 1024 1024 1 1 20" "8 2 8 8 1024 1024 0 1 40" "8 2 8 8 1024 1024 1 1 40" "64 1 4 4 1024 1024 0 1 40" "8 1 64 64 128 256 0 0 20" "8 1 16 16 128 256 0 0 40" "3 2 8 8 256 256 0 1 40"; do
  echo "-- $shape"
  for b in conv16_bench_i0 conv16_bench; do echo -n "$b  "; tools/$b $shape; done
done
done
for shape in "64 2 8 8 1024 1024 1 1 20" "64 2 8 8 1024 1024 0 1 20"; do
  echo "-- ablations $shape"
  for b in conv16_bench conv16_bench_abl1 conv16_bench_abl2 conv16_bench_abl4 conv16_bench_abl8 conv16_bench_abl6 conv16_bench_abl14; do echo -n "$b  "; tools/$b $shape; done
done
} > $O/conv16_interleave_ab.txt 2>&1
cut -c1-170 $O/conv16_interleave_ab.txt
timeout 1200 python -m pytest tests -m gpu -x -q -k "decoder or dec_ or model or shard or block or embed or encoder or determinism" > $O/tests_decoder.txt 2>&1; tail -3 $O/tests_decoder.txt
timeout 600 python tools/pipeline_probe.py > $O/pipeline_probe.txt 2>&1; cat $O/pipeline_probe.txt | cut -c1-330
timeout 300 python bench.py --steps 20 --warmup 3 --lean --per-layer $O/per_layer_bair64.csv 2>/dev/null | tail -1 > $O/bench_bair64.json
timeout 300 python bench.py --batch 8 --steps 40 --warmup 5 --lean --per-layer $O/per_layer_bair8.csv 2>/dev/null | tail -1 > $O/bench_bair8.json
timeout 300 python bench.py --config land128 --steps 20 --warmup 3 --lean 2>/dev/null | tail -1 > $O/bench_land128.json
python - <<'PY'
import json
for f in ('bench_bair64','bench_bair8','bench_land128'):
    r=json.load(open(f'gpurun_out/r06_23/{f}.json')); print(f, r['ms_per_step'], (r.get('single_call') or {}).get('ms'))
PY
head -5 $O/per_layer_bair64.csv $O/per_layer_bair8.csv
