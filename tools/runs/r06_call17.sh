#!/bin/bash
# round 6, GPU call 17: direct split-fp16 conv on the two-frame / one-frame maps (g_0): clipped staging (no temporal halo frames, conflict-free
# row pitch, masked operand reads) vs the previous staging, same binary; one-rank RCCL probe
export TMPDIR=/tmp
O=gpurun_out/r06_17; mkdir -p $O
{
for rep in 1 2; do
for clip in 0 1; do
  I2V_C16_CLIP=$clip tools/conv16_bench 64 2 8 8 1024 1024 0 1 20
  I2V_C16_CLIP=$clip tools/conv16_bench 64 2 8 8 1024 1024 1 1 20
  I2V_C16_CLIP=$clip tools/conv16_bench 8 2 8 8 1024 1024 0 1 40
  I2V_C16_CLIP=$clip tools/conv16_bench 8 2 8 8 1024 1024 1 1 40
  I2V_C16_CLIP=$clip tools/conv16_bench 16 2 8 8 1024 1024 0 1 40
  I2V_C16_CLIP=$clip tools/conv16_bench 3 2 8 8 256 256 0 1 40
  I2V_C16_CLIP=$clip tools/conv16_bench 3 2 8 8 256 256 1 1 40
  I2V_C16_CLIP=$clip tools/conv16_bench 5 2 8 8 128 128 0 0 40
done
done
} > $O/conv16_clip_ab.txt 2>&1
cat $O/conv16_clip_ab.txt
timeout 300 python tools/rccl_one_rank_probe.py > $O/rccl_one_rank_probe.txt 2>&1; echo "probe rc $?" >> $O/rccl_one_rank_probe.txt; tail -12 $O/rccl_one_rank_probe.txt
timeout 900 python -m pytest tests -m gpu -x -q -k "decoder or dec_ or model or shard or block" > $O/tests_decoder.txt 2>&1; tail -3 $O/tests_decoder.txt
timeout 300 python bench.py --steps 20 --warmup 3 --lean --per-layer $O/per_layer_bair64.csv 2>/dev/null | tail -1 > $O/bench_bair64.json
timeout 300 python bench.py --batch 8 --steps 40 --warmup 5 --lean --per-layer $O/per_layer_bair8.csv 2>/dev/null | tail -1 > $O/bench_bair8.json
python - <<'PY'
import json
for f in ('bench_bair64','bench_bair8'):
    r=json.load(open(f'gpurun_out/r06_17/{f}.json')); print(f, r['ms_per_step'], (r.get('single_call') or {}).get('ms'))
PY
head -5 $O/per_layer_bair64.csv $O/per_layer_bair8.csv
