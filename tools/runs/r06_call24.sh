#!/bin/bash
# round 6, GPU call 24: order of the side stream's work -- SPADE branches of the late levels enqueued lazily (I2V_DEC_LAZY=3, default) vs all at the fork (6)
export TMPDIR=/tmp
O=gpurun_out/r06_24; mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -x -q -k "overlap or side_stream or prepare or capture or lifetime or collation or decoder_vs or shard" > $O/tests_streams.txt 2>&1; tail -3 $O/tests_streams.txt
for rep in 1 2; do
  for lz in 6 3 2 4; do
    I2V_DEC_LAZY=$lz timeout 300 python bench.py --steps 20 --warmup 3 --lean 2>/dev/null | tail -1 > $O/bench_bair64_lazy$lz.$rep.json
    I2V_DEC_LAZY=$lz timeout 300 python bench.py --config land128 --steps 20 --warmup 3 --lean 2>/dev/null | tail -1 > $O/bench_land128_lazy$lz.$rep.json
    I2V_DEC_LAZY=$lz timeout 300 python bench.py --batch 8 --steps 40 --warmup 5 --lean 2>/dev/null | tail -1 > $O/bench_bair8_lazy$lz.$rep.json
  done
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r06_24/bench_*.json')):
    try:
        r=json.load(open(f)); print(f.split('/')[-1], 'ms/step %.3f'%r['ms_per_step'], 'single %.3f'%(r.get('single_call') or {}).get('ms'))
    except Exception as e: print(f,'ERR',e)
PY
