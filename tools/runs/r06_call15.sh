#!/bin/bash
# round 6, GPU call 15: the all-gather issued on the cINN prefetch stream (three streams: main + decoder side + prefetch/collation) vs a collation stream of its own
export TMPDIR=/tmp
O=gpurun_out/r06_15; mkdir -p $O
for k in 1 2; do
  timeout 300 python bench.py --batch 8 --steps 40 --warmup 5 --lean --emulate-collation 2>/dev/null | tail -1 > $O/bench_bair8_coll_on_prefetch.$k.json
  timeout 300 python bench.py --steps 20 --warmup 3 --lean --emulate-collation 2>/dev/null | tail -1 > $O/bench_bair64_coll_on_prefetch.$k.json
  timeout 300 python bench.py --batch 8 --steps 40 --warmup 5 --lean --emulate-collation --collation-stream own 2>/dev/null | tail -1 > $O/bench_bair8_coll_own_side_shared.$k.json
  timeout 300 python bench.py --steps 20 --warmup 3 --lean --emulate-collation --collation-stream own 2>/dev/null | tail -1 > $O/bench_bair64_coll_own_side_shared.$k.json
  timeout 300 python bench.py --batch 8 --steps 40 --warmup 5 --lean 2>/dev/null | tail -1 > $O/bench_bair8_one_gpu.$k.json
  timeout 300 python bench.py --steps 20 --warmup 3 --lean 2>/dev/null | tail -1 > $O/bench_bair64_one_gpu.$k.json
done
timeout 300 python bench.py --batch 8 --steps 40 --warmup 5 --lean --emulate-collation --collation-stream own --side-stream own 2>/dev/null | tail -1 > $O/bench_bair8_coll_own_side_own.json
timeout 300 python bench.py --config land128 --steps 20 --warmup 3 --lean --emulate-collation 2>/dev/null | tail -1 > $O/bench_land128_coll_on_prefetch.json
timeout 300 python bench.py --config land128 --steps 20 --warmup 3 --lean 2>/dev/null | tail -1 > $O/bench_land128_one_gpu.json
( time timeout 900 python bench.py 2>/dev/null | tail -1 > $O/bench_default.json ) 2> $O/bench_default.time
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r06_15/bench_*.json')):
    try:
        r=json.load(open(f)); sb=r.get('small_batch') or {}
        print(f.split('/')[-1], 'ms/step %.3f'%r['ms_per_step'], 'single', (r.get('single_call') or {}).get('ms'), r['streams']['side_stream'], r['streams'].get('collation_stream'))
        if sb: print('   n1', sb['n1'], '\n   multi', sb['multi_gpu'], '\n   proj', {k:v for k,v in sb['projected_strong_scaling'].items() if k in ('single_call','single_call_before_collation','pipelined')})
    except Exception as e: print(f, 'ERR', e)
PY
cat $O/bench_default.time
