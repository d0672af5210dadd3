#!/bin/bash
# round 6, GPU call 16: the collation of an N > 1 rank through a ONE-rank RCCL process group (torch.distributed / RCCL issue the gather) vs the device-copy stand-in;
# the tests the stream change touches
export TMPDIR=/tmp
O=gpurun_out/r06_16; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q -k "side_stream or overlap or collat or shard or capture or pipeline" > $O/tests_streams.txt 2>&1; tail -3 $O/tests_streams.txt
for k in 1 2; do
  timeout 300 python bench.py --batch 8 --steps 40 --warmup 5 --lean --emulate-collation rccl 2>$O/err_rccl8.$k.txt | tail -1 > $O/bench_bair8_rccl1_on_prefetch.$k.json
  timeout 300 python bench.py --batch 8 --steps 40 --warmup 5 --lean --emulate-collation copy 2>/dev/null | tail -1 > $O/bench_bair8_copy_on_prefetch.$k.json
  timeout 300 python bench.py --batch 8 --steps 40 --warmup 5 --lean 2>/dev/null | tail -1 > $O/bench_bair8_one_gpu.$k.json
  timeout 300 python bench.py --batch 8 --steps 40 --warmup 5 --lean --emulate-collation rccl --collation-stream own 2>/dev/null | tail -1 > $O/bench_bair8_rccl1_own_side_shared.$k.json
  timeout 300 python bench.py --steps 20 --warmup 3 --lean --emulate-collation rccl 2>/dev/null | tail -1 > $O/bench_bair64_rccl1_on_prefetch.$k.json
  timeout 300 python bench.py --steps 20 --warmup 3 --lean 2>/dev/null | tail -1 > $O/bench_bair64_one_gpu.$k.json
done
timeout 300 python bench.py --batch 8 --steps 40 --warmup 5 --lean --emulate-collation rccl --collation-stream own --side-stream own 2>/dev/null | tail -1 > $O/bench_bair8_rccl1_own_side_own.json
timeout 300 python bench.py --config land128 --steps 20 --warmup 3 --lean --emulate-collation rccl 2>/dev/null | tail -1 > $O/bench_land128_rccl1_on_prefetch.json
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r06_16/bench_*.json')):
    try:
        r=json.load(open(f))
        print(f.split('/')[-1], 'ms/step %.3f'%r['ms_per_step'], 'single %.3f'%(r.get('single_call') or {}).get('ms'), r['streams']['side_stream'], '|', r['streams'].get('collation_stream'), '|', r['streams'].get('collation_stream_emulated'))
    except Exception as e: print(f, 'ERR', e)
PY
head -5 $O/err_rccl8.1.txt
