#!/bin/bash
# round 6, GPU call 29: every N > 1 code path of bench.py on one GPU (one-rank RCCL group), plain and under torch.distributed.run
export TMPDIR=/tmp
O=gpurun_out/r06_29; mkdir -p $O
I2V_BENCH_FORCE_MULTI=1 timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/forced_multi.out 2> $O/forced_multi.err; echo "rc $?" >> $O/forced_multi.err
tail -1 $O/forced_multi.out | cut -c1-400; tail -5 $O/forced_multi.err
I2V_BENCH_FORCE_MULTI=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29731 bench.py --gpus 1 --steps 5 --warmup 2 > $O/forced_multi_torchrun.out 2> $O/forced_multi_torchrun.err; echo "rc $?" >> $O/forced_multi_torchrun.err
tail -1 $O/forced_multi_torchrun.out | cut -c1-300; tail -5 $O/forced_multi_torchrun.err
python - <<'PY'
import json
for f in ('forced_multi','forced_multi_torchrun'):
    l=open(f'gpurun_out/r06_29/{f}.out').read().strip().splitlines()
    r=json.loads(l[-1]); print(f, 'last line is the JSON line; lines on stdout:', len(l), '| ms/step', r['ms_per_step'], 'ranks_seen', r['ranks_seen'], 'rccl', r['rccl_version'], 'rank_ms', r['rank_ms_per_step'], r['streams']['collation_stream'], [k for k in ('small_batch','config_128','cpu_baseline','sustained') if k in r])
PY
