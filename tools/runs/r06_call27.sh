#!/bin/bash
# round 6, GPU call 27: final evidence of the round on ONE box with the final build (tools/collect_evidence.sh) + the N > 1 stream configuration legs
export TMPDIR=/tmp
O=gpurun_out/r06_z; mkdir -p $O
bash tools/collect_evidence.sh $O > $O/collect.log 2>&1
timeout 300 python bench.py --batch 8 --steps 40 --warmup 5 --lean --emulate-collation rccl 2>/dev/null | grep '^{' | tail -1 > $O/bench_bair8_multi_gpu_streams_rccl.json
timeout 300 python bench.py --steps 20 --warmup 3 --lean --emulate-collation rccl 2>/dev/null | grep '^{' | tail -1 > $O/bench_bair64_multi_gpu_streams_rccl.json
timeout 300 python tools/pipeline_probe.py 0 -1 > $O/pipeline_probe.txt 2>&1
python tools/results_table.py $O 2>/dev/null | head -60
ls $O
