#!/bin/bash
# round 6, GPU call 1: the GPU suite on the build with the capture-aware ordering / side-stream join, then same-box baselines
export TMPDIR=/tmp
O=gpurun_out/r06_1; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > $O/gpu_tests.txt
timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-live-traffic --sustain 0 --no-exact --per-layer $O/per_layer_bair64.csv 2>/dev/null | tail -1 > $O/bench_bair64.json
timeout 300 python bench.py --config land128 --steps 20 --warmup 3 --no-cpu-baseline --no-live-traffic --sustain 0 --no-exact --per-layer $O/per_layer_land128.csv 2>/dev/null | tail -1 > $O/bench_land128.json
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_land -o bench -- python bench.py --config land128 --steps 3 --warmup 1 --no-cpu-baseline --no-extras > $O/prof_land.log 2>&1
rm -f $O/prof_*/*/bench_kernel_trace.csv $O/prof_*/bench_kernel_trace.csv
cat $O/gpu_tests.txt
