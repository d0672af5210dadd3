#!/bin/bash
# round 4, GPU call 3: cINN chain A/B (kernarg preload, device kernargs, fp16-operand mode), full GPU suite with the new defaults
# (g_1 on F(4,3), brick order 2, PIPE off), bench lines
export TMPDIR=/tmp
O=gpurun_out/r04c
mkdir -p $O
for rep in 1 2; do
  FLOWTIME_B=64,8 timeout 200 python tools/flowtime.py >> $O/flowtime_ab.txt 2>&1
  FLOWTIME_B=64,8 FLOWTIME_LIB=tools/_tl/libi2v_hip_nopreload.so timeout 200 python tools/flowtime.py >> $O/flowtime_ab.txt 2>&1
done
FLOWTIME_B=64,8 HIP_FORCE_DEV_KERNARG=1 timeout 200 python tools/flowtime.py >> $O/flowtime_ab.txt 2>&1
FLOWTIME_B=64,8 HIP_FORCE_DEV_KERNARG=1 FLOWTIME_LIB=tools/_tl/libi2v_hip_nopreload.so timeout 200 python tools/flowtime.py >> $O/flowtime_ab.txt 2>&1
FLOWTIME_B=64,8,256 FLOWTIME_F16=1 timeout 200 python tools/flowtime.py >> $O/flowtime_ab.txt 2>&1
grep -v amdgpu.ids $O/flowtime_ab.txt > $O/flowtime_ab.tmp; mv $O/flowtime_ab.tmp $O/flowtime_ab.txt
timeout 300 python tools/flow_timeline.py > $O/flow_launch_timeline.txt 2>&1
( timeout 900 python -m pytest tests -m gpu -x -q -s 2>&1 | grep -vE "^$|amdgpu.ids" > $O/pytest.log; echo "pytest rc ${PIPESTATUS[0]}" >> $O/pytest.log )
( timeout 500 python bench.py --per-layer $O/per_layer_bair64.csv 2>$O/bench_bair64.err | tail -1 > $O/bench_bair64.json )
timeout 300 python bench.py --config land128 --steps 10 --warmup 3 --no-cpu-baseline --sustain 0 --no-exact --per-layer $O/per_layer_land128.csv 2>/dev/null | tail -1 > $O/bench_land128.json
cat $O/flowtime_ab.txt; tail -5 $O/pytest.log; cut -c1-300 $O/bench_bair64.json; tail -3 $O/bench_bair64.err
