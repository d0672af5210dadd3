#!/bin/bash
# round 4, GPU call 2: the software-pipelined persistent F(4,3) kernel (I2V_W4_PIPE=1, the default) against round 3's structure
# (I2V_W4_PIPE=0): parity of every instantiation vs the direct kernel / fp64, timing, per-brick timeline, then the GPU suite + bench
export TMPDIR=/tmp
O=gpurun_out/r04b
mkdir -p $O
for s in "8 16 64 64 128 128 0 1" "8 16 64 64 256 128 1 0" "8 16 64 64 64 64 0 1" "8 16 64 64 128 64 1 0" "8 8 32 32 256 256 0 1" "8 8 32 32 512 256 1 0" \
         "4 16 128 128 64 32 0 0" "4 16 128 128 32 32 0 1" "3 16 64 64 128 128 0 1" "1 16 64 64 64 64 0 1" "8 1 64 64 128 128 0 0" "40 16 64 64 64 64 0 1"; do
  for pipe in 0 1; do
    echo "== PIPE $pipe shape $s" >> $O/pipe_ab.txt
    I2V_W4_PIPE=$pipe timeout 120 tools/conv16w_check $s 2>&1 | grep -E "F\(4,3\)|fault|error|Error" >> $O/pipe_ab.txt
  done
done
for s in "8 16 64 64 128 128 0 1" "8 16 64 64 256 128 1 0" "8 16 64 64 64 64 0 1" "4 16 128 128 32 32 0 1"; do
  for pipe in 0 1; do
    echo "== PIPE $pipe" >> $O/f43_timeline.txt
    I2V_W4_PIPE=$pipe timeout 120 tools/conv16w_check_tl $s 2>&1 | grep -v "^$" | grep -vE "direct |F\(2,3\) |wino vs" >> $O/f43_timeline.txt
  done
done
( timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log )
( timeout 400 python bench.py --no-cpu-baseline --per-layer $O/per_layer_bair64.csv 2>$O/bench_bair64.err | tail -1 > $O/bench_bair64.json )
I2V_W4_PIPE=0 timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras --per-layer $O/per_layer_bair64_pipe0.csv 2>/dev/null | tail -1 > $O/bench_bair64_pipe0.json
timeout 300 python bench.py --config land128 --steps 10 --warmup 3 --no-cpu-baseline --no-extras --per-layer $O/per_layer_land128.csv 2>/dev/null | tail -1 > $O/bench_land128.json
I2V_W4_PIPE=0 timeout 300 python bench.py --config land128 --steps 10 --warmup 3 --no-cpu-baseline --no-extras --per-layer $O/per_layer_land128_pipe0.csv 2>/dev/null | tail -1 > $O/bench_land128_pipe0.json
cat $O/pipe_ab.txt | head -80; tail -3 $O/pytest.log; cut -c1-300 $O/bench_bair64.json
