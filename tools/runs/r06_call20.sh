#!/bin/bash
# round 6, GPU call 20: idle gaps of the main queue inside one step (kernel trace); ablations of the direct conv's stage loop
export TMPDIR=/tmp
O=gpurun_out/r06_20; mkdir -p $O
for cfg in "bair64:" "bair8:--batch 8" "land128:--config land128"; do
  n=${cfg%%:*}; fl=${cfg#*:}
  rm -rf /tmp/tr_$n
  timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/tr_$n -o t -- python bench.py --steps 4 --warmup 2 --lean $fl > $O/trace_$n.log 2>&1
  python tools/step_gaps.py /tmp/tr_$n 4 > $O/step_gaps_$n.txt 2>&1
  head -70 $O/step_gaps_$n.txt | cut -c1-200
done
{
for shape in "64 2 8 8 1024 1024 0 1 20" "64 2 8 8 1024 1024 1 1 20" "8 2 8 8 1024 1024 1 1 40"; do
  echo "-- $shape"
  for b in conv16_bench conv16_bench_abl1 conv16_bench_abl2 conv16_bench_abl4 conv16_bench_abl8 conv16_bench_abl3 conv16_bench_abl6; do echo -n "$b  "; tools/$b $shape; done
done
} > $O/conv16_ablations.txt 2>&1
cut -c1-150 $O/conv16_ablations.txt
