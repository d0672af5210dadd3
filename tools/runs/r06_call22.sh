#!/bin/bash
# round 6, GPU call 22: is the cINN pass of step k + 1 hidden under decoder k?  (events, no profiler); direct conv with interleaved operand reads
export TMPDIR=/tmp
O=gpurun_out/r06_22; mkdir -p $O
timeout 600 python tools/pipeline_probe.py > $O/pipeline_probe.txt 2>&1; cat $O/pipeline_probe.txt | cut -c1-330
{
for rep in 1 2; do
for shape in "64 2 8 8 1024 1024 0 1 20" "64 2 8 8 1024 1024 1 1 20" "8 2 8 8 1024 1024 0 1 40" "8 2 8 8 1024 1024 1 1 40" "64 1 4 4 1024 1024 0 1 40" "8 1 64 64 128 256 0 0 20" "8 1 16 16 128 256 0 0 40"; do
  echo "-- $shape"
  for b in conv16_bench_i0 conv16_bench; do echo -n "$b  "; tools/$b $shape; done
done
done
} > $O/conv16_interleave_ab.txt 2>&1
cut -c1-170 $O/conv16_interleave_ab.txt
