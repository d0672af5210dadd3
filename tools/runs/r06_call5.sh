#!/bin/bash
# round 6, GPU call 5: the operand-generating kernel after the producer rewrite (lane invariants, immediates, no SLP packing)
export TMPDIR=/tmp
O=gpurun_out/r06_5; mkdir -p $O
for s in "4 16 128 128 32 32 0 1 0 1" "4 16 128 128 64 32 0 0 0 2" "32 16 128 128 32 32 0 1 0 1" "32 16 128 128 64 32 0 0 0 2"; do
  echo "== conv16w_check $s" >> $O/gen_check.txt
  timeout 600 tools/conv16w_check $s 2>&1 | grep -v "^$" >> $O/gen_check.txt
done
cat $O/gen_check.txt
