#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r06_13; mkdir -p $O
for s in "2 16 128 128 32 32 0 0 1 1" "2 16 128 128 64 32 0 0 0 2" "8 16 128 128 32 32 0 1 0 1" "8 16 128 128 64 32 0 0 0 2"; do
  echo "== $s: $(timeout 300 tools/conv16w_check $s 2>&1 | grep -E 'GEN ' | tr -s ' ' | tr '\n' '|')" >> $O/gen_bits.txt
done
cat $O/gen_bits.txt
timeout 600 python -m pytest tests -m gpu -x -q -s -k "generated_operand" 2>&1 | tail -4
