#!/bin/bash
# round 4, GPU call 18: first V brick requested from registers before the index tables are built (A/B against the previous build)
export TMPDIR=/tmp
for s in "8 16 64 64 128 128 0 1" "8 16 64 64 256 128 1 0" "8 16 64 64 64 64 0 1" "8 8 32 32 256 256 0 1" "4 16 128 128 64 32 0 0" "4 16 128 128 32 32 0 1" "8 4 16 16 512 512 0 1" "32 16 64 64 128 128 0 1" "8 1 64 64 128 512 0 0"; do
  for b in conv16w_check_pold conv16w_check conv16w_check_pold conv16w_check; do
    echo "$b $s: $(timeout 100 tools/$b $s 2>&1 | grep -E 'F\(4,3\) +[0-9]' | tr -s ' ')"
  done
done
timeout 100 tools/conv16w_check_tl 8 16 64 64 128 128 0 1 2>&1 | grep -A8 "timeline"
timeout 400 python -m pytest tests -m gpu -x -q -k "nf8 or full_width or structure_switches or shape_sweep or baseline_batch" 2>&1 | tail -2
