#!/bin/bash
# round 4, GPU call 17: 3-tap F(4,3) kernel (SPADE's gamma|beta convs) with both V half-requests at tap 0, A/B against the previous build
export TMPDIR=/tmp
for s in "64 1 64 64 128 512 0 0" "64 1 64 64 128 256 0 0" "64 1 32 32 128 1024 0 0" "32 1 128 128 128 128 0 0" "8 1 64 64 128 512 0 0"; do
  for b in conv16w_check_pold conv16w_check; do
    echo "$b $s: $(timeout 100 tools/$b $s 2>&1 | grep -E 'F\(4,3\) ' | tr -s ' ')"
  done
done
timeout 300 python -m pytest tests -m gpu -x -q -k "nf8_bair or full_width_bair or full_width_128 or structure_switches or shape_sweep" 2>&1 | tail -2
