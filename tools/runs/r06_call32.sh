#!/bin/bash
# round 6, GPU call 32: loader form of the thin F(4,3) layers with the V rows going through VGPRs + ds_write_b128 (I2V_W4_LOADER=2) vs LDS-DMA (1) vs the
# shipping kernel (0)
export TMPDIR=/tmp
O=gpurun_out/r06_32; mkdir -p $O
for s in "8 16 128 128 32 32 0 1" "8 16 128 128 64 32 0 0" "32 16 128 128 32 32 0 1" "32 16 128 128 64 32 0 0"; do
  for rep in 1 2; do for l in 2 0 1; do
    echo "== loader=$l $s: $(I2V_W4_LOADER=$l timeout 300 tools/conv16w_check $s 2>&1 | grep -E 'F\(4,3\)' | tr -s ' ' | tr '\n' '|')" >> $O/loader_ab.txt
  done; done
done
cut -c1-260 $O/loader_ab.txt
