#!/bin/bash
# round 6, GPU call 7: the generating kernel after the lo-part fix (bits vs the writer path), SQ counters of the thin F(4,3) kernel and of the
# generating kernel on both g_4 shapes (the tool runs both kernels), and a per-kernel trace of a Landscape step with I2V_DEC_GEN=1
export TMPDIR=/tmp
O=gpurun_out/r06_7; mkdir -p $O
timeout 600 python -m pytest tests -m gpu -x -q -s -k "generated_operand" 2>&1 | tail -6 > $O/gpu_test_gen.txt
cat $O/gpu_test_gen.txt
for s in "4 16 128 128 32 32 0 1 0 1" "4 16 128 128 64 32 0 0 0 2"; do echo "== $s: $(timeout 300 tools/conv16w_check $s 2>&1 | grep -E 'GEN \(mode' | tr -s ' ')"; done
timeout 500 bash tools/pmc_sq.sh $O/pmc_c1 tools/conv16w_check 8 16 128 128 32 32 0 1 0 1 > $O/pmc_sq_gen_conv1.txt 2>&1
timeout 500 bash tools/pmc_sq.sh $O/pmc_c0 tools/conv16w_check 8 16 128 128 64 32 0 0 0 2 > $O/pmc_sq_gen_conv0.txt 2>&1
tail -3 $O/pmc_c1/p3.log
rm -rf $O/pmc_c1 $O/pmc_c0
grep -A40 "conv_wino4g\|conv_wino4_f16x3_kernel<9, 32" $O/pmc_sq_gen_conv1.txt | head -120
