# clock (GRBM_GUI_ACTIVE / 8 / duration) and MFMA-busy of ablation builds of the direct-weights variant (README.md)
export TMPDIR=/tmp
mkdir -p gpurun_out/pmc8
for n in "$@"; do
  timeout 120 ./tools/conv16_bench_$n 8 > gpurun_out/pmc8/$n.time 2>&1
  timeout 120 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAVES SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --kernel-trace --output-format csv -d gpurun_out/pmc8/$n -o pmc -- ./tools/conv16_bench_$n 8 > gpurun_out/pmc8/$n.log 2>&1 || echo "pass $n failed"
done
