export TMPDIR=/tmp
mkdir -p gpurun_out/pmc9
for v in base ws; do
  if [ $v = ws ]; then export I2V_CONV16_WS=1; else unset I2V_CONV16_WS; fi
  timeout 120 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAVES SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d gpurun_out/pmc9/$v -o pmc -- ./tools/conv16_bench 8 > gpurun_out/pmc9/$v.log 2>&1 || echo "pass $v failed"
done
