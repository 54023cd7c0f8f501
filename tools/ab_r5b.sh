#!/bin/bash
# round 5, call B: columns-only kernel with float LDS atomics (per bin / folded in the VALU first) against the previous commit's
out=gpurun_out/r5b; mkdir -p $out
python -m pytest tests/test_gpu_columns.py tests/test_gpu_dynamic_range.py -m gpu -q -x > $out/cols_tests.log 2>&1; tail -3 $out/cols_tests.log
for rep in 1 2; do
for lib in default fold head; do
  echo "=== columns only (160): $lib (rep $rep)"
  if [ $lib = default ]; then python tools/perf_probe.py 1024 20 --cols=160; else SOUNDSCOPE_HIP_LIB=$PWD/tools/bin/$lib.so python tools/perf_probe.py 1024 20 --cols=160; fi
done
echo "=== full rows: default (rep $rep)"; python tools/perf_probe.py 1024 20
done > $out/ab.log 2>&1
grep -E "===|k_fft|k_time" $out/ab.log
PROBE=tools/perf_probe.py tools/pmc_passes.sh r5b/pmc_cols "1024 2 --cols=160" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS" "SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD" 2>&1 | grep -E "##|fft4096" | cut -c1-150
