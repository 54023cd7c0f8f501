#!/bin/bash
# round 3, GPU call S: conflict-free LDS layout of the batch FFT kernels (rows of 17, ds_read_b64 reads, 2-bit publish
# swizzle, b64 twiddle reads) against the previous build; LDS counters; the whole suite
O=gpurun_out/r3s; mkdir -p $O
for i in 1 2 3; do
  for lib in prev default; do
    echo "=== $lib" >> $O/ab.log
    if [ $lib = default ]; then unset SOUNDSCOPE_HIP_LIB; else export SOUNDSCOPE_HIP_LIB=$PWD/tools/bin/$lib.so; fi
    python tools/perf_probe.py 1024 10 2>&1 | grep -E "k_fft|roofline" >> $O/ab.log
    python tools/probe_cfg5.py 64 2>&1 | grep -E "tp=4x k_fft16k" >> $O/ab.log
    python tools/probe_channels.py 2>&1 | grep -E "channels (1|6)" >> $O/ab.log
    python tools/probe_native16k.py >> $O/ab.log 2>&1
  done
done
unset SOUNDSCOPE_HIP_LIB
python -m pytest tests -m gpu -q -x > $O/gpu_all.log 2>&1; echo "suite rc $?" >> $O/gpu_all.log
bash tools/pmc_one.sh r3s/pmc_lds SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS > $O/pmc_lds.txt 2>&1
cat $O/ab.log; tail -3 $O/gpu_all.log; cat $O/pmc_lds.txt
