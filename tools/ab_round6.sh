#!/bin/bash
# round 6's same-box A/B calls (logs under gpurun_out/<tag>/; summaries copied into profiles/r06_ab_*.txt)
#   a: the non-finite-sample handling (TdState::bad_key, the exact true-peak path for tiles that hold one) against round 5's
#      library at config 3 and config 5, three interleaved repetitions; and the new tests against round 5's library (they must fail there)
set -u
what=${1:?which}; tag=${2:-r6ab}
out=gpurun_out/$tag; mkdir -p $out
case $what in
  a)
    for rep in 1 2 3; do
      for lib in tools/bin/r5.so default; do
        echo "=== config 3 rep $rep $lib"
        if [ $lib = default ]; then python tools/perf_probe.py 1024 10; else SOUNDSCOPE_HIP_LIB=$(realpath $lib) python tools/perf_probe.py 1024 10; fi
      done
    done > $out/ab_nonfinite_cfg3.txt 2>&1
    for rep in 1 2; do
      for lib in tools/bin/r5.so default; do
        echo "=== config 5 rep $rep $lib"
        if [ $lib = default ]; then python tools/probe_cfg5.py 64; else SOUNDSCOPE_HIP_LIB=$(realpath $lib) python tools/probe_cfg5.py 64; fi
      done
    done > $out/ab_nonfinite_cfg5.txt 2>&1
    SOUNDSCOPE_HIP_LIB=$(realpath tools/bin/r5.so) python -m pytest tests/test_gpu_nonfinite.py -m gpu -q 2>&1 | tail -45 > $out/nonfinite_tests_on_round5_lib.txt
    grep -E "===|k_time_domain|k_fft|sum" $out/ab_nonfinite_cfg3.txt; grep -E "===|k_time_domain|k_fft16k" $out/ab_nonfinite_cfg5.txt; tail -32 $out/nonfinite_tests_on_round5_lib.txt
    ;;
  b)  # k_time_domain's instruction cuts (v_max3 in the f32 product's loop, chunk 0 from the carried state, whole staging rounds)
      # against the tree in front of them (tools/bin/r6a.so: round 5's kernels + the non-finite handling), config 3, three interleaved repetitions + the bench-shape and non-finite tests on the new build
    for rep in 1 2 3; do
      for lib in tools/bin/r6a.so default; do
        echo "=== config 3 rep $rep $lib"
        if [ $lib = default ]; then python tools/perf_probe.py 1024 10 --check; else SOUNDSCOPE_HIP_LIB=$(realpath $lib) python tools/perf_probe.py 1024 10; fi
      done
    done > $out/ab_td_cuts_cfg3.txt 2>&1
    for lib in tools/bin/r6a.so default; do
      echo "=== config 5 $lib"
      if [ $lib = default ]; then python tools/probe_cfg5.py 64; else SOUNDSCOPE_HIP_LIB=$(realpath $lib) python tools/probe_cfg5.py 64; fi
    done > $out/ab_td_cuts_cfg5.txt 2>&1
    grep -E "===|k_time_domain|k_fft|sum|err" $out/ab_td_cuts_cfg3.txt; grep -E "===|k_time_domain|k_fft16k" $out/ab_td_cuts_cfg5.txt
    python -m pytest tests/test_gpu_nonfinite.py tests/test_gpu_bench_shapes.py -m gpu -q -x 2>&1 | tail -5
    ;;
  c)  # the stereo 4x true peak on the packed-f32 VALU (v_pk_fma_f32, fifteen frames of both channels per lane) against the f32 MFMA
      # banded product of the tree in front of it (tools/bin/r6b.so), config 3, three interleaved repetitions + the tests that hold the peaks
    for rep in 1 2 3; do
      for lib in tools/bin/r6b.so default; do
        echo "=== config 3 rep $rep $lib"
        if [ $lib = default ]; then python tools/perf_probe.py 1024 10 --check; else SOUNDSCOPE_HIP_LIB=$(realpath $lib) python tools/perf_probe.py 1024 10; fi
      done
    done > $out/ab_tp_valu_cfg3.txt 2>&1
    grep -E "===|k_time_domain|k_fft|sum|err" $out/ab_tp_valu_cfg3.txt
    python -m pytest tests/test_gpu_parity.py tests/test_gpu_nonfinite.py tests/test_gpu_bench_shapes.py tests/test_gpu_independent.py -m gpu -q -x 2>&1 | tail -5
    ;;
esac
