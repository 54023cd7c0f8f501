mkdir -p gpurun_out/r5j
for rep in 1 2; do for lib in fixup default; do echo "=== $lib (rep $rep)"; if [ $lib = default ]; then unset SOUNDSCOPE_HIP_LIB; else export SOUNDSCOPE_HIP_LIB=$PWD/tools/bin/$lib.so; fi; python tools/probe_single_file.py 2>&1 | grep -v "^ *$"; python tools/probe_file_open.py 2>&1 | tail -3; done; done > gpurun_out/r5j/ab.log 2>&1
unset SOUNDSCOPE_HIP_LIB
cat gpurun_out/r5j/ab.log
python -m pytest tests -m gpu -q -x -k "parity or session or tick or shapes or independent" 2>&1 | grep -E "passed|failed|Error" | tail -3
tools/fuzz_batch_isolate.sh 3 60001 --reuse
