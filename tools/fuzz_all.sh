#!/bin/bash
# every randomised programme set, new seeds:  tools/fuzz_all.sh [first seed]      (logs under gpurun_out/fuzz_<seed>/; a summary line per set)
first=${1:-100000}
out=gpurun_out/fuzz_$first; mkdir -p $out
run() { name=$1; shift; timeout 900 "$@" > $out/$name.log 2>&1; echo "$name rc $? : $(grep -v '^GPU\|^Failed\|coredump' $out/$name.log | tail -1 | cut -c1-160)"; grep "^FAIL" $out/$name.log | cut -c1-500 | head -5; }
for c in 0 1 2 3 4 5 6 7; do run batch_reuse_$c python tools/fuzz_batch.py 100 $((first + 100 * c)) --reuse; done
for c in 0 1; do run batch_big_$c python tools/fuzz_batch.py 100 $((first + 1000 + 100 * c)) --big; done
for c in 0 1; do run batch_wide_$c python tools/fuzz_batch.py 100 $((first + 2000 + 100 * c)) --wide; done
for c in 0 1 2 3 4 5; do run handle_$c python tools/fuzz_handle.py 100 $((first + 100 * c)); done
run columns python tools/fuzz_columns.py 500 $first
run render python tools/fuzz_render.py 300 $first
run pcm python tools/fuzz_pcm.py 500 $first
run ticks python tools/fuzz_ticks.py 1000 $first
run capture python tools/fuzz_capture.py 500 $first
run streaming python tools/fuzz_streaming.py 600 $first
