#!/bin/bash
# round 3, GPU call T: the whole -m gpu suite, smoke(), and the judged measurement set on the final tree
O=gpurun_out/r3t; mkdir -p $O
python -m pytest tests -m gpu -q > $O/gpu_all.log 2>&1; echo "suite rc $?" >> $O/gpu_all.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1
bash tools/profile_round.sh r3t/prof > $O/profile_round.log 2>&1
python bench.py > $O/bench_default.json 2> $O/bench_default.err
tail -3 $O/gpu_all.log; tail -1 $O/smoke.log; head -c 1500 $O/bench_default.json
