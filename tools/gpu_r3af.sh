#!/bin/bash
# round 3, GPU call AF: the judged measurement set on the final tree (suite, smoke, bench line + kernel trace + PMC passes, default
# bench, time-domain phase profile and issue-side counters)
O=gpurun_out/r3af; mkdir -p $O
python -m pytest tests -m gpu -q > $O/gpu_all.log 2>&1; echo "suite rc $?" >> $O/gpu_all.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1
bash tools/profile_round.sh r3af/prof > $O/profile_round.log 2>&1
python bench.py > $O/bench_default.json 2> $O/bench_default.err
SOUNDSCOPE_HIP_LIB=$PWD/tools/bin/tdprof.so python tools/probe_td_phases.py > $O/td_phases.log 2>&1
bash tools/pmc_td.sh r3af/pmc_td > /dev/null 2>&1
tail -3 $O/gpu_all.log; tail -1 $O/smoke.log; head -c 400 $O/bench_default.json; echo; cat $O/td_phases.log | head -12
