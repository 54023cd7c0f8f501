#!/bin/bash
# Refresh the judged measurement set on the GPU box (one call): the bench line, the rocprofv3 kernel-trace summary of
# the SAME bench command, and PMC passes (each counter group in its own rocprofv3 run, no tracing beside --pmc) over
# the config 3 probe (tools/perf_probe.py) and the config 5 probe (tools/probe_cfg5.py).
# usage: tools/profile_round.sh <tag>     (outputs under gpurun_out/<tag>/; copy what is judged into profiles/)
set -u
tag=${1:-prof}
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/$tag
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
if [ "${SKIP_BENCH:-0}" != 1 ]; then        # (SKIP_BENCH=1: the PMC passes only)
python $root/bench.py --steps 20 --warmup 5 > $out/bench.json 2> $out/bench.err
rocprofv3 --kernel-trace --stats -d $out/kt -o kt -- python $root/bench.py --steps 20 --warmup 5 > $out/kt_bench.json 2> $out/kt.err
db=$(find $out/kt -name '*.db' | head -1)
[ -n "$db" ] && python $root/tools/rocpd_summary.py "$db" $out/kernel_stats.txt
rm -rf $out/kt
fi
pmc() {   # pmc <name> <probe script> <probe args> <counters...>
  name=$1; probe=$2; pargs=$3; shift 3
  timeout 600 rocprofv3 --pmc "$@" -d $out/pmc_$name -o p -- python $root/$probe $pargs > $out/pmc_$name.log 2>&1
  db=$(find $out/pmc_$name -name '*.db' | head -1)
  if [ -n "$db" ]; then
    echo "## rocprofv3 --pmc $* -- python $probe $pargs" >> $out/pmc_summary.txt
    python $root/tools/rocpd_summary.py "$db" | grep -E "ssk::k_(fft|time_domain|finalize)" | grep -v "^ *[0-9]+ +[0-9.]+ +[0-9.]+ +[0-9.]+ +[0-9.]+" >> $out/pmc_summary.txt
  fi
  rm -rf $out/pmc_$name
}
[ "${SKIP_PMC:-0}" = 1 ] && { cat $out/bench.json | head -c 600; echo; cat $out/kernel_stats.txt | head -24; exit 0; }     # bench line + kernel trace only
for cfg in "c3 tools/perf_probe.py 1024_2" "c5 tools/probe_cfg5.py 64"; do
  set -- $cfg; n=$1; probe=$2; pargs=${3//_/ }
  pmc ${n}_fetch $probe "$pargs" FETCH_SIZE
  pmc ${n}_write $probe "$pargs" WRITE_SIZE
  pmc ${n}_valu $probe "$pargs" SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES
  pmc ${n}_lds $probe "$pargs" SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS
  pmc ${n}_mfma $probe "$pargs" SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_F16 SQ_INSTS_VALU_FMA_F64 GRBM_GUI_ACTIVE
  pmc ${n}_wait $probe "$pargs" SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD
done
[ -f $out/bench.json ] && { cat $out/bench.json | head -c 600; echo; cat $out/kernel_stats.txt | head -20; }; cat $out/pmc_summary.txt | cut -c1-200 | head -60
