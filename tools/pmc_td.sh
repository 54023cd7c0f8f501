#!/bin/bash
# Issue-side PMC passes for k_time_domain on the config 3 probe (each group its own rocprofv3 run): tools/pmc_td.sh <tag>
set -u
tag=${1:-pmctd}
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/$tag
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
pmc() {
  name=$1; shift
  timeout 300 rocprofv3 --pmc "$@" -d $out/pmc_$name -o p -- python $root/tools/perf_probe.py 1024 2 > $out/pmc_$name.log 2>&1
  db=$(find $out/pmc_$name -name '*.db' | head -1)
  if [ -n "$db" ]; then
    echo "## rocprofv3 --pmc $*" >> $out/summary.txt
    python $root/tools/rocpd_summary.py "$db" | grep -E "ssk::k_(fft4096|time_domain)" | grep -v "^ *[0-9]+ +[0-9.]+ +[0-9.]+ +[0-9.]+ +[0-9.]+" >> $out/summary.txt
  fi
  rm -rf $out/pmc_$name
}
pmc a SQ_INSTS_BRANCH SQ_IFETCH SQ_WAIT_ANY SQ_ACTIVE_INST_ANY
pmc b SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_MISC SQ_INSTS_SMEM
pmc c SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_SALU SQ_INSTS_VALU
pmc d SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_FLAT
pmc e SQ_WAIT_INST_ANY SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_IFETCH_LEVEL
cat $out/summary.txt
