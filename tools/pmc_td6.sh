#!/bin/bash
# round 6: k_time_domain's instruction counters by class, for one or more library builds (config 3 probe; one counter group per rocprofv3 run)
#   tools/pmc_td6.sh <tag> <lib|default> ...
set -u
tag=$1; shift
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/$tag; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
[ -f $out/counters_avail.txt ] || rocprofv3 -L > $out/counters_avail.txt 2>&1
for lib in "$@"; do
  echo "#### $lib" >> $out/summary.txt
  [ "$lib" = default ] && unset SOUNDSCOPE_HIP_LIB || export SOUNDSCOPE_HIP_LIB=$(realpath $root/$lib)
  n=0
  for grp in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VALU_FMA_F64" "SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_INT32" \
             "SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_MFMA" "SQ_INSTS_VALU_MFMA_F32 SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES" \
             "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES SQ_WAIT_INST_ANY" "SQ_INST_CYCLES_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS" \
             "SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT64 SQ_INSTS_FLAT SQ_INSTS_SMEM"; do
    n=$((n + 1))
    timeout 300 rocprofv3 --pmc $grp -d $out/p$n -o p -- python $root/tools/perf_probe.py 1024 2 > $out/log_$n.txt 2>&1
    db=$(find $out/p$n -name '*.db' | head -1)
    echo "## --pmc $grp" >> $out/summary.txt
    [ -n "$db" ] && python $root/tools/rocpd_summary.py "$db" | grep -E "ssk::k_time_domain" | grep -v "^ *[0-9]+ +[0-9.]+ +[0-9.]+ +[0-9.]+ +[0-9.]+" | cut -c1-150 >> $out/summary.txt || tail -3 $out/log_$n.txt >> $out/summary.txt
    rm -rf $out/p$n
  done
done
cat $out/summary.txt
