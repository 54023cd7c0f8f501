#!/bin/bash
# round 3, GPU call R: bench.py's N > 1 launch path on a 1-GPU box (two ranks share device 0):
#  (1) host-TCP transport: a JSON line;  (2) RCCL with both ranks on one GPU: ncclCommInitRank cannot succeed — the run must
#  end quickly and non-zero on every rank;  (3) the same with --allow-host-fallback: a JSON line whose collective is host-tcp
O=gpurun_out/r3r; mkdir -p $O
export SS_BENCH_SHARED_GPU=1 SS_COMM_TIMEOUT_S=60
run() {  # run <name> <port> <env...> -- <bench args...>
  name=$1; port=$2; shift 2
  envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  t0=$(date +%s.%N); env "${envs[@]}" timeout 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port $port \
      bench.py --gpus 2 --steps 5 --warmup 2 --total-streams 512 --no-cpu --no-extra "$@" > $O/$name.json 2> $O/$name.err
  rc=$?; echo "$name rc $rc, $(python3 -c "import time,sys; print(round(time.time()-float(sys.argv[1]),1))" $t0) s wall" >> $O/summary.txt
  tail -1 $O/$name.err >> $O/summary.txt
  head -c 700 $O/$name.json >> $O/summary.txt; echo >> $O/summary.txt
  grep -E "^\[bench\]" $O/$name.err | head -6 >> $O/summary.txt
}
run hosttcp 29611 SS_BENCH_TRANSPORT=host-tcp --
run rccl_same_gpu 29612 SS_BENCH_TRANSPORT=rccl --
run rccl_fallback 29613 SS_BENCH_TRANSPORT=rccl -- --allow-host-fallback
cat $O/summary.txt
