#!/bin/bash
# The driver's N > 1 command line on a 1-GPU box (both ranks on device 0, SS_BENCH_SHARED_GPU=1): (1) host-TCP transport, (2) RCCL
# (ncclCommInitRank refuses two ranks on one device: both ranks must end non-zero within seconds), (3) RCCL with --allow-host-fallback
out=${1:-gpurun_out/launch_path.txt}
mkdir -p "$(dirname "$out")"; : > "$out"
run() {   # name, env, extra args
  local t0=$(date +%s.%N)
  env SS_BENCH_SHARED_GPU=1 SS_COMM_TIMEOUT_S=40 $2 timeout 200 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port $((29600 + RANDOM % 300)) \
      bench.py --gpus 2 --steps 5 --warmup 2 --total-streams 512 --no-cpu --no-extra $3 > /tmp/lp.out 2> /tmp/lp.err
  local rc=$?
  echo "$1 rc $rc, $(python -c "import time,sys; print(f'{time.time()-float(sys.argv[1]):.1f}')" $t0) s wall" >> "$out"
  cut -c1-600 /tmp/lp.out >> "$out"; grep "^\[bench\]" /tmp/lp.err | cut -c1-300 >> "$out"
}
run hosttcp SS_BENCH_TRANSPORT=host-tcp ""
run rccl_same_gpu SS_BENCH_TRANSPORT=rccl ""
run rccl_fallback SS_BENCH_TRANSPORT=rccl "--allow-host-fallback"
# (4) a rank whose GPU does not exist (no shared-GPU override: LOCAL_RANK 1 on a 1-GPU box): BOTH ranks must report it within seconds
run_bad() {
  local t0=$(date +%s.%N)
  env SS_COMM_TIMEOUT_S=40 SS_BENCH_TRANSPORT=rccl SS_BENCH_FAKE_DEVICE_FOR_COMM=1 timeout 200 python tools/probe_comm_bad_rank.py > /tmp/lp.out 2>&1
  echo "bad_rank rc $?, $(python -c "import time,sys; print(f'{time.time()-float(sys.argv[1]):.1f}')" $t0) s wall" >> "$out"; cut -c1-400 /tmp/lp.out >> "$out"
}
run_bad
cat "$out"
