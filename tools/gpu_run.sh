#!/bin/bash
# One parametrised GPU-box call (replaces the per-call tools/gpu_r3*.sh scripts of round 3).
#   gpurun --timeout 900 -- 'tools/gpu_run.sh <tag> <step> [<step> ...]'
# Every step logs into gpurun_out/<tag>/<step>.log and prints the tail of it; steps:
#   suite            python -m pytest tests -m gpu -q                      (whole GPU suite)
#   test:<expr>      python -m pytest tests -m gpu -q -k '<expr>'          (a slice of it, -x)
#   file:<path>      python -m pytest <path> -m gpu -q -x
#   smoke            python __graft_entry__.py smoke-only
#   bench[:steps]    python bench.py --steps <steps, default 20> --warmup 5
#   probe3           tools/perf_probe.py 1024 10 --check                   (config 3 per-kernel times)
#   probe5           tools/probe_cfg5.py 64                                (config 5)
#   ab3:<libs>       tools/ab_libs.sh default <comma-separated tools/bin/*.so>   (A/B at config 3, one process each)
#   ab5:<libs>       tools/ab_cfg5.sh ...                                  (A/B at config 5)
#   py:<script>[:args]   python <script> <args with _ for spaces>
#   profile          tools/profile_round.sh <tag>                          (bench line + kernel trace + PMC passes)
set -u
tag=${1:?tag}; shift
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/$tag
mkdir -p "$out"
cd "$root"
# the toolchain that would pin the oracle (tools/pin_from_crates/run.sh): reported on every call, so that the day a box has it is noticed
echo "rust toolchain on this box: cargo=$(command -v cargo || echo none) rustc=$(command -v rustc || echo none)" | tee "$out/00_toolchain.log"
i=0
for step in "$@"; do
  i=$((i + 1))
  name=${step%%:*}; arg=""; [ "$step" != "$name" ] && arg=${step#*:}
  log=$out/$(printf '%02d' $i)_$name.log
  case $name in
    suite)  python -m pytest tests -m gpu -q > "$log" 2>&1; echo "rc $?" >> "$log" ;;
    test)   python -m pytest tests -m gpu -q -x -k "$arg" > "$log" 2>&1; echo "rc $?" >> "$log" ;;
    file)   python -m pytest "$arg" -m gpu -q -x -s > "$log" 2>&1; echo "rc $?" >> "$log" ;;
    smoke)  python -c 'import __graft_entry__ as g; g.smoke()' > "$log" 2>&1; echo "rc $?" >> "$log" ;;
    bench)  python bench.py --steps "${arg:-20}" --warmup 5 > "$log" 2> "$log.err"; echo "rc $?" >> "$log.err" ;;
    probe3) python tools/perf_probe.py 1024 10 --check > "$log" 2>&1 ;;
    probe5) python tools/probe_cfg5.py 64 > "$log" 2>&1 ;;
    ab3)    tools/ab_libs.sh --args "1024 10" default $(echo "$arg" | tr ',' ' ') > "$log" 2>&1 ;;
    ab5)    tools/ab_cfg5.sh default $(echo "$arg" | tr ',' ' ') > "$log" 2>&1 ;;
    py)     script=${arg%%:*}; pargs=""; [ "$arg" != "$script" ] && pargs=${arg#*:}
            python "$script" ${pargs//_/ } > "$log" 2>&1; echo "rc $?" >> "$log" ;;
    profile) tools/profile_round.sh "$tag" > "$log" 2>&1 ;;
    *) echo "unknown step $step" > "$log" ;;
  esac
  echo "=== $step"; tail -n 12 "$log"
done
