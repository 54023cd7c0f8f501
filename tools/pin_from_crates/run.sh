#!/bin/bash
# One command from a machine with cargo (and the pinned crates reachable) to a pinned oracle:
#   tools/pin_from_crates/run.sh  &&  python -m pytest tests/test_golden_crates.py -m "not gpu"
# Uses the reference's own src/analyzer.rs when /root/reference exists, this harness's restatement of it otherwise.
set -e
here=$(cd "$(dirname "$0")" && pwd)
command -v cargo >/dev/null || { echo "no cargo on PATH: the recipe cannot run here (SURVEY section 8c)"; exit 3; }
feat=""
[ -f /root/reference/src/analyzer.rs ] || feat="--no-default-features --features crates-only"
out=$(mktemp -d)
(cd "$here" && cargo run --release $feat -- "$out")
python3 "$here/pack.py" "$out"
