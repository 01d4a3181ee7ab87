#!/bin/bash
# Run on the GPU box (through gpurun): kernel-trace stats of the default bench, CSV summaries
# under gpurun_out/<tag>/.  Usage: bash tools/gpu_profile.sh <tag> [extra bench args]
set -u
TAG=${1:-prof}; shift || true
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -o kt -- python3 bench.py --steps 5 --warmup 2 --cpu-sample 0 "$@" > "$OUT/bench.log" 2>&1
grep '^{' "$OUT/bench.log" > "$OUT/bench.json"
find "$OUT" -name '*kernel_stats.csv' -exec cp {} "$OUT/kernel_stats.csv" \;
# the full trace is large; keep only the stats
find "$OUT" -name '*kernel_trace.csv' -exec sh -c 'head -1 "$1" > "$2/kernel_trace_head.csv"; grep kalman "$1" | head -8 >> "$2/kernel_trace_head.csv"; rm "$1"' _ {} "$OUT" \;
cat "$OUT/kernel_stats.csv"
