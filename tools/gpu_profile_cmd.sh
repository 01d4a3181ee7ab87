#!/bin/bash
# Run on the GPU box: rocprofv3 kernel statistics of an arbitrary python tool.  Usage: bash tools/gpu_profile_cmd.sh <tag> <script.py> [args]
set -u
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/$TAG
rm -rf "$OUT"; mkdir -p "$OUT"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -o kt -- python3 "$@" > "$OUT/run.log" 2>&1
find "$OUT" -name '*kernel_stats.csv' -exec cp {} "$OUT/kernel_stats.csv" \;
find "$OUT" -name '*kernel_trace.csv' -delete
tail -2 "$OUT/run.log"
python3 - "$OUT/kernel_stats.csv" <<'PY'
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:8]:
    print(r["Name"][:70].ljust(70), r["Calls"], round(float(r["AverageNs"]) / 1e6, 3), r["Percentage"])
PY
