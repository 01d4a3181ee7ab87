#!/bin/bash
# Run on the GPU box: one rocprofv3 --pmc pass per counter group over a short bench run; prints
# per-kernel averages.  Usage: bash tools/gpu_pmc.sh <tag> "<counters group 1>" "<group 2>" ...
set -u
TAG=${1:-pmc}; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
i=0
for GROUP in "$@"; do
  i=$((i+1))
  rocprofv3 --pmc $GROUP --kernel-trace --output-format csv -d "$OUT/g$i" -o pmc -- python3 bench.py --steps 1 --warmup 1 --cpu-sample 0 --profile-reps 1 ${PMC_BENCH_ARGS:-} > "$OUT/g$i.log" 2>&1
  python3 - "$OUT/g$i" <<'PY'
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + '/**/*counter_collection.csv', recursive=True)
if not f:
    print("no counter csv in", sys.argv[1]); sys.exit(0)
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for row in csv.DictReader(open(f[0])):
    name = row['Kernel_Name'].split('(')[0][-40:]
    acc[name][row['Counter_Name']].append(float(row['Counter_Value']))
for k, d in acc.items():
    if 'dsge' not in k: continue
    print(k, {c: round(sum(v)/len(v), 1) for c, v in d.items()}, 'dispatches', len(next(iter(d.values()))))
PY
done
