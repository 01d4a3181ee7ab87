#!/bin/bash
# GPU box: the kernel sequence (name, grid, duration) of ONE fused step: bash tools/kernel_sequence.sh <tag> [bench args]
set -u
TAG=${1:-kseq}; shift || true
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
rocprofv3 --kernel-trace --output-format csv -d "$OUT" -o kt -- python3 bench.py --steps 1 --warmup 1 --cpu-sample 0 --no-extras --profile-reps 1 "$@" > "$OUT/run.log" 2>&1
python3 - "$OUT" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)
rows = [r for r in csv.DictReader(open(f[0])) if 'dsge' in r['Kernel_Name']]
rows.sort(key=lambda r: int(r['Start_Timestamp']))
# the last full step: from the last solver-start kernel on
starts = [i for i, r in enumerate(rows) if 'gensys_reduce' in r['Kernel_Name'] or 'cr_fused' in r['Kernel_Name'] or 'cr_deflate' in r['Kernel_Name']]
i0 = starts[-2] if len(starts) > 1 else (starts[-1] if starts else 0)
i1 = starts[-1] if len(starts) > 1 else len(rows)
t0 = int(rows[i0]['Start_Timestamp'])
with open(sys.argv[1] + '/sequence.txt', 'w') as out:
    for r in rows[i0:i1]:
        nm = r['Kernel_Name'].split('(')[0].replace('void ', '').replace('dsge::', '')
        line = f"{(int(r['Start_Timestamp']) - t0) / 1e3:9.1f} us  +{(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3:8.1f} us  blocks={int(r['Grid_Size_X']) // max(1, int(r['Workgroup_Size_X'])):5d}  {nm}"
        print(line); out.write(line + "\n")
import os
for p in f: os.remove(p)
PY
