#!/bin/bash
# GPU box: bash tools/batch_scaling.sh <tag> [sizes]; kernel trace grouped by (kernel, grid) under gpurun_out/<tag>/
set -u
TAG=${1:-bscale}; SIZES=${2:-512,1024,2048,3072,4096,8192}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
rocprofv3 --kernel-trace --output-format csv -d "$OUT" -o kt -- python3 tools/batch_scaling.py "$SIZES" > "$OUT/run.log" 2>&1
cat "$OUT/run.log" | grep -v amdgpu.ids
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)
acc = collections.defaultdict(list)
for row in csv.DictReader(open(f[0])):
    nm = row['Kernel_Name']
    if 'dsge' not in nm: continue
    nm = nm.split('(')[0].replace('void ', '').replace('dsge::', '')
    acc[(nm, int(row['Grid_Size_X']) // max(1, int(row['Workgroup_Size_X'])), int(row.get('LDS_Block_Size', 0) or 0), int(row.get('VGPR_Count', 0) or 0))].append((int(row['End_Timestamp']) - int(row['Start_Timestamp'])) / 1e3)
with open(sys.argv[1] + '/summary.txt', 'w') as out:
    for (nm, g, lds, vg), v in sorted(acc.items()):
        line = f"{nm:28s} blocks={g:6d} lds={lds:6d} vgpr={vg:4d} calls={len(v):3d} avg_us={sum(v)/len(v):9.1f} min_us={min(v):9.1f}"
        print(line); out.write(line + "\n")
for p in f:
    import os; os.remove(p)
PY
