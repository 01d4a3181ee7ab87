#!/bin/bash
# Run on the GPU box: HBM traffic of the pipeline kernels from two separate rocprofv3 --pmc passes
# (FETCH_SIZE, WRITE_SIZE; MI355X_MICROARCH.md HBM section: read bytes = 2 x FETCH_SIZE KB on gfx950).
# Writes gpurun_out/<tag>/pmc_traffic.json.   Usage: bash tools/gpu_traffic.sh <tag>
set -u
TAG=${1:-traffic}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d "$OUT/$C" -o pmc -- python3 bench.py --steps 1 --warmup 1 --cpu-sample 0 --profile-reps 1 > "$OUT/$C.log" 2>&1
done
python3 - "$OUT" <<'PY'
import csv, glob, json, sys, collections
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
byname = collections.defaultdict(lambda: collections.defaultdict(lambda: collections.defaultdict(list)))
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob(f"{out}/{c}/**/*counter_collection.csv", recursive=True)
    if not f:
        print("no counter csv for", c); continue
    for row in csv.DictReader(open(f[0])):
        name = row["Kernel_Name"]
        key = "solver_deflate" if "cr_deflate_kernel" in name else "solver_inflate" if "cr_inflate_kernel" in name \
            else "solver_iterate" if ("cr_compact_kernel" in name or "gensys_kernel" in name) else "assemble" if ("rqr_kernel" in name or "assemble_kernel" in name) \
            else "kalman" if "kalman_sel_kernel<3" in name else None
        if key and row["Counter_Name"] == c:
            byname[key][name.split("(")[0]][c].append(float(row["Counter_Value"]))
# a key can match several instantiations (bench.py's untimed statistics pass runs the full-size compact kernel once):
# keep the one the timed steps launch, i.e. the most frequent
for key, names in byname.items():
    best = max(names, key=lambda nm: sum(len(v) for v in names[nm].values()))
    for c, v in names[best].items():
        acc[key][c] = v
res = {"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, tools/gpu_traffic.sh), bench.py --steps 1, 4096 SW-shaped draws, MI355X",
       "correction": "read bytes = 2 x FETCH_SIZE x 1024 (gfx950 tallies 128-B requests at 64 B, MI355X_MICROARCH.md HBM section); WRITE_SIZE x 1024 uncorrected",
       "kernels": {}}
for k, d in acc.items():
    fs = sum(d["FETCH_SIZE"]) / max(len(d["FETCH_SIZE"]), 1)
    ws = sum(d["WRITE_SIZE"]) / max(len(d["WRITE_SIZE"]), 1)
    res["kernels"][k] = {"FETCH_SIZE_KB": round(fs, 1), "WRITE_SIZE_KB": round(ws, 1), "hbm_bytes_per_launch": round((2 * fs + ws) * 1024, 1)}
parts = [v for k, v in res["kernels"].items() if k.startswith("solver_")]
if parts:  # the solver leg of the fused call: deflation + iteration + inflation (one launch each)
    res["kernels"]["solver"] = {f: round(sum(v[f] for v in parts), 1) for f in ("FETCH_SIZE_KB", "WRITE_SIZE_KB", "hbm_bytes_per_launch")}
json.dump(res, open(f"{out}/pmc_traffic.json", "w"), indent=1)
print(json.dumps(res, indent=1))
PY
