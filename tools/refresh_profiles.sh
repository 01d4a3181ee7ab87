#!/bin/bash
# Run on the GPU box (through gpurun): the bench lines and rocprofv3 kernel statistics that profiles/<tag>/ keeps.
# Usage: bash tools/refresh_profiles.sh <tag>
set -u
TAG=${1:-r1_s5}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
python3 bench.py > "$OUT/bench_default.json" 2> "$OUT/bench_default.err"
python3 bench.py --solver gensys --cpu-sample 0 > "$OUT/bench_gensys.json" 2>/dev/null
python3 bench.py --workload rbc --cpu-sample 0 > "$OUT/bench_rbc.json" 2>/dev/null
python3 bench.py --workload rbc --solver gensys --cpu-sample 0 > "$OUT/bench_rbc_gensys.json" 2>/dev/null
python3 bench.py --workload rbc --from-theta --cpu-sample 0 > "$OUT/bench_rbc_from_theta.json" 2>/dev/null
python3 bench.py --workload full_nk --cpu-sample 0 > "$OUT/bench_full_nk.json" 2>/dev/null
python3 bench.py --workload full_nk --solver gensys --cpu-sample 0 > "$OUT/bench_full_nk_gensys.json" 2>/dev/null
for V in default gensys; do
  ARGS=""; [ "$V" = gensys ] && ARGS="--solver gensys"
  rm -rf "$OUT/kt_$V"
  rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/kt_$V" -o kt -- python3 bench.py --steps 5 --warmup 2 --cpu-sample 0 $ARGS > "$OUT/kt_$V.log" 2>&1
  find "$OUT/kt_$V" -name '*kernel_stats.csv' -exec cp {} "$OUT/kernel_stats_$V.csv" \;
  rm -rf "$OUT/kt_$V"
done
python3 tools/gensys_window_phases.py > "$OUT/gensys_window_phases.txt" 2>&1
python3 tools/kalman_phases.py > "$OUT/kalman_phases.txt" 2>&1
python3 -m pytest tests -m gpu -q > "$OUT/tests_gpu.log" 2>&1
tail -3 "$OUT/tests_gpu.log"
for f in "$OUT"/bench_*.json; do echo "$f: $(cut -c1-140 "$f")"; done
