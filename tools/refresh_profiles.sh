#!/bin/bash
# Run on the GPU box (through gpurun): the bench lines, rocprofv3 kernel statistics and counters that profiles/r<N>/ keeps.
# Usage: bash tools/refresh_profiles.sh [tag] [round dir, default r6]   (then copy gpurun_out/<tag>/* into profiles/<round dir>/)
set -u
TAG=${1:-r6}
RND=${2:-r6}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
python3 bench.py --solver gensys --cpu-sample 0 > "$OUT/bench_gensys.json" 2>/dev/null
# (the ordered QZ for every draw: the library default is gensys by spectral division since round 5; DSGE_GENSYS_DOUBLING is read at load)
DSGE_GENSYS_DOUBLING=0 python3 bench.py --solver gensys --cpu-sample 0 --no-extras > "$OUT/bench_gensys_qz.json" 2>/dev/null
python3 bench.py --from-theta --cpu-sample 0 > "$OUT/bench_sw_from_theta.json" 2>/dev/null
python3 bench.py --workload rbc --cpu-sample 0 > "$OUT/bench_rbc.json" 2>/dev/null
python3 bench.py --workload full_nk --cpu-sample 0 > "$OUT/bench_full_nk.json" 2>/dev/null
python3 bench.py --batch-per-gpu 8192 --cpu-sample 0 --no-extras > "$OUT/bench_8192_per_gpu.json" 2>/dev/null
for V in default gensys gensys_qz sw_second_order; do
  ARGS="--no-extras"; [ "$V" = gensys ] && ARGS="--solver gensys --no-extras"; [ "$V" = gensys_qz ] && ARGS="--solver gensys --no-extras"
  [ "$V" = sw_second_order ] && ARGS="--workload sw_second_order --no-extras --steps 2 --warmup 1"
  unset DSGE_GENSYS_DOUBLING; [ "$V" = gensys_qz ] && export DSGE_GENSYS_DOUBLING=0
  rm -rf "$OUT/kt_$V"
  rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/kt_$V" -o kt -- python3 bench.py --steps 5 --warmup 2 --cpu-sample 0 $ARGS > "$OUT/kt_$V.log" 2>&1
  find "$OUT/kt_$V" -name '*kernel_stats.csv' -exec cp {} "$OUT/kernel_stats_$V.csv" \;
  rm -rf "$OUT/kt_$V" "$OUT/kt_$V.log"
done
python3 tools/pmc_collect.py "$OUT/pmc_default" -- --no-extras > "$OUT/pmc_default.txt" 2>&1 && cp "$OUT/pmc_default/pmc_counters.json" "$OUT/pmc_counters.json"
unset DSGE_GENSYS_DOUBLING
python3 tools/pmc_collect.py "$OUT/pmc_gensys_sd" -- --solver gensys --no-extras > "$OUT/pmc_gensys_spectral_division.txt" 2>&1 && cp "$OUT/pmc_gensys_sd/pmc_counters.json" "$OUT/pmc_counters_gensys_spectral_division.json"
export DSGE_GENSYS_DOUBLING=0  # (pmc_counters_gensys.json: the QZ kernels -- what the gensys_qz leg's roofline block reads)
python3 tools/pmc_collect.py "$OUT/pmc_gensys" -- --solver gensys --no-extras > "$OUT/pmc_gensys.txt" 2>&1 && cp "$OUT/pmc_gensys/pmc_counters.json" "$OUT/pmc_counters_gensys.json"
unset DSGE_GENSYS_DOUBLING
python3 tools/pmc_collect.py "$OUT/pmc_so" -- --workload sw_second_order --no-extras > "$OUT/pmc_so.txt" 2>&1 && cp "$OUT/pmc_so/pmc_counters.json" "$OUT/pmc_counters_sw_second_order.json"
rm -rf "$OUT/pmc_default" "$OUT/pmc_gensys" "$OUT/pmc_gensys_sd" "$OUT/pmc_so"
# the roofline blocks quote flops / traffic from the counters committed under profiles/: take the lines after they are refreshed
mkdir -p profiles/$RND
cp "$OUT/pmc_counters_sw_second_order.json" "$OUT/pmc_counters.json" "$OUT/pmc_counters_gensys.json" "$OUT/pmc_counters_gensys_spectral_division.json" profiles/$RND/ 2>/dev/null
python3 bench.py --workload sw_second_order > "$OUT/bench_sw_second_order.json" 2> "$OUT/bench_sw_second_order.err"
python3 bench.py > "$OUT/bench_default.json" 2> "$OUT/bench_default.err"
# (the window kernels run on every draw only with the ordered QZ for every draw: under the default route the phase counters of draw 0 stay zero)
DSGE_GENSYS_DOUBLING=0 python3 tools/gensys_window_phases.py > "$OUT/gensys_window_phases.txt" 2>&1
python3 tools/kalman_phases.py > "$OUT/kalman_phases.txt" 2>&1
python3 tools/mfma_ab.py > "$OUT/mfma_ab.txt" 2>&1
python3 tools/two_streams.py > "$OUT/two_streams.txt" 2>&1
python3 tools/grad_rate.py > "$OUT/grad_rate.txt" 2>&1
python3 tools/grad_rate.py 4096 gensys >> "$OUT/grad_rate.txt" 2>&1
python3 tools/grad_phases.py 4096 > "$OUT/grad_phases.txt" 2>&1
python3 tools/steady_hist.py > "$OUT/steady_hist.txt" 2>&1
rm -rf "$OUT/kt_grad"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/kt_grad" -o kt -- python3 tools/grad_rate.py > "$OUT/kt_grad.log" 2>&1
find "$OUT/kt_grad" -name '*kernel_stats.csv' -exec cp {} "$OUT/kernel_stats_gradient.csv" \;
rm -rf "$OUT/kt_grad" "$OUT/kt_grad.log"
python3 tools/gensys_doubling_rate.py > "$OUT/gensys_doubling_rate.txt" 2>&1
# models with 65 .. 96 variables (csrc/dsge_big.hpp)
{ for N in 72 80 96; do python3 tools/big_rate.py $N 1024; done; python3 tools/big_phases.py 80; python3 tools/big_phases.py 96; } > "$OUT/big_rate.txt" 2>&1
{ python3 tools/fuzz_big.py 0 40; python3 tools/fuzz_big.py 1 40; } > "$OUT/fuzz_big.txt" 2>&1
rm -rf "$OUT/kt_big"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/kt_big" -o kt -- python3 tools/big_rate.py 80 1024 > "$OUT/kt_big.log" 2>&1
find "$OUT/kt_big" -name '*kernel_stats.csv' -exec cp {} "$OUT/kernel_stats_big80.csv" \;
rm -rf "$OUT/kt_big" "$OUT/kt_big.log"
bash tools/batch_scaling.sh "$TAG/bscale" > /dev/null 2>&1; cp "$OUT/bscale/summary.txt" "$OUT/batch_scaling.txt" 2>/dev/null; rm -rf "$OUT/bscale"
python3 -m pytest tests -m gpu -q > "$OUT/tests_gpu.log" 2>&1
tail -3 "$OUT/tests_gpu.log"
for f in "$OUT"/bench_*.json; do echo "$f: $(cut -c1-160 "$f")"; done
