#!/usr/bin/env python3
"""GPU box: hardware counters of the bench step's kernels, one rocprofv3 --pmc pass per counter group
(MI355X_MICROARCH.md: FETCH_SIZE and WRITE_SIZE cannot share a pass; 8 SQ slots), plus one --kernel-trace --stats
pass for the durations.  Writes <out>/pmc_counters.json, the file bench.py's roofline block reads
(profiles/<round>/pmc_counters.json once copied there).

    python3 tools/pmc_collect.py gpurun_out/r2_pmc [-- extra bench.py args]
    python3 tools/pmc_collect.py gpurun_out/r2_pmc_grad --script tools/grad_rate.py

Per kernel (template instance, averaged over its dispatches):
    fp64_flops   = (2 FMA_F64 + ADD_F64 + MUL_F64 + TRANS_F64) x 64 lanes + 512 MFMA_MOPS_F64   (wave-level instruction
                   counts x wave width: lanes masked off by exec are counted -- issue slots, the roofline's denominator)
    hbm_bytes    = 2 x FETCH_SIZE KB + WRITE_SIZE KB  (gfx950: FETCH_SIZE tallies 128-B requests at 64 B; the
                   guide's HBM section)
This parent never touches the GPU; each pass is `rocprofv3 ... -- python3 bench.py ...` started as a child.
"""
import collections
import csv
import glob
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GROUPS = {
    "fetch": "FETCH_SIZE",
    "write": "WRITE_SIZE",
    "flops": "SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_TRANS_F64 "
             "SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_INSTS_VALU SQ_WAVES",
    "sq1": "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU "
           "SQ_ACTIVE_INST_LDS SQ_INSTS_LDS",
    "sq2": "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_VALU_MFMA_F64 SQ_ACTIVE_INST_SCA SQ_VALU_MFMA_BUSY_CYCLES",
}


def short(name):
    name = name.split("(")[0].strip()
    return name[5:] if name.startswith("void ") else name


def main():
    out = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/pmc"
    extra = sys.argv[sys.argv.index("--") + 1:] if "--" in sys.argv else []
    os.makedirs(out, exist_ok=True)
    env = dict(os.environ, TMPDIR="/tmp")
    bench = ["python3", os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--cpu-sample", "0",
             "--profile-reps", "1", *extra]
    if "--script" in sys.argv:  # counters of another driver, e.g. --script tools/grad_rate.py (the gradient pipeline)
        bench = ["python3", os.path.join(ROOT, sys.argv[sys.argv.index("--script") + 1])]
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for tag, counters in GROUPS.items():
        d = os.path.join(out, tag)
        subprocess.run(["rm", "-rf", d])
        cmd = ["rocprofv3", "--pmc", *counters.split(), "--kernel-trace", "--output-format", "csv", "-d", d, "-o", "pmc",
               "--", *bench]
        with open(os.path.join(out, tag + ".log"), "w") as log:
            rc = subprocess.run(cmd, stdout=log, stderr=subprocess.STDOUT, env=env, cwd=ROOT).returncode
        files = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
        if rc or not files:
            print(f"pass {tag}: rc={rc}, no counter csv", file=sys.stderr)
            continue
        for row in csv.DictReader(open(files[0])):
            if "dsge" not in row["Kernel_Name"]:
                continue
            acc[short(row["Kernel_Name"])][row["Counter_Name"]].append(float(row["Counter_Value"]))
        subprocess.run(["rm", "-rf", d])
    # durations: --kernel-trace --stats of the same command
    d = os.path.join(out, "kt")
    subprocess.run(["rm", "-rf", d])
    with open(os.path.join(out, "kt.log"), "w") as log:
        subprocess.run(["rocprofv3", "--kernel-trace", "--stats", "--output-format", "csv", "-d", d, "-o", "kt", "--",
                        *bench], stdout=log, stderr=subprocess.STDOUT, env=env, cwd=ROOT)
    stats = {}
    for f in glob.glob(d + "/**/*kernel_stats.csv", recursive=True):
        subprocess.run(["cp", f, os.path.join(out, "kernel_stats.csv")])
        for row in csv.DictReader(open(f)):
            if "dsge" in row["Name"]:
                stats[short(row["Name"])] = dict(calls=int(row["Calls"]), avg_ns=float(row["AverageNs"]))
    subprocess.run(["rm", "-rf", d])
    what = " ".join(bench[1:]).replace(ROOT + "/", "")
    res = {"source": "tools/pmc_collect.py: rocprofv3 --pmc (one pass per group) + --kernel-trace --stats over `" + what + "`, MI355X",
           "groups": GROUPS,
           "corrections": "hbm_bytes = (2 x FETCH_SIZE + WRITE_SIZE) x 1024; fp64_flops = (2 FMA + ADD + MUL + TRANS) x 64 "
                          "+ 512 MFMA_MOPS; SQ_*_CYCLES are quad-cycles",
           "kernels": {}}
    for name, c in sorted(acc.items()):
        avg = {k: sum(v) / len(v) for k, v in c.items()}
        k = {"dispatches": max(len(v) for v in c.values()), **{kk: round(vv, 1) for kk, vv in avg.items()}}
        if "SQ_INSTS_VALU_FMA_F64" in avg:
            k["fp64_flops"] = (2 * avg["SQ_INSTS_VALU_FMA_F64"] + avg.get("SQ_INSTS_VALU_ADD_F64", 0) +
                               avg.get("SQ_INSTS_VALU_MUL_F64", 0) + avg.get("SQ_INSTS_VALU_TRANS_F64", 0)) * 64 + \
                512 * avg.get("SQ_INSTS_VALU_MFMA_MOPS_F64", 0)
        if "FETCH_SIZE" in avg and "WRITE_SIZE" in avg:
            k["hbm_bytes"] = (2 * avg["FETCH_SIZE"] + avg["WRITE_SIZE"]) * 1024
        if name in stats:
            k["avg_ns"] = stats[name]["avg_ns"]
            k["calls_in_trace"] = stats[name]["calls"]
        res["kernels"][name] = k
    with open(os.path.join(out, "pmc_counters.json"), "w") as fh:
        json.dump(res, fh, indent=1)
    for name, k in res["kernels"].items():
        print(f"{name:60s} n={k['dispatches']:3d} flops={k.get('fp64_flops', 0):.4g} hbm={k.get('hbm_bytes', 0):.4g} "
              f"avg_us={k.get('avg_ns', 0) / 1e3:.1f} conflict/active="
              f"{k.get('SQ_LDS_BANK_CONFLICT', 0) / max(k.get('SQ_LDS_IDX_ACTIVE', 1), 1):.2f} "
              f"valu_busy={k.get('SQ_ACTIVE_INST_VALU', 0) / max(k.get('SQ_WAVE_CYCLES', 1), 1):.3f}")


if __name__ == "__main__":
    main()
