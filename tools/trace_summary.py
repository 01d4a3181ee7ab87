"""Summarise a rocprofv3 --kernel-trace csv: per step, start / end of every dsge kernel relative to the solver's start (us)."""
import csv, glob, sys
f = sorted(glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True))[-1]
rows = [r for r in csv.DictReader(open(f)) if "dsge" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
steps = []
for r in rows:
    name = r["Kernel_Name"].split("(")[0].replace("void dsge::", "")
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if "cr_fused" in name or "gensys_reduce" in name:
        steps.append([])
    if steps:
        steps[-1].append((name[:40], s, e, r.get("Queue_Id", "?")))
for st in steps[-4:]:
    t0 = st[0][1]
    print(" | ".join(f"{n} q{q} [{(s - t0) / 1e3:.0f}..{(e - t0) / 1e3:.0f}]" for n, s, e, q in st))
