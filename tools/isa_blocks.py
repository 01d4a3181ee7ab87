"""Per-basic-block instruction statistics of one kernel in a hipcc -S listing: python tools/isa_blocks.py file.s kernel_symbol"""
import re, sys
s = open(sys.argv[1]).read()
i = s.index(sys.argv[2] + ":"); j = s.index(".Lfunc_end", i)
cur = "entry"; stats = {cur: dict(mfma=0, scr=0, ds=0, vmem=0, n=0)}; order = [cur]
for l in s[i:j].split("\n")[1:]:
    t = l.strip()
    if re.match(r"^\.LBB\d+_\d+:", t):
        cur = t[:-1]; stats[cur] = dict(mfma=0, scr=0, ds=0, vmem=0, n=0); order.append(cur); continue
    if not t or t.startswith((";", ".")): continue
    st = stats[cur]; st["n"] += 1
    op = t.split()[0]
    st["mfma"] += op.startswith("v_mfma"); st["scr"] += op.startswith("scratch"); st["ds"] += op.startswith("ds_")
    st["vmem"] += op.startswith(("global_", "buffer_", "flat_"))
for k in order:
    if stats[k]["n"] > 12: print(k, stats[k])
