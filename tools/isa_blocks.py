"""Per-basic-block instruction statistics of one kernel in a hipcc -S listing: python tools/isa_blocks.py file.s kernel_symbol
(valu = v_* without v_mfma / v_readlane..., f64 = v_*_f64, salu = s_*, ds = LDS, vmem = global / buffer / flat / scratch)"""
import re, sys
s = open(sys.argv[1]).read()
i = s.index(sys.argv[2] + ":"); j = s.index(".Lfunc_end", i)
keys = ("n", "valu", "f64", "salu", "ds", "vmem", "mfma", "scr", "wait")
cur = "entry"; stats = {cur: dict.fromkeys(keys, 0)}; order = [cur]
for l in s[i:j].split("\n")[1:]:
    t = l.strip()
    if re.match(r"^\.LBB\d+_\d+:", t):
        cur = t.split(":")[0]; stats[cur] = dict.fromkeys(keys, 0); order.append(cur); continue
    if not t or t.startswith((";", ".")): continue
    st = stats[cur]; st["n"] += 1
    op = t.split()[0]
    st["mfma"] += op.startswith("v_mfma"); st["scr"] += op.startswith("scratch"); st["ds"] += op.startswith("ds_")
    st["vmem"] += op.startswith(("global_", "buffer_", "flat_", "scratch_"))
    st["valu"] += op.startswith("v_") and not op.startswith("v_mfma")
    st["f64"] += op.startswith("v_") and "_f64" in op
    st["salu"] += op.startswith("s_") and not op.startswith("s_waitcnt")
    st["wait"] += op.startswith("s_waitcnt")
tot = dict.fromkeys(keys, 0)
for k in order:
    for x in keys: tot[x] += stats[k][x]
    if stats[k]["n"] > (int(sys.argv[3]) if len(sys.argv) > 3 else 12): print(f"{k:12s}", stats[k])
print("total       ", tot)
