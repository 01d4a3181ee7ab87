import os, numpy as np
from geconpy_amd import batched, _lib, workloads as wl
here = "tests"
g = np.load(os.path.join(here, "golden", "bk_eigenvalues.npz"))
rg = np.load(os.path.join(here, "golden", "reference_goldens.npz"))
fg = np.load(os.path.join(here, "golden", "failure_cases.npz"))
groups = [[(k, tuple(rg[f"{k}_{x}"] for x in "ABCD"))] for k in ("one_block", "rbc_2_block", "full_nk")]
b = wl.sw_shaped_batch(2)
groups.append([(k, tuple(fg[f"{k}_{x}"] for x in "ABCD")) for k in ("ok", "nonunique", "noexist")]
              + [(f"sw{i}", tuple(b[x][i] for x in "ABCD")) for i in range(2)])
for grp in groups:
    A, B, C, D = (np.stack([c[1][j] for c in grp]) for j in range(4))
    for db in (0, 1):
        with _lib.options_scope({'gensys_direct_blocks': db}):
            out = batched.bk_eigenvalues_batched(A, B, C, tol=1e-8)
        for i, (name, _) in enumerate(grp):
            m = int(out["n_eig"][i])
            ref_mod = np.hypot(g[f"{name}_real"], g[f"{name}_imag"])
            mod = np.hypot(out["real"][i, :m], out["imag"][i, :m])
            finite = ref_mod < 1e4
            rel = np.abs(mod[finite]-ref_mod[finite])/(1e-7*ref_mod[finite]+1e-10)
            j = rel.argmax()
            print(db, name, m, "worst", rel.max(), mod[finite][j], ref_mod[finite][j], out["real"][i,:m][finite][j], out["imag"][i,:m][finite][j])
