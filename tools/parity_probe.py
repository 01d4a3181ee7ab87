"""GPU box: compare the fast Kalman path, the general kernel and the oracle on the worst draws."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from geconpy_amd import batched, workloads as wl
import oracle
nb = 1024
b = wl.sw_shaped_batch(nb); om = wl.sw_shaped_observation_model()
args = (b["A"], b["B"], b["C"], b["D"], b["sigma"] ** 2, om["Z"], om["y"])
kw = dict(Hdiag=om["Hdiag"], tol=1e-8, max_iter=1000, return_policy=True)
fast = batched.solve_kalman_logp_batched(*args, **kw)
gen = batched.solve_kalman_logp_batched(*args, n_state_hint=0, z_selector_hint=0, **kw)
dense = batched.solve_kalman_logp_batched(*args, z_selector_hint=0, **kw)
rel = np.abs(fast["logp"] - gen["logp"]) / np.abs(gen["logp"])
print("fast vs general: max rel", rel.max(), "argmax", rel.argmax(), " dense vs general", (np.abs(dense["logp"] - gen["logp"]) / np.abs(gen["logp"])).max())
worst = np.argsort(rel)[-3:]
for i in worst:
    T, R = fast["T"][i], fast["R"][i]
    Q = np.diag(b["sigma"][i] ** 2)
    ref_full = oracle.solve_kalman_logp(b["A"][i], b["B"][i], b["C"][i], b["D"][i], Q, om["Z"], om["y"], H=np.diag(om["Hdiag"]))
    ref_sameT = oracle.kalman_filter_logp(om["y"], T, R, Q, om["Z"], H=np.diag(om["Hdiag"]))
    rho = np.max(np.abs(np.linalg.eigvals(T)))
    print(i, "rho(T)=%.4f" % rho, "fast-ref(full) %.2e" % ((fast["logp"][i] - ref_full["logp"]) / abs(ref_full["logp"])),
          "gen-ref(full) %.2e" % ((gen["logp"][i] - ref_full["logp"]) / abs(ref_full["logp"])),
          "fast-ref(sameT) %.2e" % ((fast["logp"][i] - ref_sameT) / abs(ref_sameT)),
          "T err %.2e" % np.abs(T - ref_full["T"]).max(), "T* err %.2e" % np.abs(T - b["T_star"][i]).max())
