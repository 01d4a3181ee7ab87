import sys; import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from geconpy_amd import batched, _lib, workloads as wl
lib = _lib.load()
exact = {752: 3223.8898003502249284, 2950: 2985.4761888930427562}
om = wl.sw_shaped_observation_model()
for i in (752, 2950):
    b = wl.sw_shaped_batch(1, first_draw=i); q = b["sigma"] ** 2
    def run(**kw):
        r = batched.solve_kalman_logp_batched(b["A"], b["B"], b["C"], b["D"], q, om["Z"], om["y"], Hdiag=om["Hdiag"], tol=1e-8, max_iter=1000, **kw)
        return abs(r["logp"][0] - exact[i]) / exact[i]
    print(i, "default %.2e" % run())
    batched.set_kalman_steady_tol(0.0); print("   steady off %.2e" % run()); batched.set_kalman_steady_tol(1e-14)
    lib.dsge_set_cr_fused_selection(0); print("   explicit R %.2e" % run()); lib.dsge_set_cr_fused_selection(1)
    lib.dsge_set_cr_compact(0); print("   dense CR %.2e" % run()); lib.dsge_set_cr_compact(1)
    print("   general kalman kernel %.2e" % run(n_state_hint=0, z_selector_hint=0))
    print("   tol 1e-12 %.2e" % (abs(batched.solve_kalman_logp_batched(b["A"], b["B"], b["C"], b["D"], q, om["Z"], om["y"], Hdiag=om["Hdiag"], tol=1e-12, max_iter=1000)["logp"][0] - exact[i]) / exact[i]))
