"""GPU box: gensys kernel time (ms per 4096 draws) across model sizes."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from geconpy_amd import _lib, workloads as wl
from geconpy_amd.batched import lead_hint
from _opts import set_option  # (tools/_opts.py: per-thread dsge_options)
lib = _lib.load(); dev = torch.device("cuda", 0); nb = 4096
for n in (8, 16, 24, 32, 40, 44, 48):
    ns, nl, k = max(2, int(0.45 * n)), max(1, int(0.3 * n)), min(7, n // 2)
    base = [wl.sw_shaped_system(2000 + i, n=n, n_state=ns, n_lead=nl, k=k) for i in range(16)]
    A, B, C = (torch.as_tensor(np.tile(np.stack([b[x] for b in base]), (nb // 16, 1, 1)), device=dev) for x in range(3))
    T = torch.empty_like(A); eu = torch.empty((nb, 3), dtype=torch.int32, device=dev); st = torch.empty(nb, dtype=torch.int32, device=dev)
    nlh = lead_hint(np.stack([b[2] for b in base]))
    def run():
        _lib.check(lib.dsge_gensys_batched(A.data_ptr(), B.data_ptr(), C.data_ptr(), None, nb, n, k, 1e-8, nlh, T.data_ptr(), None, eu.data_ptr(), st.data_ptr(), None))
    run(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); run(); run(); e1.record(); torch.cuda.synchronize()
    t_auto = e0.elapsed_time(e1) / 2
    set_option("gensys_split", 0)  # the single-launch kernel for comparison
    try:
        run(); torch.cuda.synchronize()
        e0.record(); run(); run(); e1.record(); torch.cuda.synchronize()
        t_single = e0.elapsed_time(e1) / 2
    except _lib.DsgeHipError:
        t_single = float("nan")  # does not fit the 160 KB of LDS
    finally:
        set_option("gensys_split", 1)
    print(f"n={n:2d} N={n + nlh:2d}: gensys {t_auto:.2f} ms per {nb} draws (single-launch kernel {t_single:.2f} ms); ok {int((st == 0).sum())}")
