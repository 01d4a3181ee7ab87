"""GPU box: the two-draws-per-wavefront Hessenberg-triangular launch (dsge_options.gensys_pairs = 2) against the one-draw launch
(gensys_pairs = 1): same eu / status, T to 1e-10, and the call time of both."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from geconpy_amd import _lib, batched, workloads as wl
from geconpy_amd.batched import lead_hint

b = wl.sw_shaped_batch(67)
b2 = wl.sw_shaped_batch(16, n_state=14)
cases = {"sw67": tuple(b[x] for x in "ABCD"),
         "mixed": tuple(np.stack([b[x][i] if i % 2 == 0 else b2[x][i] for i in range(16)]) for x in "ABCD")}
for name, (A, B, C, D) in cases.items():
    o1 = batched.gensys_batched(A, B, C, D, tol=1e-8, options={"gensys_split": 2, "gensys_pairs": 1})
    o2 = batched.gensys_batched(A, B, C, D, tol=1e-8, options={"gensys_split": 2, "gensys_pairs": 2})
    print(name, "eu equal", np.array_equal(o1["eu"], o2["eu"]), "status equal", np.array_equal(o1["status"], o2["status"]),
          "max |dT|", float(np.abs(o1["T"] - o2["T"]).max()), "ok", int((o2["status"] == 0).sum()), "/", len(A))
lib = _lib.load()
dev = torch.device("cuda", 0)
nb = 4096
full = wl.sw_shaped_batch(nb)
n, k = 40, 7
nlh = lead_hint(full["C"][:16])
A, B, C = (torch.as_tensor(full[x], device=dev) for x in "ABC")
T = torch.empty_like(A)
eu = torch.empty((nb, 3), dtype=torch.int32, device=dev)
st = torch.empty(nb, dtype=torch.int32, device=dev)
for mode in (1, 2):
    with _lib.options_scope({"gensys_pairs": mode}):
        def run():
            _lib.check(lib.dsge_gensys_batched(A.data_ptr(), B.data_ptr(), C.data_ptr(), None, nb, n, k, 1e-8, nlh, T.data_ptr(), None,
                                               eu.data_ptr(), st.data_ptr(), None))
        run(); run()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            run()
        torch.cuda.synchronize()
        print(f"gensys_pairs = {mode}: {(time.perf_counter() - t0) / 5 * 1e3:.3f} ms per 4096 draws; ok {int((st == 0).sum())}")
