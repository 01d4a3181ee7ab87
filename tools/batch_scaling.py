"""GPU box: per-kernel durations of the gensys and cycle-reduction evaluations across batch sizes.

Run under rocprofv3 --kernel-trace (tools/batch_scaling.sh): the kernel trace is grouped by (kernel, grid size), so one
process covers every batch size.  Answers: how many dispatch rounds does a launch take, and what does a launch cost when every
draw of the batch is resident at once?"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from geconpy_amd import _lib, workloads as wl
from geconpy_amd.batched import lead_hint

lib = _lib.load()
dev = torch.device("cuda", 0)
sizes = [int(s) for s in sys.argv[1].split(",")] if len(sys.argv) > 1 else [512, 1024, 2048, 3072, 4096, 8192]
full = wl.sw_shaped_batch(max(sizes))
n, k = full["A"].shape[1], full["D"].shape[2]
nlh = lead_hint(full["C"][:16])
for nb in sizes:
    A, B, C = (torch.as_tensor(full[x][:nb], device=dev) for x in "ABC")
    T = torch.empty_like(A)
    eu = torch.empty((nb, 3), dtype=torch.int32, device=dev)
    st = torch.empty(nb, dtype=torch.int32, device=dev)

    def run():
        _lib.check(lib.dsge_gensys_batched(A.data_ptr(), B.data_ptr(), C.data_ptr(), None, nb, n, k, 1e-8, nlh,
                                           T.data_ptr(), None, eu.data_ptr(), st.data_ptr(), None))

    run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3):
        run()
    e1.record()
    torch.cuda.synchronize()
    print(f"gensys batch {nb}: {e0.elapsed_time(e1) / 3:.3f} ms per call; ok {int((st == 0).sum())}", flush=True)
