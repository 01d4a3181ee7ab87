"""geconpy_amd -- MI355X-native batched engine for gEconpy's estimation hot path
(first-order perturbation solve + Kalman-filter log-likelihood over parameter draws).

    batched        numpy arrays in / out, one C-ABI call per batch (host staging)
    engine         torch CUDA tensors resident in HBM, streams, draw sharding over ranks
    solvers        the reference's numpy-level function names on the HIP engine
    pytensor_ops   pytensor Ops mirroring gEconpy's solver Ops (lazy, optional)
    workloads      seeded synthetic inputs (RBC closed form, Smets-Wouters-shaped systems)

The compute lives in libdsge_hip.so (hand-written HIP for gfx950; include/dsge_hip.h).  There
is no CPU fallback: without the library or without a GPU every compute call raises.
"""
__version__ = "0.1.0"
