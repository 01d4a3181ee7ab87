#!/usr/bin/env python3
"""Benchmark: solve + Kalman-logp evaluations per second on Smets-Wouters-shaped systems.

    python bench.py --gpus 1 --steps 10 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

One "step" = one pass of the fused pipeline (cycle reduction -> selection/Lyapunov -> Kalman
filter) over the rank's shard of parameter draws, inputs already resident in HBM, followed
(N > 1) by the RCCL all-gather of per-draw logp/status.  Per-GPU shard is fixed (weak scaling):
4096 draws of the SW-shaped workload (n = 40 variables, 7 shocks, 7 observables, T = 200;
BASELINE.json configs[2]).  Rank 0 prints ONE JSON line.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FP64_PEAK_TFLOPS = 78.6  # MI355X FP64 vector == FP64 matrix peak (AMD datasheet)
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8 TB/s spec


def algorithmic_flops(n, k, p, T_len, cr_iters=7.0, lyap_doublings=12):
    """SURVEY.md section 8(d) formulas (the reference formulation), per evaluation."""
    m = n
    kalman = T_len * (8 * m**3 + 6 * m**2 * p + 6 * m * p**2 + p**3 / 3.0)
    lyap = lyap_doublings * 6 * m**3
    sel = (2 + 2.0 / 3.0) * n**3 + 2 * n**2 * k + 4 * n**3
    cr = 12.7 * n**3 * cr_iters + 2.7 * n**3
    return dict(kalman=kalman, lyapunov=lyap, selection=sel, solver=cr, total=kalman + lyap + sel + cr)


def deflated_static(n, A0, C0):
    """Static variables the cycle-reduction launcher deflates (launch_cr_deflated's rule): 0 = full-size iteration."""
    h = min(int(np.count_nonzero(~((A0 != 0).any(axis=0) | (C0 != 0).any(axis=0)))), 16)
    nd = n - h
    if n < 8 or h < 1 or nd < 4 or ((nd + 7) // 8 == (n + 7) // 8 and 5 * h < n):
        return 0
    return h


def executed_flops(n, k, p, T_len, s, l, u, cr_iters, n_full, n_doublings, selector=True, h=0):
    """FLOPs the kernels actually execute per draw (structure exploited).
    s = state columns of A, l = lead columns of C, u = variables the filter keeps (states + observed),
    n_full = time steps that ran the full covariance update (the rest ran the steady-state mean recursion),
    h = static variables deflated in front of the cycle reduction (the iteration then runs on n - h variables)."""
    wr = s + l
    nd = n - h
    cr = cr_iters * (nd**3 + 2 * nd * nd * wr + 2 * nd * wr * wr) + nd**3 + 2 * nd * nd * s
    cr += 2 * nd * nd * k  # the final elimination also carries D: R = -A1_hat^-1 D (fused selection)
    if h:
        cr += 4 * n * h * (h + 3 * nd + k)  # h reflectors over [B_st | B_dy | A_dy | C_dy | D]
        cr += 2 * h * nd * (2 * nd + k) + h * h * (nd + k)  # static rows: G1, right-hand sides, back-substitution
    asm = 2 * n * n * k + n * n  # what is left for the assemble kernel: sym(R Q R')
    full = 2 * s * s * u + 2 * u * u * s + 2 * u * s + 2 * u * 64 + 16 * u * u + 2 * 512 + 4 * u
    steady = 2 * u * s + 2 * u * 8 + 2 * 64
    doubling = 2 * s * s * u + 2 * u * u * s + 2 * u * s * s
    kal = n_full * full + (T_len - n_full) * steady + n_doublings * doubling
    return dict(solver=cr, assemble=asm, kalman=kal)


def algorithmic_bytes(n, k, p):
    """Compulsory HBM bytes per evaluation: inputs A,B,C,D + q + outputs."""
    return (3 * n * n + n * k + k + p) * 8 + 12


def _cpu_worker(args):
    import oracle

    A, B, C, D, q, Z, y, Hd, solver = args
    r = oracle.solve_kalman_logp(A, B, C, D, np.diag(q), Z, y, H=np.diag(Hd), solver=solver, tol=1e-8, max_iter=1000)
    return r["logp"]


def cpu_baseline(batch, om, n_sample, cores, solver="cycle_reduction"):
    """Time the CPU oracle (numpy/scipy port of the reference path) on a bounded sample of the
    same workload, one process per host core with single-threaded BLAS."""
    import multiprocessing as mp

    os.environ["OMP_NUM_THREADS"] = "1"
    os.environ["OPENBLAS_NUM_THREADS"] = "1"
    os.environ["MKL_NUM_THREADS"] = "1"
    jobs = [
        (batch["A"][i], batch["B"][i], batch["C"][i], batch["D"][i], batch["sigma"][i] ** 2, om["Z"], om["y"],
         om["Hdiag"], solver)
        for i in range(n_sample)
    ]
    ctx = mp.get_context("spawn")  # never fork a process that has initialised the GPU
    with ctx.Pool(cores) as pool:
        pool.map(_cpu_worker, jobs[: min(cores, n_sample)])  # warm-up: imports, page-in
        t0 = time.perf_counter()
        logp = pool.map(_cpu_worker, jobs, chunksize=max(1, n_sample // (4 * cores)))
        dt = time.perf_counter() - t0
    return np.array(logp), n_sample / dt, dt

GRAD_FD_DRAWS = tuple(range(0, 4096, 64))  # 64 draws spread over the batch (was 4)
GRAD_FD_EPS = 1e-5  # (largest step of the per-draw choice, see gradient_fd_reference)


def _grad_directions(shard, i):
    """Seeded direction of the directional-derivative check of draw i: dA respects the structural zeros of A (the contract of the
    gradient entry point), dq is relative to q."""
    rng = np.random.default_rng((20261003, i))
    A = shard["A"][i]
    maskA = (A != 0).any(axis=0)[None, :] * np.ones_like(A)
    q = shard["sigma"][i] ** 2
    return dict(A=rng.standard_normal(A.shape) * maskA * 0.1, B=rng.standard_normal(A.shape) * 0.1,
                C=rng.standard_normal(A.shape) * 0.1, D=rng.standard_normal(shard["D"][i].shape) * 0.1,
                q=rng.standard_normal(q.shape) * q * 0.3)


def _cpu_fd_worker(args):
    import oracle

    A, B, C, D, q, Z, y, Hd = args
    return oracle.solve_kalman_logp(A, B, C, D, np.diag(q), Z, y, H=np.diag(Hd), tol=1e-13, max_iter=1000)["logp"]


GRAD_FD_LEVELS = (1e-5, 1e-6, 1e-7)  # steps tried per draw, largest first


def gradient_fd_reference(shard, om, cores):
    """Extrapolated central differences of the CPU oracle's logp along one seeded direction per draw (GRAD_FD_DRAWS): what the device
    gradient of the `gradient` leg is checked against.  D(h) = (f(x + h d) - f(x - h d)) / (2 h) has an h^2 error term; (4 D(h/2) -
    D(h)) / 3 removes it.  The step is chosen PER DRAW: the largest h of GRAD_FD_LEVELS whose extrapolation correction |D(h/2) - D(h)| is
    below 1e-4 |D| (a draw next to the solvability boundary -- draw 2240 of the bench batch: D(1e-5) is 25 % off, f(x + 1e-4 d) does not
    exist -- needs h = 1e-7; for the others 1e-5 keeps the rounding of the oracle's logp / h at ~1e-9 of the derivative).
    -> {draw: extrapolated directional derivative}"""
    import multiprocessing as mp

    draws = [i for i in GRAD_FD_DRAWS if i < len(shard["A"])]
    jobs = []
    for i in draws:
        d = _grad_directions(shard, i)
        q = shard["sigma"][i] ** 2
        for h0 in GRAD_FD_LEVELS:
            for h in (h0, 0.5 * h0):
                for sgn in (1.0, -1.0):
                    e = sgn * h
                    jobs.append((shard["A"][i] + e * d["A"], shard["B"][i] + e * d["B"], shard["C"][i] + e * d["C"],
                                 shard["D"][i] + e * d["D"], q + e * d["q"], om["Z"], om["y"], om["Hdiag"]))
    with mp.get_context("spawn").Pool(min(cores, len(jobs))) as pool:
        vals = pool.map(_cpu_fd_worker, jobs)
    out = {}
    per = 4 * len(GRAD_FD_LEVELS)
    for j, i in enumerate(draws):
        best = None
        for k, h0 in enumerate(GRAD_FD_LEVELS):
            v = vals[per * j + 4 * k: per * j + 4 * k + 4]
            d1 = (v[0] - v[1]) / (2 * h0)
            d2 = (v[2] - v[3]) / h0
            est = (4.0 * d2 - d1) / 3.0
            if np.isfinite(est):
                best = est
                if abs(d2 - d1) <= 1e-4 * abs(d2):
                    break
        out[i] = best if best is not None else float("nan")
    return out


def gather_info(dist, world, rank, device):
    """Who took part in the logp gather of an N > 1 run -- backend (nccl = RCCL on ROCm), world size, one (rank, device index, PCI
    bus id) triple per rank -- collected with a collective of its own after the timed region, so that "did RCCL see N ranks on N
    devices" is answerable from the JSON line.  Every rank must call it."""
    import torch

    if world <= 1:
        return None
    try:
        bus = torch.cuda.get_device_properties(device).pci_bus_id
    except Exception:
        bus = None
    mine = {"rank": rank, "device": int(device.index if device.index is not None else 0), "pci_bus_id": bus,
            "host": os.uname().nodename}
    everyone = [None] * world
    dist.all_gather_object(everyone, mine)
    return {"backend": str(dist.get_backend()), "world": int(dist.get_world_size()), "ranks": everyone,
            "distinct_devices": len({(e["host"], e["device"], e["pci_bus_id"]) for e in everyone})}



PMC_LEGS = {  # kernels of one fused step, by leg: alternative GROUPS of rocprofv3 kernel-name substrings, first match wins
    "solver": (("cr_fused_kernel",), ("cr_deflate_kernel", "cr_compact_kernel", "cr_inflate_kernel"), ("cr_compact_kernel",),
               ("cr_solve_kernel",)),
    "assemble": (("rqr_kernel",),),
    "kalman": (("kalman_mf_kernel",), ("kalman_nt_kernel",), ("kalman_sel_kernel",)),
}


def load_pmc():
    """Newest profiles/r*/pmc_counters.json (tools/pmc_collect.py) -> ({leg: {flops, hbm_bytes, kernels}}, path).  Only
    kernels dispatched once per profiled step or more count (the collection run also launches one-off statistics
    kernels, e.g. the full-size cycle reduction that bench.py uses for the iteration counts)."""
    import glob

    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", "pmc_counters.json")))
    if not files:
        return {}, None
    try:
        with open(files[-1]) as fh:
            kernels = json.load(fh)["kernels"]
    except (OSError, KeyError, ValueError):
        return {}, None
    steps = max((v.get("dispatches", 0) for v in kernels.values()), default=0)
    per_step = {nm: v for nm, v in kernels.items() if steps and v.get("dispatches", 0) >= max(2, steps - 1)}
    legs = {}
    for leg, groups in PMC_LEGS.items():
        for pats in groups:
            chosen = []
            for pat in pats:  # several instantiations can match (second passes): keep the one that carries the time
                cands = [(v.get("calls_in_trace", v.get("dispatches", 0)) * v.get("avg_ns", 0.0), nm, v)
                         for nm, v in per_step.items() if pat in nm]
                if cands:
                    chosen.append(max(cands)[1:])
            if len(chosen) == len(pats):
                legs[leg] = {"kernels": [nm for nm, _ in chosen],
                             "fp64_flops": sum(v.get("fp64_flops", 0.0) for _, v in chosen),
                             "hbm_bytes": sum(v.get("hbm_bytes", 0.0) for _, v in chosen),
                             "mfma_insts": sum(v.get("SQ_INSTS_VALU_MFMA_F64", 0.0) for _, v in chosen),
                             "mfma_busy_cycles": sum(v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) for _, v in chosen),
                             "valu_insts": sum(v.get("SQ_INSTS_VALU", 0.0) for _, v in chosen),
                             "avg_ns": sum(v.get("avg_ns", 0.0) for _, v in chosen),
                             "lds_conflict_share": max((v.get("SQ_LDS_BANK_CONFLICT", 0.0) / max(v.get("SQ_LDS_IDX_ACTIVE", 1.0), 1.0)
                                                        for _, v in chosen), default=0.0)}
                break
    return legs, os.path.relpath(files[-1], ROOT)


def so_counters(kernel, nloc):
    """traffic (HBM bytes per launch, 2 x FETCH + WRITE as the guide prescribes) and the matrix-core busy fraction of the
    second-order filter kernel from the newest committed profiles/r*/pmc_counters_sw_second_order.json -- only when it was
    collected at this batch size (the counters scale with the batch)."""
    import glob

    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", "pmc_counters_sw_second_order.json")))
    none = {"traffic": None, "traffic_source": None}
    if not files or nloc != 1024:
        return none
    try:
        with open(files[-1]) as fh:
            kern = json.load(fh)["kernels"]
    except (OSError, KeyError, ValueError):
        return none
    hit = [v for nm, v in kern.items() if kernel in nm]
    if not hit:
        return none
    v = hit[0]
    src = os.path.relpath(files[-1], ROOT)
    out = {"traffic": int(v.get("hbm_bytes", 0)),
           "traffic_source": f"{src}: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE of the same bench command (committed counters, not "
                             "collected in this run)"}
    if v.get("SQ_VALU_MFMA_BUSY_CYCLES") and v.get("avg_ns"):
        # busy cycles are summed over the SIMDs; 1024 SIMDs x launch duration x 2.4 GHz is the denominator
        out["mfma_busy"] = round(v["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024 * v["avg_ns"] * 2.4), 4)
        out["mfma_busy_source"] = f"{src}: SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x committed launch duration x 2.4 GHz)"
        out["counted_tflops"] = round(v.get("fp64_flops", 0.0) / v["avg_ns"] / 1e3, 2)
    return out


def roofline_block(kern, dom, kms, pmc_src, flops, ex, nloc, total_kernel_s, b_eval, hints, u_dim, h_defl, stats):
    """The roofline object of the JSON line, for the kernel with the longest launch.  `achieved` / `frac` are EXECUTED
    FP64 flops (hardware instruction counters x 64 lanes when profiles/ holds them for this configuration, otherwise the
    analytic model of executed_flops) over the launch duration measured in this run -- a utilisation figure.  The SURVEY
    8(d) contract count of the reference formulation is reported under contract_* (it can exceed the peak: the kernels reach
    the same logp with far fewer flops)."""
    k = kern[dom]
    counted = k["counted_tflops"] is not None
    achieved = k["counted_tflops"] if counted else k["model_tflops"]
    total_counted = sum(v["counted_mflop_per_eval"] for v in kern.values()) if counted else None
    return {
        "kernel": k["kernel"],
        "bound": "valu-latency",
        "pipe": (("fp64 VALU + FP64 matrix core: the downdate and both prediction products of a full filter step are v_mfma_f64_4x4x4f64 "
                  "issues (63 per step on the SW-shaped model), the update's 7 x 7 elimination and the steady-state mean recursion are "
                  "VALU.  Neither HBM- nor MFMA-bound: a dependent chain per draw at two waves per SIMD")
                 if "kalman_mf_kernel" in k["kernel"] else
                 ("fp64 VALU (v_fma_f64; no MFMA issued in this kernel).  Neither HBM- nor MFMA-bound: a dependent chain per draw at "
                  "1-2 waves per SIMD")),
        "achieved": achieved,
        "peak": FP64_PEAK_TFLOPS,
        "unit": "TFLOP/s",
        "frac": round(achieved / FP64_PEAK_TFLOPS, 5),
        # USEFUL flops: the arithmetic of the algorithm the kernel runs on the UNPADDED reduced model (bench.py::executed_flops:
        # u filtered variables, s state columns, the measured number of full / steady steps), over the same duration.  frac counts
        # what the VALU executes (tile padding, replicated 8-lane groups included); useful_frac is what of the peak is algorithm
        "useful_tflops": k["model_tflops"],
        "useful_frac": round(k["model_tflops"] / FP64_PEAK_TFLOPS, 5),
        "flops_source": (f"{pmc_src}: rocprofv3 --pmc SQ_INSTS_VALU_{{FMA,ADD,MUL,TRANS}}_F64 x 64 lanes of the same bench "
                         f"command (tools/pmc_collect.py; committed counters, not collected in this run); duration measured in "
                         f"this run with HIP events") if counted else "analytic model bench.py::executed_flops (no counters "
                                                                     "committed for this configuration)",
        "traffic": k["hbm_bytes_per_launch"],
        "traffic_source": (f"{pmc_src}: (2 x FETCH_SIZE + WRITE_SIZE) x 1024 per launch, separate --pmc passes; committed "
                           f"counters, not collected in this run") if counted else None,
        "algorithmic_bytes_per_launch": b_eval * nloc,
        "contract_tflops": k["contract_tflops"],
        "contract_frac": round(k["contract_tflops"] / FP64_PEAK_TFLOPS, 5),
        "contract_note": ("SURVEY 8(d) flop count of the REFERENCE formulation (dense m=40 filter, 200 full covariance "
                          "updates) over the measured duration; not a utilisation figure"),
        "structure": {"n_state": hints[0], "z_selector": hints[1], "filtered_variables": u_dim,
                      "static_variables_deflated": h_defl, **stats},
        "kernel_ms": {k_: round(v, 4) for k_, v in kms.items()},
        "kernel_ms_source": ("a SEPARATE synchronised pass after the timed loop (dsge_profile_pipeline: each launch bracketed by HIP "
                             "events on the launch stream, mean of --profile-reps repetitions), not the timed steps themselves: "
                             "their sum may differ from ms_per_step by launch gaps and clock state"),
        "kernels": kern,
        "whole_eval_contract_tflops": round(flops["total"] * nloc / total_kernel_s / 1e12, 4),
        "whole_eval_model_tflops": round(sum(ex.values()) * nloc / total_kernel_s / 1e12, 4),
        "whole_eval_counted_tflops": round(total_counted * 1e6 * nloc / total_kernel_s / 1e12, 4) if counted else None,
        "hbm": {
            "algorithmic_bytes_per_eval": b_eval,
            "achieved_GBs": round(b_eval * nloc / total_kernel_s / 1e9, 3),
            "peak_GBs": HBM_PEAK_GBS,
            "frac": round(b_eval * nloc / total_kernel_s / 1e9 / HBM_PEAK_GBS, 7),
            "measured_bytes_per_step": (sum(v["hbm_bytes_per_launch"] for v in kern.values()) if counted else None),
        },
    }


GENSYS_STAGES = ("gensys_reduce_kernel<2> (structural deflation)", "gensys_hesstri_kernel (Hessenberg-triangular reduction)",
                 "gensys_sweeps_pair_kernel (real double-shift QZ sweeps, two draws per wavefront)",
                 "gensys_qzwin_kernel (complex QZ + reordering)", "gensys_eu_kernel (existence / uniqueness SVDs)",
                 "gensys_post_kernel (T in the window basis)")


def gensys_roofline(eng, call, nloc, n, n_lead):
    """The roofline object of the gensys leg: the launch durations of the window path measured live with HIP events on the launch
    stream (dsge_debug_gensys_stage_ms), executed FP64 flops of the dominant launch from the newest committed
    profiles/r*/pmc_counters_gensys.json (same batch size only), and the useful flops of the algorithm it runs."""
    import ctypes
    import glob

    import torch

    from geconpy_amd import _lib

    lib = _lib.load()
    ms = (ctypes.c_float * 8)()
    call()
    torch.cuda.synchronize()
    acc = np.zeros(8)
    reps = 3
    for _ in range(reps):
        _lib.check(lib.dsge_debug_gensys_stage_ms(1, None))
        call()
        torch.cuda.synchronize()
        _lib.check(lib.dsge_debug_gensys_stage_ms(0, ctypes.addressof(ms)))
        acc += np.array(list(ms))
    acc /= reps
    dom = int(np.argmax(acc[:6]))
    N = n + n_lead
    # real-QZ formulation 66 N^3 + 24 N^3 (reordering, worst case) + (2/3 + 2 + 4) N^3
    contract = (66 + 24 + 2.0 / 3 + 6) * N ** 3
    out = {"kernel": "dsge::" + GENSYS_STAGES[dom], "bound": "valu-latency",
           "pipe": "fp64 VALU, one or two draws per wavefront, a dependent chain of 3-row reflectors per draw (no MFMA: the 30 x 30 "
                   "window's transformations are rank-1 updates of three rows / columns)",
           "launch_ms": {GENSYS_STAGES[i].split(" ")[0]: round(float(acc[i]), 4) for i in range(6)},
           "gensys_ms": round(float(acc[6]), 4), "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s",
           "contract_mflop_per_eval": round(contract / 1e6, 3),
           "contract_tflops_whole_gensys": round(contract * nloc / (acc[6] * 1e-3) / 1e12, 3),
           "contract_note": "SURVEY 8(d) real-QZ count on the FULL (n + #lead)-dimensional pencil over the measured duration of the "
                            "gensys launches; the kernels deflate the zero columns of A first and iterate on the 30 x 30 window"}
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", "pmc_counters_gensys.json")))
    counted = None
    if files and nloc == 4096:
        try:
            with open(files[-1]) as fh:
                kern = json.load(fh)["kernels"]
            key = GENSYS_STAGES[dom].split(" ")[0].split("<")[0]
            hit = [v for nm, v in kern.items() if key in nm]
            if hit:
                counted = hit[0]
        except (OSError, KeyError, ValueError):
            counted = None
    if counted:
        out["achieved"] = round(counted.get("fp64_flops", 0.0) / (acc[dom] * 1e-3) / 1e12, 3)
        out["frac"] = round(out["achieved"] / FP64_PEAK_TFLOPS, 5)
        out["traffic"] = int(counted.get("hbm_bytes", 0))
        out["flops_source"] = (f"{os.path.relpath(files[-1], ROOT)}: rocprofv3 --pmc SQ_INSTS_VALU_{{FMA,ADD,MUL,TRANS}}_F64 x 64 lanes "
                               "(committed counters, not collected in this run); duration measured in this run with HIP events")
    else:
        out["achieved"] = out["frac"] = out["traffic"] = None
        out["flops_source"] = "no committed counters for this configuration (profiles/r*/pmc_counters_gensys.json)"
    return out


def second_order_leg(timeout=900):
    """BASELINE configs[4] as a leg of the default line: a CHILD process runs `bench.py --workload sw_second_order` (1024 draws)
    before this process touches the GPU and its JSON line is embedded; None if it fails."""
    import subprocess

    cmd = [sys.executable, os.path.abspath(__file__), "--workload", "sw_second_order", "--steps", "3", "--warmup", "1",
           "--cpu-sample", "0", "--no-extras"]
    try:
        res = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, check=False)
        line = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
        if res.returncode != 0 or not line:
            return {"error": f"child exited with {res.returncode}: {res.stderr[-300:]}"}
        full = json.loads(line[-1])
        keep = ("value", "unit", "ms_per_step", "config", "roofline", "full_recursion", "failed_draws", "stage_ms")
        leg = {k_: full[k_] for k_ in keep if k_ in full}
        leg["note"] = ("second-order perturbation + pruned-state-space filter, 1024 SW-shaped draws, measured by a child process "
                       "of this run; the reference has no second-order solver (perturbation.py:97-98 raises): parity UNPINNED "
                       "against it, checked against oracle/second_order.py by tests/test_gpu_second_order.py")
        return leg
    except (subprocess.TimeoutExpired, OSError, ValueError) as exc:
        return {"error": str(exc)}


def spawn_ranks(n_ranks, argv, shared_gpu=False, timeout=None):
    """Start ``n_ranks`` fresh ``python bench.py`` processes (RANK/LOCAL_RANK/WORLD_SIZE/MASTER_* in their
    environment, 127.0.0.1 rendezvous), wait for them and return the worst exit code.  Rank 0 prints the JSON
    line on the inherited stdout.  A rank that dies takes the others down (exact PIDs, never a pattern)."""
    import socket
    import subprocess

    import torch  # device_count() does not initialise the GPU on this image; the launcher stays off it

    n_dev = torch.cuda.device_count()
    if n_dev < n_ranks and not shared_gpu:
        print(f"bench.py: --gpus {n_ranks} but only {n_dev} device(s) visible (pass --allow-shared-gpu to put several "
              f"ranks on one device over gloo: a functional check, not a measurement)", file=sys.stderr)
        return 2
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(n_ranks):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n_ranks), LOCAL_WORLD_SIZE=str(n_ranks),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__), *argv], env=env))
    t_end = None if timeout is None else time.time() + timeout
    worst = 0
    alive = list(procs)
    while alive:
        time.sleep(0.2)
        for p in list(alive):
            rc = p.poll()
            if rc is None:
                continue
            alive.remove(p)
            if rc != 0:
                worst = worst or (rc if rc > 0 else 128 - rc)
                for q in alive:  # a dead rank leaves the others blocked in a collective
                    q.terminate()
        if t_end is not None and time.time() > t_end:
            for q in alive:
                q.kill()
            worst = worst or 124
    return worst


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=800,
                    help="timed steps (default 800: ~1.2 s of timed region at 1.5 ms per step, long enough for an outside "
                         "observer -- rocm-smi sampling -- to see the GPU busy)")
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch-per-gpu", type=int, default=None,
                    help="draws per GPU; default: 4096 at --gpus 1 (BASELINE configs[2]), 8192 at --gpus N > 1 (configs[3]: "
                         "65 536 draws over 8 GPUs), 1024 for --workload sw_second_order (configs[4])")
    ap.add_argument("--cpu-sample", type=int, default=192, help="evaluations timed on the host cores (0 = skip)")
    ap.add_argument("--profile-reps", type=int, default=3)
    ap.add_argument("--tol", type=float, default=1e-8)
    ap.add_argument("--max-iter", type=int, default=1000)
    ap.add_argument("--workload", default="sw_shaped", choices=["sw_shaped", "rbc", "full_nk", "sw_second_order"])
    ap.add_argument("--no-extras", action="store_true",
                    help="skip the two extra legs of the default line (full recursion: kalman_steady_tol = 0; solver = gensys)")
    ap.add_argument("--solver", default="cycle_reduction", choices=["cycle_reduction", "gensys"])
    ap.add_argument("--from-theta", action="store_true",
                    help="rbc workload only: start each step from the parameter draws (generated Jacobian kernel on the "
                         "device, SURVEY 8 f1) instead of from resident A,B,C,D")
    ap.add_argument("--no-hints", action="store_true", help="disable the structure hints (general kernels only)")
    ap.add_argument("--allow-shared-gpu", action="store_true",
                    help="functional check on a box with fewer devices than ranks: rank r uses device r %% device_count "
                         "and the gather runs over gloo (the JSON line says so; not a measurement)")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # plain `python bench.py --gpus N`: this process becomes the launcher of its N rank processes.  It never
        # touches the GPU (no torch.cuda call, no HIP call) and never re-execs itself; the ranks are fresh children.
        sys.exit(spawn_ranks(args.gpus, sys.argv[1:], shared_gpu=args.allow_shared_gpu))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if rank == 0:
            print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}", file=sys.stderr)
        sys.exit(2)

    # CPU baseline first (rank 0, N = 1 only), in spawned workers, before this process touches the GPU
    from geconpy_amd import workloads as wl

    if args.workload == "sw_second_order":
        return main_second_order(args, world, rank, local_rank)
    per_gpu = args.batch_per_gpu if args.batch_per_gpu else (4096 if world == 1 else 8192)
    global_batch = per_gpu * world
    lo, hi = wl.shard_bounds(global_batch, world, rank)
    if args.workload == "rbc":  # BASELINE configs[1] (informational; the metric is quoted on sw_shaped)
        n, k, p, T_len = 8, 1, 1, 200
        shard, om = wl.rbc_batch(hi - lo, first_draw=lo)
    elif args.workload == "full_nk":  # SURVEY 8d sanity configuration (informational)
        n, k, p, T_len = 24, 4, 3, 200
        shard, om = wl.full_nk_batch(hi - lo, first_draw=lo)
    else:
        sh = wl.SW_SHAPE
        n, k, p, T_len = sh["n"], sh["k"], sh["p"], sh["T_len"]
        om = wl.sw_shaped_observation_model()
        if args.from_theta:  # the 37-parameter affine family around draw 0; the host twin feeds the CPU checker only
            sw_theta = wl.sw_theta_draws(hi - lo, first_draw=lo)
            tA, tB, tC, tD, tq = wl.sw_theta_jacobians(sw_theta)
            shard = dict(A=tA, B=tB, C=tC, D=tD, sigma=np.sqrt(tq))
        else:
            shard = wl.sw_shaped_batch(hi - lo, first_draw=lo)

    cpu = None
    cpu_logp = None
    if rank == 0 and args.cpu_sample > 0:  # (also at N > 1: rank 0's shard; the other ranks wait in the rendezvous)
        try:  # one worker per PHYSICAL core: with SMT siblings as workers the per-core rate halves and says nothing more
            import psutil

            cores = psutil.cpu_count(logical=False) or os.cpu_count() or 1
        except ImportError:
            cores = os.cpu_count() or 1
        n_sample = min(max(args.cpu_sample, 24 * cores), hi - lo)  # ~10 s of CPU work per core, a few seconds of wall
        cpu_logp, cpu_rate, cpu_dt = cpu_baseline(shard, om, n_sample, cores)
        cpu = {
            "value": round(cpu_rate, 3),
            "unit": "evals/s",
            "cores": cores,
            "kind": "port",
            "sample": f"first {n_sample} draws of {'rank 0 shard of ' if world > 1 else ''}the same SW-shaped batch, numpy/scipy oracle "
                      f"(cycle reduction + bilinear Lyapunov + Joseph-form Kalman), {cores} worker processes (one per physical "
                      f"core) x 1 BLAS thread, {cpu_dt:.1f} s wall",
        }

    # checker samples of the extra legs (rank 0, N = 1, default line only): the oracle with solver = gensys, and with the
    # observation model of the realistic-structure leg (seven observed JUMP variables) -- also before the GPU is touched
    want_extras = (world == 1 and not args.no_extras and not args.from_theta and args.workload == "sw_shaped"
                   and args.solver == "cycle_reduction")
    cpu_gensys = cpu_jumps = om_j = b80 = om80 = cpu_n80 = om_c = cpu_cons = grad_fd = None
    if want_extras:
        om_j = wl.sw_shaped_observation_model(observed=wl.SW_OBSERVED_JUMPS)
        shape80 = dict(n=80, n_state=36, n_lead=24, k=10, p=7, T_len=200)  # the `n80` leg: a model beyond 64 variables
        b80 = wl.sw_shaped_batch(128, **shape80)
        om80 = wl.sw_shaped_observation_model(**shape80)
        # the `conservative` leg: observed jump variables AND 10 % of the entries of y missing at random (seeded): the mask
        # changes at almost every step, so the filter never reaches a steady state and every step is a full covariance update
        om_c = dict(om_j)
        y_c = om_j["y"].copy()
        y_c[np.random.default_rng(20261003).random(y_c.shape) < 0.10] = np.nan
        om_c["y"] = y_c
    if want_extras and rank == 0 and args.cpu_sample > 0:
        n_x = min(hi - lo, max(64, 2 * cores))
        cpu_gensys = cpu_baseline(shard, om, n_x, cores, solver="gensys")
        cpu_jumps = cpu_baseline(shard, om_j, n_x, cores)
        cpu_n80 = cpu_baseline(b80, om80, min(128, max(16, cores)), cores)
        cpu_cons = cpu_baseline(shard, om_c, n_x, cores, solver="gensys")
        grad_fd = gradient_fd_reference(shard, om, cores)

    so_leg = second_order_leg() if (want_extras and rank == 0) else None

    import torch
    import torch.distributed as dist

    from geconpy_amd.engine import LogpEngine, ShardedLogpEvaluator

    n_dev = torch.cuda.device_count()
    shared = world > 1 and n_dev < world
    if shared and not args.allow_shared_gpu:
        print(f"bench.py: rank {rank}: {world} ranks but {n_dev} device(s)", file=sys.stderr)
        sys.exit(2)
    dev_index = local_rank % max(n_dev, 1)
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    if world > 1:
        # one rank per GPU over RCCL ("nccl" IS RCCL on ROCm); RCCL refuses two ranks on one device, so the
        # functional shared-device check gathers over gloo instead (ShardedLogpEvaluator stages through the host)
        if shared:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=device)
    eng = LogpEngine(device)
    dA, dB, dC, dD = (eng.to_device(shard[x]) for x in "ABCD")
    dq = eng.to_device(shard["sigma"] ** 2)
    dZ, dy, dH = eng.to_device(om["Z"]), eng.to_device(om["y"]), eng.to_device(om["Hdiag"])
    nloc = hi - lo
    logp_buf = torch.empty(nloc, dtype=torch.float64, device=device)
    stat_buf = torch.empty(nloc, dtype=torch.int32, device=device)

    # structure hints (verified on the device per draw; they never change results)
    hints = eng.structure_hints(dA, dZ) if not args.no_hints else (0, 0)
    from geconpy_amd import _lib
    from geconpy_amd.batched import lead_hint

    n_lead = lead_hint(shard["C"], args.tol) if args.solver == "gensys" else 0

    prog = d_theta = jac_out = None
    if args.from_theta:
        if args.workload == "rbc":
            from geconpy_amd.jacobian_codegen import rbc_linearized_program

            prog = rbc_linearized_program()
            th = wl.rbc_prior_draws(hi, seed=1)
            d_theta = eng.to_device(np.stack([th[k_][lo:hi] for k_ in ("sigma", "phi", "alpha", "beta", "delta", "rho_A",
                                                                      "sigma_A")], axis=1))
        elif args.workload == "sw_shaped":
            # the 37-parameter affine family around draw 0 (jacobian_codegen.sw_shaped_program): every step starts from
            # 296 B of parameters per draw; A, B, C, D (40 KB per draw) are produced in HBM by the generated kernel
            from geconpy_amd.jacobian_codegen import sw_shaped_program

            prog = sw_shaped_program()
            d_theta = eng.to_device(sw_theta)
        else:
            ap.error("--from-theta needs --workload rbc or sw_shaped")
        jac_out = (dA, dB, dC, dD, dq)
        eng.jacobians_from_theta(prog, d_theta, out=jac_out)  # the hints below describe THIS family's structure
        torch.cuda.synchronize()
        hints = eng.structure_hints(dA, dZ) if not args.no_hints else (0, 0)

    # the number of static variables is a property of the model like the other two hints: with it the fused call is a
    # pure enqueue (no measuring launch, no read-back inside the library)
    opts = {"n_static_hint": eng.static_hint(dA, dC)} if (args.solver == "cycle_reduction" and not args.no_hints) else None

    def local_eval(lo_, hi_):
        if prog is not None:
            return eng.logp_from_theta(prog, d_theta, dZ, dy, Hdiag=dH, jac_out=jac_out, tol=args.tol,
                                       max_iter=args.max_iter, logp=logp_buf, status=stat_buf, solver=args.solver,
                                       n_state_hint=hints[0], z_selector_hint=hints[1], n_lead_hint=n_lead, options=opts)
        return eng.solve_kalman_logp(dA, dB, dC, dD, dq, dZ, dy, Hdiag=dH, q_mode=1, tol=args.tol,
                                     max_iter=args.max_iter, logp=logp_buf, status=stat_buf, solver=args.solver,
                                     n_state_hint=hints[0], z_selector_hint=hints[1], n_lead_hint=n_lead, options=opts)

    ev = ShardedLogpEvaluator(global_batch, local_eval, device)
    if world > 1 and not ev.host_staged:  # the engine writes its outputs straight into the rank's slice of the gather record
        logp_buf, stat_buf = ev.local_logp, ev.local_status

    def barrier():
        if world > 1:
            dist.barrier()

    for _ in range(args.warmup):
        ev.step()
    torch.cuda.synchronize()
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        logp_all, stat_all = ev.step()
    torch.cuda.synchronize()
    barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([dt], dtype=torch.float64, device=device)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())

    # When the caller's --steps make the timed region shorter than a second (the driver's 20 steps are 30 ms), the same step runs
    # on for >= 1.2 s OUTSIDE the headline timer: an outside observer (rocm-smi sampling) then sees the GPU busy, and the
    # sustained rate corroborates `value` (same loop, same buffers; never reported as `value`).
    sustained = None
    if dt < 1.0:
        n_sus = int(np.ceil(1.2 / (dt / args.steps)))
        torch.cuda.synchronize()
        barrier()
        t1 = time.perf_counter()
        for _ in range(n_sus):
            logp_all, stat_all = ev.step()
        torch.cuda.synchronize()
        barrier()
        dt_s = time.perf_counter() - t1
        sustained = {"value": round(global_batch * n_sus / dt_s, 2), "unit": "evals/s", "steps": n_sus,
                     "seconds": round(dt_s, 3), "note": "the headline step repeated for >= 1.2 s after the timed region "
                                                        "(rank-0 clock; corroboration for outside observers, not `value`)"}

    logp_host = logp_all.cpu().numpy()
    stat_host = stat_all.cpu().numpy()
    n_fail = int((stat_host != 0).sum())
    ginfo = gather_info(dist, world, rank, device)

    # Two more legs of the same loop, outside the headline timer (N = 1): the full recursion (kalman_steady_tol = 0 per call:
    # what data with changing missing-data masks costs, statespace.py:1432-1505) and the reference's default estimation solver
    # (configure(..., solver="gensys"), statespace.py:832).
    extras = {}
    if want_extras:
        def timed(fn, steps):
            for _ in range(2):
                fn()
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(steps):
                fn()
            torch.cuda.synchronize()
            return (time.perf_counter() - t1) / steps

        opts_full = dict(opts or {}, kalman_steady_tol=0.0)
        lp_full = torch.empty_like(logp_buf)
        dt_full = timed(lambda: eng.solve_kalman_logp(dA, dB, dC, dD, dq, dZ, dy, Hdiag=dH, q_mode=1, tol=args.tol,
                                                      max_iter=args.max_iter, logp=lp_full, status=stat_buf, solver=args.solver,
                                                      n_state_hint=hints[0], z_selector_hint=hints[1], options=opts_full),
                        max(3, min(args.steps, 20) // 2))
        extras["full_recursion"] = {"value": round(nloc / dt_full, 2), "ms_per_step": round(dt_full * 1e3, 4), "unit": "evals/s",
                                    "note": "same step with kalman_steady_tol = 0 (per call): every one of the T_len filter steps "
                                            "updates the covariance, as pymc_extras' standard filter does",
                                    "max_rel_logp_diff_vs_headline": float((torch.abs(lp_full - logp_all[lo:hi]) /
                                                                            torch.abs(lp_full)).max().item())}
        # the documented accuracy / speed knob of the steady-state switch: dsge_options.kalman_steady_tol = 1e-10 instead of the
        # rounding-level default 1e-14 (the default stays: the suite's own bars are 1e-10 per step)
        opts_relaxed = dict(opts or {}, kalman_steady_tol=1e-10)
        lp_rel = torch.empty_like(logp_buf)
        dt_rel = timed(lambda: eng.solve_kalman_logp(dA, dB, dC, dD, dq, dZ, dy, Hdiag=dH, q_mode=1, tol=args.tol,
                                                     max_iter=args.max_iter, logp=lp_rel, status=stat_buf, solver=args.solver,
                                                     n_state_hint=hints[0], z_selector_hint=hints[1], options=opts_relaxed),
                       max(3, min(args.steps, 20) // 2))
        extras["steady_tol_1e-10"] = {"value": round(nloc / dt_rel, 2), "ms_per_step": round(dt_rel * 1e3, 4), "unit": "evals/s",
                                      "note": "same step with kalman_steady_tol = 1e-10 (per call; default 1e-14): the covariance "
                                              "recursion freezes earlier, the never-steady draws included; NOT the headline",
                                      "max_rel_logp_diff_vs_full_recursion": float((torch.abs(lp_rel - lp_full) /
                                                                                   torch.abs(lp_full)).max().item())}
        nl_g = lead_hint(shard["C"], args.tol)

        def gensys_leg(extra_opts):
            lp_x = torch.empty_like(logp_buf)
            st_x = torch.empty_like(stat_buf)
            o_x = dict(opts or {}, **extra_opts) if extra_opts else opts
            call = lambda: eng.solve_kalman_logp(dA, dB, dC, dD, dq, dZ, dy, Hdiag=dH, q_mode=1, tol=args.tol,  # noqa: E731
                                                 max_iter=args.max_iter, logp=lp_x, status=st_x, solver="gensys",
                                                 n_state_hint=hints[0], z_selector_hint=hints[1], n_lead_hint=nl_g, options=o_x)
            dt_x = timed(call, max(3, min(args.steps, 20) // 2))
            leg = {"value": round(nloc / dt_x, 2), "ms_per_step": round(dt_x * 1e3, 4), "unit": "evals/s",
                   "failed_draws": int((st_x != 0).sum().item()),
                   "max_rel_logp_diff_vs_headline": float((torch.abs(lp_x - logp_all[lo:hi]) / torch.abs(lp_x)).max().item())}
            if cpu_gensys is not None:  # against the ORACLE's gensys (LAPACK's ordered QZ), not against the headline
                ref_g = cpu_gensys[0]
                rel_g = np.abs(lp_x[: len(ref_g)].cpu().numpy() - ref_g) / np.abs(ref_g)
                leg["parity"] = {"max_rel_logp_err_vs_cpu_oracle_gensys": float(rel_g.max()),
                                 "median_rel_logp_err_vs_cpu_oracle_gensys": float(np.median(rel_g)),
                                 "n_checked": int(len(ref_g)), "cpu_oracle_gensys_evals_per_s": round(cpu_gensys[1], 2)}
            return leg, call, lp_x, st_x, o_x

        # solver = gensys, the reference's default estimation solver (configure(..., solver='gensys'), statespace.py:832), with the
        # library's defaults: gensys by spectral division (csrc/dsge_gensys_doubling.hpp) -- the doubling iteration computes the
        # solvent, a per-draw certificate stands for eu = [1, 1, 0], the ordered QZ takes every draw without one
        extras["gensys"], _call_g, lp_g, st_g, o_g = gensys_leg(None)
        extras["gensys"]["note"] = ("same step with solver = gensys at the library's defaults (dsge_options.gensys_doubling = 1): cycle "
                                    "reduction computes the solvent T, the device certifies rho(T[S,S]) < 1 and rho(((B + C T)^-1 C)[L,L]) "
                                    "< 1 per draw (= gensys's eu = [1, 1, 0]); a draw without the certificate is solved by the ordered QZ. "
                                    "Same eu / status as the QZ on every system of tests/test_gpu_gensys_doubling.py and "
                                    "profiles/r5/fuzz_gensys_doubling.txt")
        try:
            with _lib.options_scope(o_g or {}):
                ms_g = eng.profile_kernels(dA, dB, dC, dD, dq, dZ, dy, Hdiag=dH, q_mode=1, tol=args.tol, max_iter=args.max_iter,
                                           reps=3, n_state_hint=hints[0], z_selector_hint=hints[1], solver="gensys", n_lead_hint=nl_g)
            extras["gensys"]["stage_ms"] = {k_: round(float(v_), 4) for k_, v_ in ms_g.items()}
        except Exception as exc:  # (never lose the line to a diagnostic)
            extras["gensys"]["stage_ms"] = {"error": repr(exc)}
        # HBM bytes of one step of this leg from the committed counters of the same command (tools/refresh_profiles.sh)
        try:
            import glob as _glob
            _f = sorted(_glob.glob(os.path.join(ROOT, "profiles", "r*", "pmc_counters_gensys_spectral_division.json")))
            if _f and nloc == 4096:
                with open(_f[-1]) as fh:
                    _k = json.load(fh)["kernels"]
                _per = {nm: v["hbm_bytes"] for nm, v in _k.items() if nm.startswith("dsge::") and v.get("dispatches", 0) > 1}
                extras["gensys"]["hbm"] = {"measured_bytes_per_step": int(sum(_per.values())),
                                           "largest": {nm: int(b_) for nm, b_ in sorted(_per.items(), key=lambda kv: -kv[1])[:4]},
                                           "source": os.path.relpath(_f[-1], ROOT) + ": (2 x FETCH_SIZE + WRITE_SIZE) per launch of every "
                                                     "kernel that runs once per step (committed counters, not collected in this run)"}
        except Exception as exc:
            extras["gensys"]["hbm"] = {"error": repr(exc)}
        # the ordered QZ for EVERY draw (dsge_options.gensys_doubling = 0): the reference's algorithm operation by operation
        extras["gensys_qz"], _call_q, lp_q, st_q, _ = gensys_leg({"gensys_doubling": 0})
        extras["gensys_qz"]["note"] = ("solver = gensys with dsge_options.gensys_doubling = 0: the ordered QZ of the pencil for every draw "
                                       "(window path, six launches)")
        extras["gensys"]["status_equal_to_qz_path"] = bool(torch.equal(st_q, st_g))
        extras["gensys"]["max_rel_logp_diff_vs_qz_path"] = float((torch.abs(lp_q - lp_g) / torch.abs(lp_q)).max().item())
        with _lib.options_scope({"gensys_doubling": 0}):
            extras["gensys_qz"]["roofline"] = gensys_roofline(eng, _call_q, nloc, n, nl_g)
        # realistic observation structure: the seven observed series are JUMP variables (growth rates, inflation, hours in a
        # Smets-Wouters data set; _make_design_matrix allows any, statespace.py:260-332), so the filter runs on the 18 state
        # variables + 7 observed ones = 25 (32-wide tile) instead of the 18 of SURVEY 8(d)'s generator (24-wide tile)
        dZj, dyj, dHj = eng.to_device(om_j["Z"]), eng.to_device(om_j["y"]), eng.to_device(om_j["Hdiag"])
        hints_j = eng.structure_hints(dA, dZj) if not args.no_hints else (0, 0)
        lp_j = torch.empty_like(logp_buf)
        st_j = torch.empty_like(stat_buf)
        dt_j = timed(lambda: eng.solve_kalman_logp(dA, dB, dC, dD, dq, dZj, dyj, Hdiag=dHj, q_mode=1, tol=args.tol,
                                                   max_iter=args.max_iter, logp=lp_j, status=st_j, solver=args.solver,
                                                   n_state_hint=hints_j[0], z_selector_hint=hints_j[1], options=opts),
                     max(3, min(args.steps, 20) // 2))
        extras["observe_jumps"] = {"value": round(nloc / dt_j, 2), "ms_per_step": round(dt_j * 1e3, 4), "unit": "evals/s",
                                   "note": f"same step, Z selects the non-state variables {list(wl.SW_OBSERVED_JUMPS)}: 25 filtered "
                                           "variables (18 states + 7 observed jumps), the 32-wide filter tile",
                                   "failed_draws": int((st_j != 0).sum().item())}
        if cpu_jumps is not None:
            ref_j = cpu_jumps[0]
            rel_j = np.abs(lp_j[: len(ref_j)].cpu().numpy() - ref_j) / np.abs(ref_j)
            extras["observe_jumps"]["parity"] = {"max_rel_logp_err_vs_cpu_oracle": float(rel_j.max()),
                                                 "median_rel_logp_err_vs_cpu_oracle": float(np.median(rel_j)),
                                                 "n_checked": int(len(ref_j))}
        # CONSERVATIVE configuration -- what a reference user with configure()'s defaults and real data runs: solver = gensys (the
        # default, statespace.py:832), observed jump variables, 10 % of the observations missing at random (mixed-frequency / ragged
        # data: prepare_mixed_frequency_data, statespace.py:1432-1505; the mask of statespace.py:1143 then changes at almost every
        # step) and kalman_steady_tol = 0: no structure of the data or of the recursion is exploited beyond the exact state reduction
        dyc = eng.to_device(om_c["y"])
        lp_c = torch.empty_like(logp_buf)
        st_c = torch.empty_like(stat_buf)
        opts_c = {"kalman_steady_tol": 0.0, "gensys_doubling": 0}
        dt_c = timed(lambda: eng.solve_kalman_logp(dA, dB, dC, dD, dq, dZj, dyc, Hdiag=dHj, q_mode=1, tol=args.tol,
                                                   max_iter=args.max_iter, logp=lp_c, status=st_c, solver="gensys",
                                                   n_state_hint=hints_j[0], z_selector_hint=hints_j[1], n_lead_hint=nl_g,
                                                   options=opts_c), max(3, min(args.steps, 20) // 2))
        extras["conservative"] = {"value": round(nloc / dt_c, 2), "ms_per_step": round(dt_c * 1e3, 4), "unit": "evals/s",
                                  "note": "solver = gensys (the reference's default, statespace.py:832), Z selects seven JUMP variables "
                                          "(25 filtered variables), 10 % of the entries of y missing at random (the missing-data mask "
                                          "changes at almost every step), kalman_steady_tol = 0: every one of the T_len steps is a full "
                                          "covariance update -- the rate of a user with configure() defaults and ragged data",
                                  "missing_entries": int(np.isnan(om_c["y"]).sum()), "failed_draws": int((st_c != 0).sum().item())}
        extras["conservative"]["note"] += "; the ordered QZ for every draw (gensys_doubling = 0)"
        lp_c2 = torch.empty_like(logp_buf)
        dt_c2 = timed(lambda: eng.solve_kalman_logp(dA, dB, dC, dD, dq, dZj, dyc, Hdiag=dHj, q_mode=1, tol=args.tol,
                                                    max_iter=args.max_iter, logp=lp_c2, status=st_c, solver="gensys",
                                                    n_state_hint=hints_j[0], z_selector_hint=hints_j[1], n_lead_hint=nl_g,
                                                    options={"kalman_steady_tol": 0.0}), max(3, min(args.steps, 20) // 2))
        extras["conservative"]["with_default_gensys"] = {
            "value": round(nloc / dt_c2, 2), "ms_per_step": round(dt_c2 * 1e3, 4),
            "max_rel_logp_diff": float((torch.abs(lp_c2 - lp_c) / torch.abs(lp_c)).max().item()),
            "note": "same leg with gensys at the library's default (spectral division, QZ for uncertified draws)"}
        if cpu_cons is not None:
            ref_c = cpu_cons[0]
            rel_c = np.abs(lp_c[: len(ref_c)].cpu().numpy() - ref_c) / np.abs(ref_c)
            extras["conservative"]["parity"] = {"max_rel_logp_err_vs_cpu_oracle_gensys": float(rel_c.max()),
                                                "median_rel_logp_err_vs_cpu_oracle_gensys": float(np.median(rel_c)),
                                                "n_checked": int(len(ref_c)), "bar": 1e-8,
                                                "cpu_oracle_evals_per_s": round(cpu_cons[1], 2)}
        # logp + reverse-mode gradient of the same batch (what a NUTS step costs: solver pullback, reverse Kalman sweep, assembly;
        # SURVEY 8 f2), with the directional-derivative check against central differences of the CPU oracle
        try:
            g_out = [None]

            def grad_call():
                g_out[0] = eng.solve_kalman_logp_grad(dA, dB, dC, dD, dq, dZ, dy, Hdiag=dH, tol=args.tol, max_iter=args.max_iter,
                                                      n_filter_hint=hints[0], out=g_out[0])

            dt_gr = timed(grad_call, 3)
            go = g_out[0]
            extras["gradient"] = {"value": round(nloc / dt_gr, 2), "ms_per_step": round(dt_gr * 1e3, 4), "unit": "logp+gradient evals/s",
                                  "note": "dsge_solve_kalman_logp_grad_batched on the same 4096 draws: logp and its cotangents with respect "
                                          "to A, B, C, D, q (policy-function adjoints by doubling on the compact nl x ns Stein equation, the reverse "
                                          "of the state-space assembly on the same elimination of B + C T; forward filter sweep = the "
                                          "tile-layout logp kernel with record output, reverse sweep with its products on the FP64 matrix "
                                          "core)",
                                  "failed_draws": int((go["status"] != 0).sum().item()),
                                  "max_rel_logp_diff_vs_headline": float((torch.abs(go["logp"] - logp_all[lo:hi]) /
                                                                          torch.abs(go["logp"])).max().item())}
            if grad_fd is not None:
                errs = []
                for i, fd in grad_fd.items():
                    dirs = _grad_directions(shard, i)
                    an = sum(float((go[f"{k_}_bar"][i].cpu().numpy() * dirs[k_]).sum()) for k_ in ("A", "B", "C", "D", "q"))
                    errs.append(abs(an - fd) / max(1.0, abs(fd)))
                extras["gradient"]["parity"] = {"max_rel_directional_derivative_err_vs_cpu_oracle_fd": float(max(errs)),
                                                "median_rel_directional_derivative_err_vs_cpu_oracle_fd": float(np.median(errs)),
                                                "n_checked": len(errs), "fd_steps": list(GRAD_FD_LEVELS), "bar": 1e-6,
                                                "note": "<gradient, direction> against the Richardson-extrapolated central difference "
                                                        "(steps h, h / 2; h chosen per draw from fd_steps) of the CPU oracle's logp along "
                                                        "one seeded direction per draw (all five cotangents at once), 64 draws spread "
                                                        "over the batch"}
        except Exception as exc:  # (never lose the headline line to an extra leg)
            extras["gradient"] = {"error": repr(exc)}
        # the same batch evaluated by TWO callers at once, each on its own stream (two PyMC chains sharing the GPU, or a sampler
        # that splits its particles): the library keeps its scratch per (device, stream); one sequence leaves most of the chip idle
        # while the Kalman launch waits for its never-steady draw, a second one fills that time.  Whole-GPU rate, never `value`.
        s2 = [torch.cuda.Stream(), torch.cuda.Stream()]
        lp2 = [torch.empty_like(logp_buf) for _ in s2]
        st2 = [torch.empty_like(stat_buf) for _ in s2]

        def both():
            for i, stq in enumerate(s2):
                with torch.cuda.stream(stq):
                    eng.solve_kalman_logp(dA, dB, dC, dD, dq, dZ, dy, Hdiag=dH, q_mode=1, tol=args.tol, max_iter=args.max_iter,
                                          logp=lp2[i], status=st2[i], solver=args.solver, n_state_hint=hints[0],
                                          z_selector_hint=hints[1], options=opts)

        dt_2 = timed(both, max(3, min(args.steps, 20) // 2))
        extras["two_streams"] = {"value": round(2 * nloc / dt_2, 2), "ms_per_round": round(dt_2 * 1e3, 4), "unit": "evals/s",
                                 "note": "two independent callers, each evaluating the same batch on its own HIP stream, whole-GPU "
                                         "rate (2 x batch per round); the headline `value` is the ONE-stream figure",
                                 "bit_identical_to_headline": bool(all(torch.equal(x, logp_all[lo:hi]) for x in lp2))}
        if so_leg is not None:
            extras["second_order"] = so_leg
        # a model beyond 64 variables (csrc/dsge_big.hpp: cycle reduction with one workgroup per draw, the filter on the model
        # restricted to its state and observed variables): SW-shaped systems scaled to n = 80, 1024 draws (128 distinct)
        try:
            t8 = lambda x: eng.to_device(np.tile(x, (8,) + (1,) * (x.ndim - 1)))  # noqa: E731
            A8, B8, C8, D8, q8 = t8(b80["A"]), t8(b80["B"]), t8(b80["C"]), t8(b80["D"]), t8(b80["sigma"] ** 2)
            Z8, y8, H8 = eng.to_device(om80["Z"]), eng.to_device(om80["y"]), eng.to_device(om80["Hdiag"])
            lp8 = torch.empty(1024, dtype=torch.float64, device=device)
            st8 = torch.empty(1024, dtype=torch.int32, device=device)
            dt8 = timed(lambda: eng.solve_kalman_logp(A8, B8, C8, D8, q8, Z8, y8, Hdiag=H8, q_mode=1, tol=args.tol,
                                                      max_iter=args.max_iter, logp=lp8, status=st8, solver="cycle_reduction",
                                                      z_selector_hint=1), 3)
            extras["n80"] = {"value": round(1024 / dt8, 2), "ms_per_step": round(dt8 * 1e3, 4), "unit": "evals/s",
                             "note": "SW-shaped systems scaled to n = 80 (36 states, 24 forward-looking, 10 shocks, 7 observed), "
                                     "1024 draws (128 distinct), cycle reduction: the 65..96-variable path (csrc/dsge_big.hpp)",
                             "failed_draws": int((st8 != 0).sum().item())}
            # the same model with solver = gensys: beyond 64 variables gensys exists by spectral division only (csrc/dsge_big.hpp:
            # gensys_certify_big_kernel -- the doubling iteration, then the certificate of eu = [1, 1, 0]; no ordered QZ at this size)
            lp8g = torch.empty(1024, dtype=torch.float64, device=device)
            st8g = torch.empty(1024, dtype=torch.int32, device=device)
            dt8g = timed(lambda: eng.solve_kalman_logp(A8, B8, C8, D8, q8, Z8, y8, Hdiag=H8, q_mode=1, tol=args.tol,
                                                       max_iter=args.max_iter, logp=lp8g, status=st8g, solver="gensys",
                                                       z_selector_hint=1), 3)
            extras["n80"]["gensys"] = {"value": round(1024 / dt8g, 2), "ms_per_step": round(dt8g * 1e3, 4),
                                       "failed_draws": int((st8g != 0).sum().item()),
                                       "max_rel_logp_diff_vs_cycle_reduction": float((torch.abs(lp8g - lp8) / torch.abs(lp8)).max().item())}
            if cpu_n80 is not None:
                ref8 = cpu_n80[0]
                rel8 = np.abs(lp8[: len(ref8)].cpu().numpy() - ref8) / np.abs(ref8)
                extras["n80"]["parity"] = {"max_rel_logp_err_vs_cpu_oracle": float(rel8.max()),
                                           "median_rel_logp_err_vs_cpu_oracle": float(np.median(rel8)),
                                           "n_checked": int(len(ref8)), "cpu_oracle_evals_per_s": round(cpu_n80[1], 2)}
        except Exception as exc:  # (never lose the headline line to an extra leg)
            extras["n80"] = {"error": repr(exc)}
        local_eval(0, nloc)  # (leave the buffers as the headline loop left them)
        torch.cuda.synchronize()

    # per-kernel durations, HIP events on the launch stream (rank 0's shard)
    kms = eng.profile_kernels(dA, dB, dC, dD, dq, dZ, dy, Hdiag=dH, q_mode=1, tol=args.tol, max_iter=args.max_iter,
                              reps=args.profile_reps, n_state_hint=hints[0], z_selector_hint=hints[1],
                              solver=args.solver, n_lead_hint=n_lead)
    torch.cuda.synchronize()

    # structure statistics of rank 0's shard (one untimed pass): CR iterations, first steady-state step
    import ctypes  # noqa: F401

    from geconpy_amd import _lib

    stats = {}
    if rank == 0 and args.workload == "sw_shaped":
        lib = _lib.load()
        it_buf = torch.zeros(nloc, dtype=torch.int32, device=device)
        st_buf = torch.zeros(nloc, dtype=torch.int32, device=device)
        T_buf = torch.empty_like(dA)
        _lib.check(lib.dsge_cycle_reduction_batched(dA.data_ptr(), dB.data_ptr(), dC.data_ptr(), nloc, n, args.max_iter,
                                                    args.tol, T_buf.data_ptr(), st_buf.data_ptr(), it_buf.data_ptr(),
                                                    torch.cuda.current_stream(device).cuda_stream))
        at = torch.full((nloc,), -1, dtype=torch.int32, device=device)
        eng.record_steady_steps(at)
        local_eval(0, nloc)
        torch.cuda.synchronize()
        eng.record_steady_steps(None)
        at_h = at.cpu().numpy()
        stats = {
            "cr_iters_mean": float(it_buf.float().mean().item()),
            "lead_columns": int((shard["C"][0] != 0).any(axis=0).sum()),
            "first_steady_step_median": int(np.median(at_h)),
            "full_steps_mean": float(np.where(at_h < 0, T_len, at_h).mean()),
            "never_steady": int((at_h < 0).sum()),
        }
        del T_buf

    if rank == 0:
        # hardware counters of the same command, collected by tools/pmc_collect.py on an MI355X and committed under
        # profiles/ (rocprofv3 cannot profile the process it runs in): FP64 VALU instruction counts and HBM bytes per
        # launch.  Only valid for the default workload / batch / solver; the JSON line names the file.
        pmc, pmc_src = {}, None
        default_cfg = (per_gpu == 4096 and hints[1] and args.workload == "sw_shaped" and args.solver == "cycle_reduction"
                       and not args.no_hints)
        if default_cfg:
            pmc, pmc_src = load_pmc()
        cr_it = stats.get("cr_iters_mean", 7.0)
        flops = algorithmic_flops(n, k, p, T_len, cr_iters=cr_it)
        # contract flops by the kernel that does the work (the Lyapunov solve now runs inside the Kalman kernel)
        # (selection: R comes out of the cycle-reduction kernel's final elimination unless --solver gensys)
        fused_sel = args.solver == "cycle_reduction"
        contract = {"solver": flops["solver"] + (flops["selection"] if fused_sel else 0.0),
                    "assemble": 2.0 * n * n * k if fused_sel else flops["selection"],
                    "kalman": flops["kalman"] + flops["lyapunov"]}
        s_cols = hints[0] or n
        u_dim = min(n, s_cols + p) if not hints[1] else int(np.count_nonzero((shard["A"][0] != 0).any(axis=0)
                                                                             | (om["Z"] != 0).any(axis=0)))
        h_defl = deflated_static(n, shard["A"][0], shard["C"][0]) if args.solver == "cycle_reduction" else 0
        ex = executed_flops(n, k, p, T_len, s_cols, stats.get("lead_columns", n), u_dim, cr_it,
                            stats.get("full_steps_mean", T_len), 8, selector=bool(hints[1]), h=h_defl)
        bs_n, bs_d = (n + 7) // 8, (n - h_defl + 7) // 8
        np_lo = 8 * (((hints[0] if 0 < (hints[0] or 0) < n else n) + 7) // 8)  # launch_kalman.hip::kalman_folds_rqr
        folds_rqr = bool(args.solver == "cycle_reduction" and hints[1] and p <= 8 and 1 <= k <= 16
                         and not (p <= 3 and 0 < (hints[0] or 0) and (hints[0] or 0) + p <= 6)
                         and n * ((k + 1) & ~1) <= np_lo * (np_lo + 2))
        nd_ = n - h_defl
        ncol_ = h_defl + 3 * nd_ + k  # launch_cr_fused: two columns of the QR per lane up to 128, three up to 192
        one_launch = (h_defl and nd_ + k <= 64 and
                      ((ncol_ <= 128 and (bs_n, bs_d) in {(3, 2), (3, 3), (4, 3), (4, 4), (5, 4), (6, 4), (6, 5)}) or
                       (128 < ncol_ <= 192 and (bs_n, bs_d) in {(6, 5), (7, 5), (7, 6), (8, 6)})))
        mf_default = __import__('geconpy_amd._lib', fromlist=['make_options']).make_options().kalman_mfma == 2
        names = {"solver": ((f"dsge::cr_fused_kernel{'_occ2' if bs_d == 4 and ncol_ <= 128 else ''}<{bs_n},{bs_d}> (static-variable deflation {n} -> {nd_}: "
                             "QR of the static columns + cycle reduction + back-substitution, one launch)") if one_launch else
                            (f"dsge::cr_deflate_kernel<{bs_n}> + cr_compact_kernel<{bs_d}> + cr_inflate_kernel<{bs_d}> (static-variable "
                             f"deflation {n} -> {nd_}, three launches)") if h_defl else f"dsge::cr_compact_kernel<{bs_n}>")
                 if args.solver == "cycle_reduction" else ("dsge::gensys_reduce_kernel + gensys_qzwin_kernel + gensys_post_kernel (window path, three launches)"
                                if n > 16 else "dsge::gensys_kernel"),
                 "assemble": (("none: sym(R Q R')[U,U] is formed in the prologue of the Kalman kernel (rqr_kernel<16> only for "
                               "draws handed on to the general filter)") if folds_rqr else
                              "dsge::rqr_kernel<16>" if (args.solver == "cycle_reduction" and k <= 16)
                              else f"dsge::assemble_kernel<{(n + 7) // 8}>"),
                 "kalman": ((f"dsge::kalman_mf_kernel<5,{5 if u_dim <= 20 else 7}> (covariance in the tile layout of v_mfma_f64_4x4x4f64: downdate "
                             f"and both prediction products on the FP64 matrix core)")
                            if (hints[1] and p <= 8 and 17 <= hints[0] <= 20 and u_dim <= 28 and mf_default) else
                            f"dsge::kalman_nt_kernel<{(u_dim + 7) // 8}>" if hints[1] and p <= 8
                            else f"dsge::kalman_sel_kernel<{(u_dim + 7) // 8},{'true' if hints[1] else 'false'}>")}
        kern = {}
        for key in ("solver", "assemble", "kalman"):
            sec = kms[key] * 1e-3
            leg = pmc.get(key, {})
            kern[key] = {
                "kernel": names[key],
                "ms": round(kms[key], 4),
                "contract_mflop_per_eval": round(contract[key] / 1e6, 3),
                "contract_tflops": round(contract[key] * nloc / sec / 1e12, 3),
                "model_mflop_per_eval": round(ex[key] / 1e6, 3),
                "model_tflops": round(ex[key] * nloc / sec / 1e12, 3),
                "counted_mflop_per_eval": round(leg["fp64_flops"] / nloc / 1e6, 3) if leg else None,
                "counted_tflops": round(leg["fp64_flops"] / sec / 1e12, 3) if leg else None,
                "hbm_bytes_per_launch": round(leg["hbm_bytes"]) if leg else None,
                "lds_conflict_share": round(leg["lds_conflict_share"], 3) if leg else None,
                # FP64 matrix-core evidence from the committed counters (north_star: "MFMA only for the F P F' products of the
                # Kalman recursion"): wave-level MFMA instructions per launch, their share of the VALU instruction stream, and the
                # busy share of the 1024 SIMDs' matrix pipes over the committed launch duration at 2.4 GHz
                "mfma_insts_per_launch": round(leg["mfma_insts"]) if leg else None,
                "mfma_share_of_valu_insts": round(leg["mfma_insts"] / leg["valu_insts"], 4) if leg and leg.get("valu_insts") else None,
                "mfma_busy": (round(leg["mfma_busy_cycles"] / (1024 * leg["avg_ns"] * 2.4), 4)
                              if leg and leg.get("avg_ns") else None),
            }
        dom = max(kern, key=lambda k_: kern[k_]["ms"])
        total_kernel_s = sum(kms.values()) * 1e-3
        b_eval = algorithmic_bytes(n, k, p)
        value = global_batch * args.steps / dt
        out = {
            "metric": "solve+Kalman-logp evals/sec, Smets-Wouters n~40 T=200",
            "value": round(value, 2),
            "unit": "evals/s",
            "n_gpus": world,
            **({"shared_device": f"{world} ranks on {n_dev} device(s), gloo gather: functional check, not a measurement"}
               if shared else {}),
            **({"gather": ginfo} if ginfo else {}),
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 4),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {
                "workload": (f"sw_shaped synthetic (SURVEY 8d): n=m={n}, n_state=18, n_lead=12, k={k}, p={p}, "
                             f"T_len={T_len}, {per_gpu} draws per GPU " +
                             ("(BASELINE configs[2])" if world == 1 else
                              f"x {world} GPUs = {global_batch} draws (BASELINE configs[3]: 65 536 draws sharded over 8 GPUs, "
                              f"8192 per GPU)")) if args.workload == "sw_shaped"
                else (f"full_nk golden system with seeded 1e-3 relative perturbations (SURVEY 8d sanity configuration): n={n}, "
                      f"k={k}, p={p}, T_len={T_len}, {per_gpu} draws per GPU") if args.workload == "full_nk"
                else f"rbc_linearized closed form: n={n}, k={k}, p={p}, T_len={T_len}, {per_gpu} draws per GPU (BASELINE configs[1])",
                "global_batch": global_batch,
                "solver": args.solver,
                "inputs": (f"theta resident in HBM ({d_theta.shape[1]} parameters = {8 * d_theta.shape[1]} B per draw); A,B,C,D are produced "
                           f"by the generated Jacobian kernel inside every timed step") if args.from_theta else "A,B,C,D resident in HBM",
                "tol": args.tol,
                "kalman_steady_tol": __import__('geconpy_amd._lib', fromlist=['make_options']).make_options().kalman_steady_tol,
                "cr_static_deflation": (f"{h_defl} static variables (zero columns of A and C) eliminated by a QR of their columns of B before the "
                                        f"iteration, which runs on {n - h_defl} variables; verified per draw on the device") if h_defl else "none",
                "kalman_dispatch": ("workgroups in descending order of the draws' cycle-reduction iteration counts" if args.solver == "cycle_reduction"
                                    else "workgroups in descending order of a persistence key of T (power iteration)") + " (slow draws first; outputs stay in draw order)",
                "parallelism": f"draw-sharded x{world}, one all_gather of packed (logp,status) records" if world > 1 else "single GPU",
            },
            "roofline": roofline_block(kern, dom, kms, pmc_src, flops, ex, nloc, total_kernel_s, b_eval, hints, u_dim, h_defl,
                                       stats),
            "failed_draws": n_fail,
            **({"sustained": sustained} if sustained else {}),
            **extras,
        }
        if cpu is not None:
            out["cpu_baseline"] = cpu
            ns = len(cpu_logp)
            rel = np.abs(logp_host[:ns] - cpu_logp) / np.abs(cpu_logp)
            out["parity"] = {"max_rel_logp_err_vs_cpu_oracle": float(rel.max()),
                             "median_rel_logp_err_vs_cpu_oracle": float(np.median(rel)),
                             "n_above_1e-12": int((rel > 1e-12).sum()), "n_checked": ns}
            out["gpu_over_cpu"] = round(value / cpu["value"], 1)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


def _cpu_worker_so(args):
    from oracle import second_order as so

    A, B, C, D, idx, val, q, Z, y, Hd = args
    return so.solve_second_order_logp(A, B, C, D, idx, val, np.diag(q), Z, y, H=np.diag(Hd), tol=1e-8)["logp"]


def main_second_order(args, world, rank, local_rank):
    """BASELINE configs[4]: second-order perturbation (generalised Sylvester) + pruned-state-space Kalman filter on the
    SW-shaped systems, 1024 draws per GPU.  One step = first-order solve -> second-order coefficients -> pruned system ->
    stationary covariance -> 207-dimensional filter, inputs (A, B, C, D, Hessian values) resident in HBM."""
    import ctypes
    import multiprocessing as mp

    from geconpy_amd import workloads as wl

    per_gpu = args.batch_per_gpu if args.batch_per_gpu else 1024
    global_batch = per_gpu * world
    lo, hi = wl.shard_bounds(global_batch, world, rank)
    nloc = hi - lo
    sh = wl.SW_SHAPE
    n, k, p, T_len = sh["n"], sh["k"], sh["p"], sh["T_len"]
    shard = wl.sw_second_order_batch(nloc, first_draw=lo)
    om = wl.sw_shaped_observation_model()
    cpu = None
    cpu_logp = None
    if rank == 0 and args.cpu_sample > 0:
        try:
            import psutil

            cores = psutil.cpu_count(logical=False) or os.cpu_count() or 1
        except ImportError:
            cores = os.cpu_count() or 1
        os.environ["OMP_NUM_THREADS"] = os.environ["OPENBLAS_NUM_THREADS"] = os.environ["MKL_NUM_THREADS"] = "1"
        n_sample = min(max(16, 2 * cores), nloc)  # ~3 s of CPU work per draw
        jobs = [(shard["A"][i], shard["B"][i], shard["C"][i], shard["D"][i], shard["hess_idx"], shard["hess_val"][i],
                 shard["sigma"][i] ** 2, om["Z"], om["y"], om["Hdiag"]) for i in range(n_sample)]
        with mp.get_context("spawn").Pool(cores) as pool:
            pool.map(_cpu_worker_so, jobs[: min(cores, n_sample)])
            t0 = time.perf_counter()
            cpu_logp = np.array(pool.map(_cpu_worker_so, jobs, chunksize=1))
            cpu_dt = time.perf_counter() - t0
        cpu = {"value": round(n_sample / cpu_dt, 3), "unit": "evals/s", "cores": cores, "kind": "port",
               "sample": f"first {n_sample} draws of the same batch, numpy/scipy oracle (oracle/second_order.py: cycle reduction, "
                         f"Sylvester by a Schur back-substitution, 207-dimensional Joseph-form filter), {cores} worker processes x 1 "
                         f"BLAS thread, {cpu_dt:.1f} s wall"}

    import torch
    import torch.distributed as dist

    from geconpy_amd import _lib
    from geconpy_amd.engine import LogpEngine, ShardedLogpEvaluator

    n_dev = torch.cuda.device_count()
    shared = world > 1 and n_dev < world
    if shared and not args.allow_shared_gpu:
        print(f"bench.py: rank {rank}: {world} ranks but {n_dev} device(s)", file=sys.stderr)
        sys.exit(2)
    dev_index = local_rank % max(n_dev, 1)
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    if world > 1:
        dist.init_process_group("gloo") if shared else dist.init_process_group("nccl", device_id=device)
    eng = LogpEngine(device)
    dA, dB, dC, dD = (eng.to_device(shard[x]) for x in "ABCD")
    dq = eng.to_device(shard["sigma"] ** 2)
    dhv = eng.to_device(shard["hess_val"])
    dhi = torch.as_tensor(shard["hess_idx"], dtype=torch.int32, device=device).contiguous()
    dZ, dy, dH = eng.to_device(om["Z"]), eng.to_device(om["y"]), eng.to_device(om["Hdiag"])
    structure = eng.second_order_structure(dA, dC, dZ)
    S, Lc, U = structure
    s, u = len(S), len(U)
    m = 2 * u + s * (s + 1) // 2
    logp_buf = torch.empty(nloc, dtype=torch.float64, device=device)
    stat_buf = torch.empty(nloc, dtype=torch.int32, device=device)

    def local_eval(lo_, hi_, stage_ms=None, options=None):
        return eng.second_order_logp(dA, dB, dC, dD, dhi, dhv, dq, dZ, dy, structure, Hdiag=dH, tol=args.tol,
                                     max_iter=args.max_iter, logp=logp_buf, status=stat_buf, stage_ms=stage_ms, options=options)

    ev = ShardedLogpEvaluator(global_batch, local_eval, device)
    if world > 1 and not ev.host_staged:  # the engine writes its outputs straight into the rank's slice of the gather record
        logp_buf, stat_buf = ev.local_logp, ev.local_status
    for _ in range(args.warmup):
        ev.step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        logp_all, stat_all = ev.step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([dt], dtype=torch.float64, device=device)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
    logp_host = logp_all.cpu().numpy()
    n_fail = int((stat_all != 0).sum().item())
    ginfo = gather_info(dist, world, rank, device)
    # stage durations (HIP events on the launch stream inside the library) and the number of full filter steps per draw
    ms = (ctypes.c_float * 4)()
    at = torch.full((nloc,), -1, dtype=torch.int32, device=device)
    eng.record_steady_steps(at)
    local_eval(0, nloc, stage_ms=ms)
    torch.cuda.synchronize()
    eng.record_steady_steps(None)
    at_h = at.cpu().numpy()
    n_full = np.where(at_h < 0, T_len, at_h).astype(np.float64)
    full = None
    if world == 1 and not args.no_extras:
        ms_f = (ctypes.c_float * 4)()
        lp_h = logp_buf.clone()
        local_eval(0, nloc, stage_ms=ms_f, options={"kalman_steady_tol": 0.0})
        torch.cuda.synchronize()
        full = {"value": round(nloc / (sum(ms_f) * 1e-3), 2), "ms_per_step": round(float(sum(ms_f)), 3), "unit": "evals/s",
                "note": "kalman_steady_tol = 0: all T_len steps update the covariance",
                "max_rel_logp_diff_vs_headline": float((torch.abs(logp_buf - lp_h) / torch.abs(lp_h)).max().item())}
        local_eval(0, nloc)
        torch.cuda.synchronize()
    if rank == 0:
        mp16 = 16 * ((m + 15) // 16)
        kq = k + s * k + k * (k + 1) // 2
        # the prediction step's two products (unpadded): W' = P Az' in full, X = W Az' on and above the diagonal only (it is
        # symmetric: so_gemm_sym multiplies m (m + 1) / 2 of its m^2 entries)
        flops_step = 2.0 * m ** 3 + 2.0 * m * (m * (m + 1) / 2)
        flops_filter = float(n_full.sum()) * flops_step
        filt_s = ms[3] * 1e-3
        value = global_batch * args.steps / dt
        soc = so_counters(f"so_filter_kernel<{mp16 // 16}>", nloc)
        mfma_frac = flops_filter / filt_s / 1e12 / FP64_PEAK_TFLOPS
        hbm_frac = (soc["traffic"] / filt_s / 1e9 / HBM_PEAK_GBS) if soc.get("traffic") else None
        out = {
            "metric": "second-order solve + pruned-state-space Kalman-logp evals/sec, Smets-Wouters-shaped n=40 T=200",
            "value": round(value, 2), "unit": "evals/s", "n_gpus": world,
            **({"shared_device": f"{world} ranks on {n_dev} device(s), gloo gather: functional check, not a measurement"} if shared else {}),
            **({"gather": ginfo} if ginfo else {}),
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {
                "workload": (f"sw_second_order (BASELINE configs[4]): SW-shaped systems n={n}, n_state={s}, n_lead={len(Lc)}, k={k}, p={p}, "
                             f"T_len={T_len}, synthetic sparse model Hessian ({shard['hess_idx'].shape[0]} entries), pruned state "
                             f"2u + s(s+1)/2 = {m}; {per_gpu} draws per GPU"),
                "global_batch": global_batch, "solver": "cycle_reduction", "tol": args.tol,
                "kalman_steady_tol": __import__('geconpy_amd._lib', fromlist=['make_options']).make_options().kalman_steady_tol,
                "inputs": "A,B,C,D and the Hessian values resident in HBM",
                "parity_note": "the reference has no second-order solver (perturbation.py:97-98 raises): checked against "
                               "oracle/second_order.py, parity unpinned by construction",
                "parallelism": f"draw-sharded x{world}, one all_gather of packed (logp,status) records" if world > 1 else "single GPU",
            },
            "roofline": {
                "kernel": f"dsge::so_filter_kernel<{mp16 // 16}> (512 threads per draw; v_mfma_f64_16x16x4_f64 on {mp16 // 16} x {mp16 // 16} tiles)",
                # the bound is read off the counters: whichever of the two fractions is larger (committed HBM traffic over the
                # duration measured here vs useful matrix-core flops over the same duration)
                "bound": "hbm" if (hbm_frac is not None and hbm_frac > mfma_frac) else "mfma",
                "hbm_frac": round(hbm_frac, 4) if hbm_frac is not None else None, "mfma_frac": round(mfma_frac, 4),
                "unit": "TFLOP/s", "peak": FP64_PEAK_TFLOPS,
                "achieved": round(flops_filter / filt_s / 1e12, 3),
                "frac": round(flops_filter / filt_s / 1e12 / FP64_PEAK_TFLOPS, 5),
                "flops_source": f"2 m^3 (W' = P Az') + m^2 (m + 1) (the symmetric X = W Az', upper half only) with m = {m}, unpadded, per "
                                f"full filter step x the measured number of full steps per draw (mean {n_full.mean():.1f} of {T_len}); "
                                "duration: HIP events around the kernel in this run",
                **soc,
                "stage_ms": {"first_order_solver": round(ms[0], 3), "second_order_setup": round(ms[1], 3),
                             "stationary_covariance": round(ms[2], 3), "filter": round(ms[3], 3)},
                "full_steps_mean": float(n_full.mean()), "never_steady": int((at_h < 0).sum()),
                "qz_product_K": kq,
            },
            "failed_draws": n_fail,
        }
        if full is not None:
            out["full_recursion"] = full
        if cpu is not None:
            out["cpu_baseline"] = cpu
            ns = len(cpu_logp)
            rel = np.abs(logp_host[:ns] - cpu_logp) / np.abs(cpu_logp)
            out["parity"] = {"max_rel_logp_err_vs_cpu_oracle": float(rel.max()), "median_rel_logp_err_vs_cpu_oracle": float(np.median(rel)),
                             "n_checked": ns}
            out["gpu_over_cpu"] = round(value / cpu["value"], 1)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
