"""BASELINE.json configs on the device, as stated there (all through the C ABI):

* configs[1]  RBC, 4096 prior draws, batched *gensys* + Kalman logp (and cycle reduction) vs tests/golden/rbc_wide.npz
* configs[2]  SW-shaped, 512 distinct draws of the 4096-draw bench batch (incl. the ill-conditioned draws 752 and 2950)
              vs tests/golden/sw_shaped_wide.npz, both solvers; the reference's one absolute filter check (two
              representations of one model give the same logp at rtol 1e-7, tests/model/test_statespace.py:583-630)
* configs[3]  65 536 SW-shaped draws: on one device through size-independent properties, and the draw-sharded
              evaluator with the REAL engine on two ranks (RCCL when two devices are present, otherwise both ranks on
              device 0 with a gloo gather) + bench.py's own rank launcher.

The fixtures' expected values come from the reference's extracted solver bodies + the oracle filter
(tests/golden/make_wide_golden.py).  Tolerance: logp relative <= 1e-8 (BASELINE north_star).
"""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
from numpy.testing import assert_allclose

import oracle
from geconpy_amd import batched
from geconpy_amd import workloads as wl

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def load_golden(name):
    return np.load(os.path.join(ROOT, "tests", "golden", name))

LOGP_RTOL = 1e-8


def _sw_draws(idx):
    """Inputs of the listed draws (runs of consecutive indices generated together)."""
    parts = []
    idx = np.asarray(idx)
    start = 0
    for j in range(1, len(idx) + 1):
        if j == len(idx) or idx[j] != idx[j - 1] + 1:
            parts.append(wl.sw_shaped_batch(int(idx[j - 1] - idx[start] + 1), first_draw=int(idx[start])))
            start = j
    return {k: np.concatenate([p[k] for p in parts]) for k in ("A", "B", "C", "D", "sigma")}


@pytest.mark.parametrize("solver", ["cycle_reduction", "gensys"])
def test_config2_sw_shaped_512_distinct_draws(solver):
    g = load_golden("sw_shaped_wide.npz")
    idx = g["draw_idx"]
    assert len(np.unique(idx)) == len(idx) >= 512 and 752 in idx and 2950 in idx
    b = _sw_draws(idx)
    assert_allclose([np.abs(b[x][:8]).sum() for x in "ABCD"], g["input_checksum_first8"], rtol=1e-13)
    om = wl.sw_shaped_observation_model()
    r = batched.solve_kalman_logp_batched(b["A"], b["B"], b["C"], b["D"], b["sigma"] ** 2, om["Z"], om["y"],
                                          Hdiag=om["Hdiag"], tol=1e-8, max_iter=1000, solver=solver)
    assert np.all(r["status"] == 0)
    ref = g["ref_cr_logp"] if solver == "cycle_reduction" else g["ref_gensys_logp"]
    rel = np.abs(r["logp"] - ref) / np.abs(ref)
    assert rel.max() <= LOGP_RTOL, (int(idx[rel.argmax()]), rel.max())
    assert np.median(rel) < 1e-13
    # the other solver's reference value is the same number to the reference's own cross-solver tolerance
    other = g["ref_gensys_logp"] if solver == "cycle_reduction" else g["ref_cr_logp"]
    assert np.max(np.abs(r["logp"] - other) / np.abs(other)) <= LOGP_RTOL
    if solver == "cycle_reduction":
        _T, st, it = batched.cycle_reduction_batched(b["A"], b["B"], b["C"], max_iter=1000, tol=1e-8)
        assert np.all(st == 0) and np.array_equal(it, g["ref_cr_iters"])


@pytest.mark.parametrize("narrow", [1, 0])
def test_config2_observed_jump_variables_512_distinct_draws(narrow):
    """The realistic observation structure (bench.py leg `observe_jumps`): Z selects seven NON-state variables
    (`_make_design_matrix` allows any, statespace.py:260-332), so the filter keeps 18 states + 7 observed jumps = 25 variables and
    runs on the 32-wide tile.  512 distinct draws against tests/golden/sw_shaped_wide_jumps.npz (reference cycle reduction +
    oracle filter, make_jumps_golden.py), with the narrow-row instance of the filter (dsge_options.kalman_narrow, default) and
    with the generic one: same logp to the bit."""
    g = load_golden("sw_shaped_wide_jumps.npz")
    idx = g["draw_idx"]
    b = _sw_draws(idx)
    om = wl.sw_shaped_observation_model(observed=tuple(int(v) for v in g["observed"]))
    assert_allclose(np.abs(om["y"]).sum(), g["y_checksum"], rtol=1e-13)
    r = batched.solve_kalman_logp_batched(b["A"], b["B"], b["C"], b["D"], b["sigma"] ** 2, om["Z"], om["y"],
                                          Hdiag=om["Hdiag"], tol=1e-8, max_iter=1000, options={"kalman_narrow": narrow})
    assert np.all(r["status"] == 0)
    rel = np.abs(r["logp"] - g["ref_cr_logp"]) / np.abs(g["ref_cr_logp"])
    assert rel.max() <= LOGP_RTOL, (int(idx[rel.argmax()]), rel.max())
    assert np.median(rel) < 1e-13
    if narrow == 0:
        r1 = batched.solve_kalman_logp_batched(b["A"], b["B"], b["C"], b["D"], b["sigma"] ** 2, om["Z"], om["y"],
                                               Hdiag=om["Hdiag"], tol=1e-8, max_iter=1000, options={"kalman_narrow": 1})
        assert np.array_equal(r1["logp"], r["logp"])


@pytest.mark.parametrize("solver", ["gensys", "cycle_reduction"])
def test_config1_rbc_4096_draws(solver):
    g = load_golden("rbc_wide.npz")
    b, om = wl.rbc_batch(4096)
    assert_allclose([np.abs(b[x][:8]).sum() for x in "ABCD"], g["input_checksum_first8"], rtol=1e-13)
    r = batched.solve_kalman_logp_batched(b["A"], b["B"], b["C"], b["D"], b["sigma"] ** 2, om["Z"], om["y"],
                                          Hdiag=om["Hdiag"], tol=1e-8, max_iter=1000, solver=solver,
                                          q_mode="diag_batched")
    assert np.all(r["status"] == 0) and np.all(np.isfinite(r["logp"]))
    idx = g["draw_idx"]
    ref = g["ref_gensys_logp"] if solver == "gensys" else g["ref_cr_logp"]
    assert_allclose(r["logp"][idx], ref, rtol=LOGP_RTOL)
    # the whole batch again in two halves: a draw's value does not depend on where it sits or on the batch size
    lo = batched.solve_kalman_logp_batched(b["A"][:2048], b["B"][:2048], b["C"][:2048], b["D"][:2048],
                                           b["sigma"][:2048] ** 2, om["Z"], om["y"], Hdiag=om["Hdiag"], tol=1e-8,
                                           max_iter=1000, solver=solver, q_mode="diag_batched")
    assert np.array_equal(lo["logp"], r["logp"][:2048])


def test_two_representations_same_logp_on_device():
    """tests/model/test_statespace.py:583-630: the same model written two ways (here: the variables in two different
    solver orders, observed series permuted accordingly) must give the same logp at rtol 1e-7 -- on the device, not
    only in the oracle."""
    b = wl.sw_shaped_batch(32)
    om = wl.sw_shaped_observation_model()
    n = b["A"].shape[1]
    perm = np.random.default_rng(3).permutation(n)
    eqp = np.random.default_rng(4).permutation(n)
    A2, B2, C2 = (b[x][:, eqp][:, :, perm] for x in "ABC")
    D2 = b["D"][:, eqp]
    Z2 = om["Z"][:, perm]
    for solver in ("cycle_reduction", "gensys"):
        r1 = batched.solve_kalman_logp_batched(b["A"], b["B"], b["C"], b["D"], b["sigma"] ** 2, om["Z"], om["y"],
                                               Hdiag=om["Hdiag"], tol=1e-8, max_iter=1000, solver=solver)
        r2 = batched.solve_kalman_logp_batched(np.ascontiguousarray(A2), np.ascontiguousarray(B2),
                                               np.ascontiguousarray(C2), np.ascontiguousarray(D2), b["sigma"] ** 2, Z2,
                                               om["y"], Hdiag=om["Hdiag"], tol=1e-8, max_iter=1000, solver=solver)
        assert np.all(r1["status"] == 0) and np.all(r2["status"] == 0)
        assert_allclose(r2["logp"], r1["logp"], rtol=1e-7)


def test_config3_65536_draws_one_device():
    """The multi-GPU config's total on ONE device, device-resident: status all-clear, a repeat run is bit-identical,
    every 4096-draw slice evaluated on its own gives the same bits, and a sample matches the oracle."""
    import torch

    from geconpy_amd.engine import LogpEngine

    nb, chunk = 65536, 8192
    eng = LogpEngine(0)
    om = wl.sw_shaped_observation_model()
    dev = {x: torch.empty((nb, 40, 7 if x == "D" else 40), dtype=torch.float64, device="cuda") for x in "ABCD"}
    dq = torch.empty((nb, 7), dtype=torch.float64, device="cuda")
    keep = {}
    for c0 in range(0, nb, chunk):
        part = wl.sw_shaped_batch(chunk, first_draw=c0)
        for x in "ABCD":
            dev[x][c0:c0 + chunk].copy_(torch.from_numpy(part[x]))
        dq[c0:c0 + chunk].copy_(torch.from_numpy(part["sigma"] ** 2))
        for i in (c0, c0 + chunk - 1):
            keep[i] = {x: part[x][i - c0].copy() for x in "ABCD"} | {"q": part["sigma"][i - c0] ** 2}
    dZ, dy, dH = eng.to_device(om["Z"]), eng.to_device(om["y"]), eng.to_device(om["Hdiag"])
    ns, zs = eng.structure_hints(dev["A"][:64], dZ)
    kw = dict(Hdiag=dH, q_mode=1, tol=1e-8, max_iter=1000, n_state_hint=ns, z_selector_hint=zs)
    logp, status = eng.solve_kalman_logp(dev["A"], dev["B"], dev["C"], dev["D"], dq, dZ, dy, **kw)
    torch.cuda.synchronize()
    l1, s1 = logp.cpu().numpy().copy(), status.cpu().numpy().copy()
    assert np.all(s1 == 0) and np.all(np.isfinite(l1))
    logp2, _ = eng.solve_kalman_logp(dev["A"], dev["B"], dev["C"], dev["D"], dq, dZ, dy, **kw)
    torch.cuda.synchronize()
    assert np.array_equal(logp2.cpu().numpy(), l1)
    for c0 in (0, 4096, 61440):
        lp, _ = eng.solve_kalman_logp(*(dev[x][c0:c0 + 4096] for x in "ABCD"), dq[c0:c0 + 4096], dZ, dy, **kw)
        torch.cuda.synchronize()
        assert np.array_equal(lp.cpu().numpy(), l1[c0:c0 + 4096])
    for i, d in keep.items():
        ref = oracle.solve_kalman_logp(d["A"], d["B"], d["C"], d["D"], np.diag(d["q"]), om["Z"], om["y"],
                                       H=np.diag(om["Hdiag"]), tol=1e-8, max_iter=1000)
        assert abs(l1[i] - ref["logp"]) <= LOGP_RTOL * abs(ref["logp"]), i


# ---- draw-sharded evaluation with the real engine ---------------------------------------------------------------------
_RANK_SCRIPT = r"""
import os, sys, json
import numpy as np
sys.path.insert(0, {root!r})
import torch, torch.distributed as dist
from geconpy_amd import workloads as wl
from geconpy_amd.engine import LogpEngine, ShardedLogpEvaluator
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
n_dev = torch.cuda.device_count()
dev_index = rank % n_dev
torch.cuda.set_device(dev_index)
device = torch.device("cuda", dev_index)
if n_dev >= world:
    dist.init_process_group("nccl", device_id=device)
else:
    dist.init_process_group("gloo")
global_batch = int(sys.argv[1])
lo, hi = wl.shard_bounds(global_batch, world, rank)
eng = LogpEngine(device)
b = wl.sw_shaped_batch(hi - lo, first_draw=lo)
om = wl.sw_shaped_observation_model()
dA, dB, dC, dD = (eng.to_device(b[x]) for x in "ABCD")
dq = eng.to_device(b["sigma"] ** 2)
dZ, dy, dH = eng.to_device(om["Z"]), eng.to_device(om["y"]), eng.to_device(om["Hdiag"])
ns, zs = eng.structure_hints(dA, dZ)
def local_eval(lo_, hi_):
    return eng.solve_kalman_logp(dA, dB, dC, dD, dq, dZ, dy, Hdiag=dH, q_mode=1, tol=1e-8, max_iter=1000,
                                 n_state_hint=ns, z_selector_hint=zs)
ev = ShardedLogpEvaluator(global_batch, local_eval, device)
for _ in range(2):
    logp, status = ev.step()
torch.cuda.synchronize()
np.savez(sys.argv[2] + f".rank{{rank}}.npz", logp=logp.cpu().numpy(), status=status.cpu().numpy(), lo=lo, hi=hi,
         backend=dist.get_backend())
dist.barrier()
dist.destroy_process_group()
"""


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("global_batch", [256, 203])
def test_sharded_evaluator_real_engine_two_ranks(tmp_path, global_batch):
    """ShardedLogpEvaluator around LogpEngine.solve_kalman_logp on two rank processes: every rank ends up with all
    draws in draw order, bit-identical to the single-process evaluation of the same global batch."""
    import torch

    world = 2
    script = tmp_path / "rank.py"
    script.write_text(_RANK_SCRIPT.format(root=ROOT))
    port = _free_port()
    out = str(tmp_path / "out")
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, str(script), str(global_batch), out], env=env))
    for p in procs:
        assert p.wait(timeout=600) == 0
    b = wl.sw_shaped_batch(global_batch)
    om = wl.sw_shaped_observation_model()
    single = batched.solve_kalman_logp_batched(b["A"], b["B"], b["C"], b["D"], b["sigma"] ** 2, om["Z"], om["y"],
                                               Hdiag=om["Hdiag"], tol=1e-8, max_iter=1000)
    for r in range(world):
        g = np.load(out + f".rank{r}.npz")
        assert str(g["backend"]) == ("nccl" if torch.cuda.device_count() >= world else "gloo")
        assert g["logp"].shape == (global_batch,)
        assert np.array_equal(g["logp"], single["logp"])  # bit-exact draw indexing on every rank
        assert np.array_equal(g["status"], single["status"])


def test_bench_launches_its_own_ranks():
    """`python bench.py --gpus 2` without torch.distributed.run: the launcher starts two rank processes (before anything
    touches the GPU) and rank 0 prints one JSON line with n_gpus = 2 -- a COMPLETE line: roofline, cpu_baseline and parity
    (rank 0 computes them on its shard before it touches the GPU), so that a multi-GPU driver run counts as measured."""
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
           "--batch-per-gpu", "512", "--cpu-sample", "8", "--profile-reps", "1", "--allow-shared-gpu"]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=1500, env=env)
    assert res.returncode == 0, res.stderr[-2000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["config"]["global_batch"] == 1024 and out["failed_draws"] == 0
    assert out["value"] > 0 and out["scaling"] == "weak"
    assert out["roofline"]["achieved"] > 0 and out["roofline"]["peak"] > 0 and "frac" in out["roofline"]
    assert out["cpu_baseline"]["value"] > 0 and out["cpu_baseline"]["cores"] >= 1 and out["cpu_baseline"]["kind"] == "port"
    assert out["parity"]["n_checked"] >= 8 and out["parity"]["max_rel_logp_err_vs_cpu_oracle"] <= 1e-8
    # who took part in the gather (round 6): backend, world and one entry per rank -- here two ranks on ONE device over gloo, and the
    # line says exactly that (on a multi-GPU node: "nccl", distinct_devices == world)
    g = out["gather"]
    assert g["world"] == 2 and g["backend"] == "gloo" and [e["rank"] for e in g["ranks"]] == [0, 1] and g["distinct_devices"] == 1


def test_bench_sizes_a_multi_gpu_run_as_configs3():
    """Without --batch-per-gpu a run on N > 1 GPUs is BASELINE configs[3]'s share: 8192 draws per GPU (65 536 over 8), and the
    line says so (argument handling only: the launcher is asked for more ranks than devices and refuses before any work)."""
    import importlib.util

    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    src = open(os.path.join(ROOT, "bench.py")).read()
    assert "4096 if world == 1 else 8192" in src and "BASELINE configs[3]" in src
    assert hasattr(mod, "main_second_order") and hasattr(mod, "spawn_ranks")


@pytest.mark.parametrize("solver", ["cycle_reduction", "gensys"])
def test_a_draw_gives_the_same_bits_in_a_batch_of_any_size(solver):
    """The batch size only changes the dispatch (ordering kernel, grid sizes, second passes): the first nb draws of a
    4097-draw batch evaluated as a batch of nb = 1, 63, 65, 1023, 1025, 2049 draws give bit-identical logp, with and
    without the structure hints; the gradient path likewise (cycle reduction)."""
    import torch

    from geconpy_amd.engine import LogpEngine

    eng = LogpEngine(0)
    b = wl.sw_shaped_batch(130)
    om = wl.sw_shaped_observation_model()
    total = 4097
    rep = (total + 129) // 130
    dev = [eng.to_device(np.tile(b[x], (rep, 1, 1))[:total]) for x in "ABCD"]
    dq = eng.to_device(np.tile(b["sigma"] ** 2, (rep, 1))[:total])
    dZ, dy, dH = eng.to_device(om["Z"]), eng.to_device(om["y"]), eng.to_device(om["Hdiag"])
    ns, zs = eng.structure_hints(dev[0][:64], dZ)
    for hinted in (True, False):
        kw = dict(Hdiag=dH, q_mode=1, tol=1e-8, max_iter=1000, solver=solver)
        if hinted:
            kw.update(n_state_hint=ns, z_selector_hint=zs)
        full, st = eng.solve_kalman_logp(*dev, dq, dZ, dy, **kw)
        torch.cuda.synchronize()
        full = full.cpu().numpy().copy()
        assert int((st != 0).sum()) == 0
        for nb in (1, 63, 65, 1023, 1025, 2049):
            lp, st2 = eng.solve_kalman_logp(*(x[:nb] for x in dev), dq[:nb], dZ, dy, **kw)
            torch.cuda.synchronize()
            assert int((st2 != 0).sum()) == 0
            assert np.array_equal(lp.cpu().numpy(), full[:nb]), (solver, hinted, nb)
    if solver == "cycle_reduction":
        g_full = eng.solve_kalman_logp_grad(*dev, dq, dZ, dy, Hdiag=dH, tol=1e-8, max_iter=1000)
        torch.cuda.synchronize()
        g_full = {k_: v.cpu().numpy().copy() for k_, v in g_full.items() if hasattr(v, "cpu")}
        for nb in (1, 65, 1025):
            g = eng.solve_kalman_logp_grad(*(x[:nb] for x in dev), dq[:nb], dZ, dy, Hdiag=dH, tol=1e-8, max_iter=1000)
            torch.cuda.synchronize()
            for k_, v in g.items():
                if hasattr(v, "cpu"):
                    assert np.array_equal(v.cpu().numpy(), g_full[k_][:nb]), (k_, nb)
