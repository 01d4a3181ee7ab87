"""How far apart are two float64 cycle reductions on an ill-conditioned system?  (fuzz_cr seed 3101 found one: 54 variables,
cond(A1) = 2..5e6 in every iteration, |C| = 3e5; the reference's LAPACK path is 1.8e-8 from the 40-digit T there, the device
2.4e-8 / 5.0e-8.)  Twenty rounding-level perturbations of that system (relative 1e-12: other systems with the same
conditioning), device T on the GPU box, oracle and 40-digit T in the container.

  GPU box:    python tools/cr_accuracy_study.py device tests/golden/cr_ill_conditioned_54.npz gpurun_out/cr_accuracy_device.npz
  container:  python tools/cr_accuracy_study.py judge gpurun_out/cr_accuracy_device.npz
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np


def systems(path, count=20):
    d = np.load(path)
    rng = np.random.default_rng(5)
    out = []
    for j in range(count):
        pert = lambda M: M * (1.0 + (1e-12 * rng.standard_normal(M.shape) if j else 0.0))
        out.append((pert(d["A"]), pert(d["B"]), pert(d["C"])))
    return out, float(d["tol"])


if sys.argv[1] == "device":
    from geconpy_amd import batched

    sysm, tol = systems(sys.argv[2])
    A, B, C = (np.stack([s[j] for s in sysm]) for j in range(3))
    res = {}
    for name, opts in (("default", {}), ("one_wave", {"cr_four_waves": 0})):
        T, st, it = batched.cycle_reduction_batched(A, B, C, max_iter=200, tol=tol, options=opts)
        res["T_" + name], res["it_" + name] = T, it
        print(name, "status", st.tolist(), "iterations", it.tolist())
    np.savez(sys.argv[3], A=A, B=B, C=C, tol=tol, **res)
else:
    import oracle
    from fuzz_cr import cycle_reduction_exact

    d = np.load(sys.argv[2])
    tol = float(d["tol"])
    rows = []
    for j in range(d["A"].shape[0]):
        Tc, conv, itc = oracle.cycle_reduction_core(d["A"][j], d["B"][j], d["C"][j], 200, tol)
        Tx, itx = cycle_reduction_exact(d["A"][j], d["B"][j], d["C"][j], tol)
        e_orc = np.abs(Tc - Tx).max()
        e_def, e_one = np.abs(d["T_default"][j] - Tx).max(), np.abs(d["T_one_wave"][j] - Tx).max()
        rows.append((e_orc, e_def, e_one))
        print(f"system {j}: iterations oracle {itc} exact {itx} device {int(d['it_default'][j])}/{int(d['it_one_wave'][j])};  |T - T_exact|: "
              f"oracle {e_orc:.2e}, device (four wavefronts) {e_def:.2e}, device (one wavefront) {e_one:.2e}", flush=True)
    r = np.array(rows)
    print("median |T - T_exact|: oracle %.2e, four wavefronts %.2e, one wavefront %.2e" % tuple(np.median(r, axis=0)))
    print("max:                  oracle %.2e, four wavefronts %.2e, one wavefront %.2e" % tuple(r.max(axis=0)))
    print("ratio device / oracle per system: four wavefronts min %.2f median %.2f max %.2f; one wavefront min %.2f median %.2f max %.2f"
          % (*(f(r[:, 1] / r[:, 0]) for f in (np.min, np.median, np.max)), *(f(r[:, 2] / r[:, 0]) for f in (np.min, np.median, np.max))))
