"""GPU box: the `gradient` leg's parity block of bench.py, draw by draw: <device gradient, direction> against the extrapolated central
differences of the CPU oracle (bench.gradient_fd_reference) for the 64 draws of bench.GRAD_FD_DRAWS."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from geconpy_amd import workloads as wl
from geconpy_amd.engine import LogpEngine


def main():
    nb = 4096
    shard = wl.sw_shaped_batch(nb); om = wl.sw_shaped_observation_model()
    fd = bench.gradient_fd_reference(shard, om, os.cpu_count() or 8)
    eng = LogpEngine(0)
    dev = [eng.to_device(shard[x]) for x in "ABCD"]
    dq = eng.to_device(shard["sigma"] ** 2); dZ = eng.to_device(om["Z"]); dy = eng.to_device(om["y"]); dH = eng.to_device(om["Hdiag"])
    hints = eng.structure_hints(dev[0], dZ)
    go = eng.solve_kalman_logp_grad(*dev, dq, dZ, dy, Hdiag=dH, tol=1e-8, max_iter=1000, n_filter_hint=hints[0])
    torch.cuda.synchronize()
    st = go["status"].cpu().numpy()
    errs = []
    for i, f in fd.items():
        dirs = bench._grad_directions(shard, i)
        an = sum(float((go[f"{k_}_bar"][i].cpu().numpy() * dirs[k_]).sum()) for k_ in ("A", "B", "C", "D", "q"))
        e = abs(an - f) / max(1.0, abs(f))
        errs.append(e)
        if not (e < 1e-7):
            print(f"draw {i}: status {st[i]} device {an:.10e} fd {f:.10e} rel err {e:.3e}")
    errs = np.array(errs)
    print("n", len(errs), "max", np.nanmax(errs), "median", np.nanmedian(errs), "nan", int(np.isnan(errs).sum()))


if __name__ == "__main__":  # (the FD reference uses a spawn pool: the module must be importable without side effects)
    main()
