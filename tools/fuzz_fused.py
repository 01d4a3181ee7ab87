"""GPU box: randomized sizes through the fused evaluation (default options) against the oracle."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from geconpy_amd import batched, workloads as wl
import oracle
def run(seed, trials, verbose=True):
  """-> number of disagreements with the oracle over `trials` random configurations"""
  rng = np.random.default_rng(seed)
  bad = 0
  for trial in range(trials):
      n = int(rng.integers(6, 65))
      ns = int(rng.integers(1, max(2, n // 2)))
      nl = int(rng.integers(1, max(2, n // 3)))
      k = int(rng.integers(1, min(n, 24 if rng.random() < 0.2 else 12) + 1))
      p = int(rng.integers(1, min(n, 8) + 1)) if rng.random() < 0.3 else int(rng.integers(1, min(k, 8) + 1))  # (p > k: H > 0 keeps F regular)
      T_len = int(rng.choice([1, 2, 7, 40]))
      nb = 3
      try:
          sysm = [wl.sw_shaped_system(int(rng.integers(1 << 30)), n=n, n_state=ns, n_lead=nl, k=k) for _ in range(nb)]
      except Exception as e:
          continue
      A, B, C, D = (np.stack([s_[j] for s_ in sysm]) for j in range(4))
      q = rng.uniform(0.5, 2.0, (nb, k)) * 1e-4
      Z = np.zeros((p, n)); Z[np.arange(p), rng.choice(n, p, replace=False)] = rng.uniform(0.5, 1.5, p)
      y = rng.normal(0, 0.02, (T_len, p))
      if T_len > 2: y[1, 0] = np.nan
      H = rng.uniform(0.5, 2.0, p) * 1e-4
      d = rng.normal(0, 0.01, p)
      variant = int(rng.integers(0, 6))
      kw = dict(q_mode="diag_batched")
      Qor = [np.diag(q[i]) for i in range(nb)]
      if variant == 1 and n + nl <= 60:  # gensys (the on-chip pencil holds n + n_lead <= 62)
          kw["solver"] = "gensys"
      elif variant == 2 and p > 1:  # dense design matrix
          Z = rng.standard_normal((p, n)) * (rng.random((p, n)) < 0.4)
          Z[np.arange(p), rng.choice(n, p, replace=False)] += 1.0
      elif variant == 3:  # full shock covariance, shared
          L = rng.standard_normal((k, k)) * 0.01
          Qf = L @ L.T + 1e-5 * np.eye(k)
          q = Qf
          kw = dict(q_mode="full")
          Qor = [Qf] * nb
      elif variant == 4:  # policy outputs requested (full-size iteration contract)
          kw["return_policy"] = True
      elif variant == 5 and T_len > 4:  # a whole period missing, and a different mask later
          y[2, :] = np.nan
          y[T_len - 1, p - 1] = np.nan
      if rng.random() < 0.5:  # a random combination of the kernel-variant switches: every one must give the same answer
          kw["options"] = {name: int(rng.integers(0, 2)) for name in
                           ("cr_compact", "cr_fused_selection", "cr_deflation", "cr_two_waves", "cr_fused_deflation",
                            "cr_four_waves", "kalman_tiny", "kalman_nt_products") if rng.random() < 0.5}
          kw["options"]["kalman_order"] = int(rng.integers(0, 3))
          if rng.random() < 0.3:
              kw["options"]["kalman_steady_tol"] = 0.0
      try:
          out = batched.solve_kalman_logp_batched(A, B, C, D, q, Z, y, d=d, Hdiag=H, tol=1e-10, max_iter=1000, **kw)
      except Exception as e:
          print("EXC", variant, dict(n=n, ns=ns, nl=nl, k=k, p=p, T_len=T_len), repr(e)[:200])
          bad += 1
          continue
      for i in range(nb):
          r = oracle.solve_kalman_logp(A[i], B[i], C[i], D[i], Qor[i], Z, y, H=np.diag(H), d=d, tol=1e-10, max_iter=1000)
          ok_o = bool(r.get("converged", True)) and np.isfinite(r["logp"])
          ok_d = out["status"][i] == 0
          if ok_o != ok_d or (ok_o and abs(out["logp"][i] - r["logp"]) > 1e-8 * max(1.0, abs(r["logp"]))):
              bad += 1
              print("MISMATCH", "variant", variant, kw.get("options"), dict(n=n, ns=ns, nl=nl, k=k, p=p, T_len=T_len, draw=i), out["status"][i], out["logp"][i], r["logp"])
  if verbose:
    print("trials done, mismatches:", bad)
  return bad


if __name__ == "__main__":
    run(int(sys.argv[1]) if len(sys.argv) > 1 else 0, int(sys.argv[2]) if len(sys.argv) > 2 else 60)
