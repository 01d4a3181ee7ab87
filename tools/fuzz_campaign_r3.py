"""GPU box: the round-3 additions at ten times the trial counts of tests/test_gpu_fuzz.py -- the gradient with random shock
covariance / design matrix modes, the second-order path, gensys with the real double-shift stage, the policy adjoints with
their second pass -- plus the fused evaluation and cycle reduction once more (seeds from the command line)."""
import sys, os, time, importlib
sys.path.insert(0, "tools")
plan = [("fuzz_grad", 400, dict(modes=True, rtol=1e-6)), ("fuzz_second_order", 120, {}), ("fuzz_gensys", 1500, {}),
        ("fuzz_adjoints", 300, {}), ("fuzz_fused", 4000, {}), ("fuzz_cr", 3000, {}), ("fuzz_theta", 25, {})]
for seed in [int(a) for a in sys.argv[1:]] or [3101, 3102]:
    for name, n, kw in plan:
        mod = importlib.import_module(name)
        t0 = time.time()
        try:
            bad = mod.run(seed, n, verbose=False, **kw)
        except Exception as e:
            bad = f"EXC {type(e).__name__}: {e}"
        print(f"{name} seed={seed} trials={n} {kw or ''}: mismatches={bad}  ({time.time()-t0:.0f} s)", flush=True)
