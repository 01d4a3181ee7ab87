"""GPU box: every fuzz suite of tools/ at ten times the trial counts of tests/test_gpu_fuzz.py (seeds from the command line)."""
import sys, os, time, importlib
sys.path.insert(0, "tools")
plan = [("fuzz_fused", 4000), ("fuzz_cr", 3000), ("fuzz_kalman", 3000), ("fuzz_mixed", 300), ("fuzz_adjoints", 300),
        ("fuzz_gensys", 1500), ("fuzz_pencil", 1500), ("fuzz_hostpath", 300), ("fuzz_acf", 3000), ("fuzz_grad", 1200), ("fuzz_theta", 25)]
for seed in [int(a) for a in sys.argv[1:]] or [9001, 9002]:
    for name, n in plan:
        mod = importlib.import_module(name)
        t0 = time.time()
        try:
            bad = mod.run(seed, n, verbose=False)
        except Exception as e:
            bad = f"EXC {type(e).__name__}: {e}"
        print(f"{name} seed={seed} trials={n}: mismatches={bad}  ({time.time()-t0:.0f} s)", flush=True)
