"""CPU (numpy): why so_gemm_sym mirrors the diagonal tiles too.  The covariance recursion of the pruned second-order filter for
SW-shaped draw 996 with P_new symmetrised in different ways: not at all / upper TILES mirrored (diagonal tiles left alone) /
averaged / upper or lower TRIANGLE mirrored.  Prints max |P_new - P| / max |P| at a few steps."""
import numpy as np, sys, scipy.linalg as sla
sys.path.insert(0,'/root/repo')
from geconpy_amd import workloads as wl
from oracle import second_order as so
from oracle.cycle_reduction import cycle_reduction_core
from oracle.shared import compute_selection_matrix
b = wl.sw_second_order_batch(1, first_draw=996); om = wl.sw_shaped_observation_model()
A,B,C,D = (b[x][0] for x in "ABCD"); Sigma = np.diag(b["sigma"][0]**2)
Tm, ok, _ = cycle_reduction_core(A,B,C,1000,1e-8); Rm = compute_selection_matrix(B,C,D,Tm)
S = np.flatnonzero((A!=0).any(axis=0))
sol = so.second_order_solution_reduced(B,C,Tm,Rm,b["hess_idx"],b["hess_val"][0],Sigma,S=S)
Z = om["Z"]; obs = np.flatnonzero((Z!=0).any(axis=0))
ps = so.pruned_state_space_reduced(Tm,Rm,sol,Sigma,obs)
Az, Qz, m = ps["Az"], ps["Qz"], ps["m"]
Za = so.pruned_design(Z, None, ps["U"], m); H = np.diag(om["Hdiag"]); jit = 1e-8
P0 = sla.solve_discrete_lyapunov(Az, Qz)
Qzj = Qz + jit*Az@Az.T
def run(mode, steps=200):
    P = P0.copy(); hist=[]
    for t in range(steps):
        PZ = P.T @ Za.T              # rows of P (device reads rows)
        F = Za @ PZ + H + jit*np.eye(Za.shape[0])
        K = np.linalg.solve(F.T, PZ.T).T
        V = PZ + jit*K
        AK = (Az@P)@Za.T @ np.linalg.inv(F); AV = (Az@P)@Za.T + jit*AK
        X = (Az @ P) @ Az.T
        Pn = X + Qzj - 0.5*(AK@AV.T + AV@AK.T)
        if mode == "mirror":
            for i in range(0, m, 16):
                for j in range(i+16, m, 16):
                    Pn[j:j+16, i:i+16] = Pn[i:i+16, j:j+16].T
        elif mode == "avg":
            Pn = 0.5*(Pn+Pn.T)
        d = np.abs(Pn-P).max()/np.abs(P).max(); hist.append(d)
        P = Pn
        if not np.isfinite(d): break
    return hist
for mode in ("none","mirror","avg"):
    h = run(mode)
    print(mode, ["%.1e"%h[i] for i in (10,30,50,80,120,160,min(199,len(h)-1))], "min eig last", )
def run2(mode, steps=200):
    P = P0.copy(); hist=[]; asym=[]
    for t in range(steps):
        PZ = P.T @ Za.T
        F = Za @ PZ + H + jit*np.eye(Za.shape[0])
        AK = (Az@P)@Za.T @ np.linalg.inv(F); AV = (Az@P)@Za.T + jit*AK
        X = (Az @ P) @ Az.T
        Pn = X + Qzj - 0.5*(AK@AV.T + AV@AK.T)
        asym.append(np.abs(Pn-Pn.T).max()/np.abs(Pn).max())
        if mode == "triu":
            Pn = np.triu(Pn) + np.triu(Pn,1).T
        elif mode == "tril":
            Pn = np.tril(Pn) + np.tril(Pn,-1).T
        elif mode == "consistent_mirror":   # use P (not P') everywhere
            pass
        d = np.abs(Pn-P).max()/np.abs(P).max(); hist.append(d); P = Pn
        if not np.isfinite(d): break
    return hist, asym
for mode in ("triu","tril"):
    h, a = run2(mode)
    print(mode, ["%.1e"%h[i] for i in (10,50,80,120,160,min(199,len(h)-1))], "asym of f(P) before mirroring", ["%.1e"%a[i] for i in (10,50,80,120,160)])
# is the asymmetry of f(P) for symmetric P really rounding-level?
Ps = 0.5*(P0+P0.T)
X = (Az@Ps)@Az.T; print("asym X for symmetric P:", np.abs(X-X.T).max()/np.abs(X).max())
F = Za@Ps@Za.T + H + jit*np.eye(7); print("asym F", np.abs(F-F.T).max())
