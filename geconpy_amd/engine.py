"""Device-resident evaluator: torch owns HBM buffers and streams, libdsge_hip does the work.

PyTorch is plumbing here (device memory, the current HIP stream, ``torch.distributed``);
every FLOP of the path runs in the hand-written kernels behind the C ABI.
"""
from __future__ import annotations

import numpy as np

from . import _lib
from .batched import JITTER_DEFAULT, MISSING_FILL
from .workloads import shard_bounds


def _torch():
    import torch

    return torch


class LogpEngine:
    """Fused ``A,B,C,D -> logp`` on one GPU with inputs and outputs resident in HBM.

    All tensor arguments are float64 CUDA tensors on ``device`` (contiguous, row-major,
    leading draw axis); work is enqueued on torch's current stream for that device.
    """

    def __init__(self, device=0):
        torch = _torch()
        if not torch.cuda.is_available():
            raise _lib.DsgeHipError("LogpEngine needs a GPU: torch.cuda.is_available() is False (no CPU fallback)")
        self.torch = torch
        self.device = torch.device("cuda", device) if not isinstance(device, torch.device) else device
        self.lib = _lib.load()
        torch.cuda.set_device(self.device)
        _lib.check(self.lib.dsge_set_device(self.device.index or 0))

    # -- helpers ---------------------------------------------------------------------------
    def to_device(self, x, dtype=None):
        torch = self.torch
        if isinstance(x, torch.Tensor):
            return x.to(self.device, dtype or torch.float64).contiguous()
        return torch.as_tensor(np.ascontiguousarray(x), dtype=dtype or torch.float64, device=self.device)

    def _chk(self, t, shape=None):
        torch = self.torch
        if t is None:
            return None
        if not (isinstance(t, torch.Tensor) and t.is_cuda and t.dtype == torch.float64 and t.is_contiguous()):
            raise ValueError("expected a contiguous float64 CUDA tensor")
        if shape is not None and tuple(t.shape) != tuple(shape):
            raise ValueError(f"expected shape {shape}, got {tuple(t.shape)}")
        return t

    @staticmethod
    def _p(t):
        return None if t is None else t.data_ptr()

    def _stream(self):
        return self.torch.cuda.current_stream(self.device).cuda_stream

    def _pack(self, A, B, C, D, Q, Z, y, d, Hdiag, q_mode):
        nb, n, _ = A.shape
        k = D.shape[2]
        T_len, p = y.shape
        for t in (A, B, C):
            self._chk(t, (nb, n, n))
        self._chk(D, (nb, n, k))
        self._chk(y, (T_len, p))
        self._chk(Q)
        if q_mode is None:
            q_mode = {(k,): 0, (nb, k, k): 3}.get(tuple(Q.shape))
            if q_mode is None:
                if tuple(Q.shape) == (nb, k) and nb != k:
                    q_mode = 1
                elif tuple(Q.shape) == (k, k) and nb != k:
                    q_mode = 2
                else:
                    raise ValueError("ambiguous Q layout; pass q_mode")
        zb = int(self._chk(Z).dim() == 3)
        db = int(d is not None and self._chk(d).dim() == 2)
        hb = int(Hdiag is not None and self._chk(Hdiag).dim() == 2)
        return nb, n, k, p, T_len, int(q_mode), zb, db, hb

    def structure_hints(self, A, Z):
        """(n_state_hint, z_selector_hint) from device tensors: one small reduction + host sync;
        the structure is a property of the model, so call this once, not per step."""
        torch = self.torch
        n_state = int(torch.count_nonzero((A != 0).reshape(-1, A.shape[-1]).any(dim=0)).item())
        nz = (Z != 0).reshape(-1, Z.shape[-2], Z.shape[-1])
        sel = bool(((nz.sum(dim=2) == 1).all() & (nz.sum(dim=1) <= 1).all()).item())
        return n_state, int(sel)

    def static_hint(self, A, C):
        """``dsge_options.n_static_hint`` from device tensors: variables whose columns of A and C are exactly zero in
        every draw (one reduction + host sync; a property of the model: call once).  With it the fused call is a pure
        enqueue -- no measuring launch, no read-back inside the library."""
        torch = self.torch
        n = A.shape[-1]
        nz = (A != 0).reshape(-1, n).any(dim=0) | (C != 0).reshape(-1, n).any(dim=0)
        return int(n - torch.count_nonzero(nz).item())

    def record_steady_steps(self, buf):
        """Debug: ``buf`` (int32 CUDA tensor [batch]) receives, from the fast-path Kalman launches that
        follow, the first time step each draw ran in steady-state mode (-1 = never); ``None`` stops."""
        if buf is not None and not (buf.is_cuda and buf.dtype == self.torch.int32 and buf.is_contiguous()):
            raise ValueError("expected a contiguous int32 CUDA tensor")
        _lib.check(self.lib.dsge_debug_kalman_steady_steps(None if buf is None else buf.data_ptr()))

    # -- on-device Jacobians ---------------------------------------------------
    def jacobians_from_theta(self, program, theta, out=None):
        """``theta`` (float64 CUDA tensor [batch][npar]) -> (A, B, C, D, q) resident in HBM, evaluated by
        the generated kernel of ``program`` (geconpy_amd.jacobian_codegen.JacobianProgram) on the current
        stream; ``q`` is None when the program carries no shock variances.  ``out`` may hold preallocated
        tensors (A, B, C, D, q) to reuse across MCMC steps."""
        torch = self.torch
        self._chk(theta)
        nb, npar = theta.shape
        if npar != len(program.params):
            raise ValueError(f"theta has {npar} columns, the program has {len(program.params)} parameters")
        n, k = program.n, program.k
        if out is None:
            mk = lambda *shape: torch.empty(shape, dtype=torch.float64, device=self.device)  # noqa: E731
            out = (mk(nb, n, n), mk(nb, n, n), mk(nb, n, n), mk(nb, n, k), mk(nb, k) if program.q is not None else None)
        A, B, C, D, q = out
        program.launch(theta.data_ptr(), nb, A.data_ptr(), B.data_ptr(), C.data_ptr(), D.data_ptr(),
                       None if q is None else q.data_ptr(), self._stream())
        return A, B, C, D, q

    def observation_from_theta(self, program, theta, out=None):
        """``theta`` -> (Z [batch][p][n] or None, d [batch][p] or None): the parameter-dependent observation equation of a
        program built with ``Z=`` / ``d=`` (statespace.py:298-388), by its second generated kernel."""
        torch = self.torch
        self._chk(theta)
        nb = theta.shape[0]
        if program.Z is None and program.d is None:
            return None, None
        if out is None:
            mk = lambda *shape: torch.empty(shape, dtype=torch.float64, device=self.device)  # noqa: E731
            out = (mk(nb, program.p, program.n) if program.Z is not None else None,
                   mk(nb, program.p) if program.d is not None else None)
        Zb, db = out
        program.launch_obs(theta.data_ptr(), nb, self._p(Zb), self._p(db), self._stream())
        return Zb, db

    def logp_from_theta(self, program, theta, Z, y, d=None, Hdiag=None, jac_out=None, **kw):
        """theta -> A,B,C,D,q -> T,R -> P0 -> logp without the matrices ever leaving the device: the generated
        Jacobian kernel followed by the fused pipeline on the same stream.  Returns (logp, status).  ``Z`` / ``d`` = None
        take the program's own parameter-dependent observation equation (``observation_from_theta``)."""
        A, B, C, D, q = self.jacobians_from_theta(program, theta, out=jac_out)
        if q is None:
            raise ValueError("the program has no shock variances; call jacobians_from_theta + solve_kalman_logp")
        if Z is None or (d is None and getattr(program, "d", None) is not None):
            Zg, dg = self.observation_from_theta(program, theta)
            Z = Zg if Z is None else Z
            d = dg if d is None else d
            if Z is None:
                raise ValueError("no design matrix: pass Z or build the program with Z=")
        return self.solve_kalman_logp(A, B, C, D, q, Z, y, d=d, Hdiag=Hdiag, q_mode=1, **kw)

    def logp_and_grad_from_theta(self, program, theta, Z, y, d=None, Hdiag=None, jac_out=None, grad_out=None,
                                 theta_bar=None, **kw):
        """theta -> (logp, status, d logp / d theta) entirely on the device: generated Jacobian kernel, the fused
        logp + reverse-mode pipeline, and the generated pullback kernel theta_bar = J' (A_bar, B_bar, C_bar, D_bar,
        q_bar) -- the interface a gradient-based sampler (NUTS) needs.  Cotangents of d / Hdiag stay available in the
        returned dict (``grad``)."""
        torch = self.torch
        A, B, C, D, q = self.jacobians_from_theta(program, theta, out=jac_out)
        if q is None:
            raise ValueError("the program has no shock variances")
        # the observation equation: the same resolution as logp_from_theta, so that the pair (logp, gradient) belongs to the
        # SAME function of theta -- a program built with d= filters with its parameter-dependent intercept and the cotangent
        # of d flows back through the generated pullback; a parameter-dependent Z(theta) takes the dense-Z gradient entry
        # point, whose Z_bar goes through the generated pullback of Z
        z_from_program = Z is None and getattr(program, "Z", None) is not None
        d_from_program = d is None and getattr(program, "d", None) is not None
        if z_from_program or d_from_program:
            Zp, dp = self.observation_from_theta(program, theta)
            if z_from_program:
                Z = Zp
            if d_from_program:
                d = dp
        if Z is None:
            raise ValueError("no design matrix: pass Z")
        if z_from_program:
            kw = dict(kw, dense_z=True)
        g = self.solve_kalman_logp_grad(A, B, C, D, q, Z, y, d=d, Hdiag=Hdiag, out=grad_out, **kw)
        if theta_bar is None:
            theta_bar = torch.empty_like(theta)
        program.launch_vjp(theta.data_ptr(), theta.shape[0], g["A_bar"].data_ptr(), g["B_bar"].data_ptr(),
                           g["C_bar"].data_ptr(), g["D_bar"].data_ptr(), g["q_bar"].data_ptr(), theta_bar.data_ptr(),
                           self._stream())
        if d_from_program:  # theta_bar += (d d / d theta)' d_bar
            program.launch_obs_vjp(theta.data_ptr(), theta.shape[0], g["d_bar"].data_ptr(), theta_bar.data_ptr(), self._stream())
        if z_from_program:  # theta_bar += (d Z / d theta)' Z_bar
            program.launch_obs_z_vjp(theta.data_ptr(), theta.shape[0], g["Z_bar"].data_ptr(), theta_bar.data_ptr(), self._stream())
        return g["logp"], g["status"], theta_bar, g

    # -- product entry points --------------------------------------------------------------
    def solve_kalman_logp(self, A, B, C, D, Q, Z, y, d=None, Hdiag=None, q_mode=None, solver="cycle_reduction",
                          tol=1e-6, max_iter=50, jitter=JITTER_DEFAULT, missing_fill_value=MISSING_FILL,
                          logp=None, status=None, n_state_hint=0, z_selector_hint=0, n_lead_hint=0, T_out=None,
                          R_out=None, options=None):
        """Enqueue one fused evaluation of the whole batch; returns (logp, status) tensors
        (asynchronous: synchronize the stream before reading them on the host).  ``T_out`` [batch][n][n] /
        ``R_out`` [batch][n][k]: optional float64 CUDA tensors that receive the policy matrices."""
        torch = self.torch
        nb, n, k, p, T_len, qm, zb, db, hb = self._pack(A, B, C, D, Q, Z, y, d, Hdiag, q_mode)
        if logp is None:
            logp = torch.empty(nb, dtype=torch.float64, device=self.device)
        if status is None:
            status = torch.empty(nb, dtype=torch.int32, device=self.device)
        op, _keep = _lib.opt_ptr(options)
        _lib.check(
            self.lib.dsge_solve_kalman_logp_batched_opt(
                op, self._p(A), self._p(B), self._p(C), self._p(D), self._p(Q), qm, self._p(Z), zb, self._p(d), db,
                self._p(Hdiag), hb, self._p(y), nb, n, k, p, T_len, _lib.SOLVER_CODES[solver], float(tol),
                int(max_iter), float(jitter), float(missing_fill_value), int(n_state_hint), int(z_selector_hint),
                int(n_lead_hint), self._p(logp), status.data_ptr(), self._p(T_out), self._p(R_out), None, None,
                self._stream(),
            )
        )
        return logp, status

    def solve_kalman_logp_grad(self, A, B, C, D, q, Z, y, d=None, Hdiag=None, solver="cycle_reduction", tol=1e-6, max_iter=50,
                               jitter=JITTER_DEFAULT, missing_fill_value=MISSING_FILL, n_filter_hint=0, n_lead_hint=0,
                               out=None, options=None, full_covariance=False, dense_z=False):
        """logp and its reverse-mode gradient for the whole batch, device-resident (dsge_solve_kalman_logp_grad_batched).
        ``q``: (k,) or (batch, k) diagonal shock variances; with ``full_covariance=True`` a full symmetric Q, (k, k) or
        (batch, k, k) (statespace.py:247-251), and ``q_bar`` is (batch, k, k).  Returns a dict of tensors: logp, status, A_bar,
        B_bar, C_bar, D_bar, q_bar[, d_bar][, h_bar] (asynchronous).  ``out`` may carry the same dict from an earlier call to
        reuse the buffers.  ``dense_z=True``: any design matrix (observation equations), through
        ``dsge_solve_kalman_logp_grad_dense_z_batched`` (n + p <= 56); the dict then also holds ``Z_bar`` (batch, p, n) and
        ``n_filter_hint`` counts the state variables (non-zero columns of A)."""
        torch = self.torch
        nb, n, _ = A.shape
        k = D.shape[2]
        T_len, p = y.shape
        for t in (A, B, C):
            self._chk(t, (nb, n, n))
        self._chk(D, (nb, n, k))
        self._chk(y, (T_len, p))
        self._chk(q)
        if full_covariance:
            if tuple(q.shape) not in ((k, k), (nb, k, k)):
                raise ValueError("Q must be (k, k) or (batch, k, k)")
            q_mode = 2 + int(q.dim() == 3)
        else:
            if tuple(q.shape) not in ((k,), (nb, k)):
                raise ValueError("q must be (k,) or (batch, k)")
            q_mode = int(q.dim() == 2)
        zb = int(self._chk(Z).dim() == 3)
        db = int(d is not None and self._chk(d).dim() == 2)
        hb = int(Hdiag is not None and self._chk(Hdiag).dim() == 2)
        mk = lambda *shape: torch.empty(shape, dtype=torch.float64, device=self.device)  # noqa: E731
        if out is None:
            out = dict(logp=mk(nb), status=torch.empty(nb, dtype=torch.int32, device=self.device), A_bar=mk(nb, n, n),
                       B_bar=mk(nb, n, n), C_bar=mk(nb, n, n), D_bar=mk(nb, n, k),
                       q_bar=mk(nb, k, k) if full_covariance else mk(nb, k))
        # buffers this call needs that a reused ``out`` (from a call with another observation model) does not carry yet
        if d is not None and "d_bar" not in out:
            out["d_bar"] = mk(nb, p)
        if Hdiag is not None and "h_bar" not in out:
            out["h_bar"] = mk(nb, p)
        if dense_z and "Z_bar" not in out:
            out["Z_bar"] = mk(nb, p, n)
        if dense_z:
            with _lib.options_scope(options):
                _lib.check(
                    self.lib.dsge_solve_kalman_logp_grad_dense_z_batched(
                        self._p(A), self._p(B), self._p(C), self._p(D), self._p(q), q_mode, self._p(Z), zb, self._p(d), db,
                        self._p(Hdiag), hb, self._p(y), nb, n, k, p, T_len, _lib.SOLVER_CODES[solver], float(tol), int(max_iter),
                        float(jitter), float(missing_fill_value), int(n_filter_hint), int(n_lead_hint), self._p(out["logp"]),
                        out["status"].data_ptr(), self._p(out["A_bar"]), self._p(out["B_bar"]), self._p(out["C_bar"]),
                        self._p(out["D_bar"]), self._p(out["q_bar"]), self._p(out.get("d_bar")), self._p(out.get("h_bar")),
                        self._p(out.get("Z_bar")), self._stream(),
                    )
                )
            return out
        op, _keep = _lib.opt_ptr(options)
        _lib.check(
            self.lib.dsge_solve_kalman_logp_grad_batched_opt(
                op, self._p(A), self._p(B), self._p(C), self._p(D), self._p(q), q_mode, self._p(Z), zb, self._p(d), db,
                self._p(Hdiag), hb, self._p(y), nb, n, k, p, T_len, _lib.SOLVER_CODES[solver], float(tol), int(max_iter),
                float(jitter), float(missing_fill_value), int(n_filter_hint), int(n_lead_hint), self._p(out["logp"]),
                out["status"].data_ptr(), self._p(out["A_bar"]), self._p(out["B_bar"]), self._p(out["C_bar"]),
                self._p(out["D_bar"]), self._p(out["q_bar"]), self._p(out.get("d_bar")), self._p(out.get("h_bar")),
                self._stream(),
            )
        )
        return out

    def second_order_structure(self, A, C, Z):
        """(S, L, U) int32 numpy index lists of ``batched.second_order_structure`` from device tensors (a property of the
        model: one reduction + host sync, call once)."""
        n = A.shape[-1]
        nzA = (A != 0).reshape(-1, n).any(dim=0).cpu().numpy()
        nzC = (C != 0).reshape(-1, n).any(dim=0).cpu().numpy()
        nzZ = (Z != 0).reshape(-1, n).any(dim=0).cpu().numpy()
        S = np.flatnonzero(nzA)
        U = np.concatenate([S, np.setdiff1d(np.flatnonzero(nzZ), S)])
        return S.astype(np.int32), np.flatnonzero(nzC).astype(np.int32), U.astype(np.int32)

    def second_order_logp(self, A, B, C, D, hess_idx, hess_val, q, Z, y, structure, d=None, Hdiag=None,
                          solver="cycle_reduction", tol=1e-8, max_iter=1000, jitter=JITTER_DEFAULT,
                          missing_fill_value=MISSING_FILL, logp=None, status=None, stage_ms=None, options=None):
        """Second-order perturbation + pruned-state-space quasi-likelihood of the whole batch, device-resident
        (include/dsge_hip.h: ``dsge_second_order_logp_batched``; BASELINE configs[4]).  ``hess_idx``: int32 CUDA tensor
        (nnz, 3); ``hess_val``: float64 (batch, nnz); ``q``: (k,) or (batch, k); ``structure`` = ``second_order_structure``.
        ``stage_ms``: a ctypes float[4] that receives the stage durations (synchronises).  Returns (logp, status)."""
        torch = self.torch
        nb, n, _ = A.shape
        k = D.shape[2]
        T_len, p = y.shape
        for t in (A, B, C):
            self._chk(t, (nb, n, n))
        self._chk(D, (nb, n, k))
        self._chk(y, (T_len, p))
        self._chk(Z, (p, n))
        self._chk(q)
        if not (hess_idx.is_cuda and hess_idx.dtype == torch.int32 and hess_idx.is_contiguous() and hess_idx.shape[1] == 3):
            raise ValueError("hess_idx must be a contiguous int32 CUDA tensor (nnz, 3)")
        nnz = hess_idx.shape[0]
        self._chk(hess_val, (nb, nnz))
        S, Lc, U = (np.ascontiguousarray(x, dtype=np.int32) for x in structure)
        if logp is None:
            logp = torch.empty(nb, dtype=torch.float64, device=self.device)
        if status is None:
            status = torch.empty(nb, dtype=torch.int32, device=self.device)
        with _lib.options_scope(options):
            _lib.check(
                self.lib.dsge_second_order_logp_batched(
                    self._p(A), self._p(B), self._p(C), self._p(D), hess_idx.data_ptr(), nnz, self._p(hess_val), self._p(q),
                    int(q.dim() == 2), self._p(Z), self._p(d), self._p(Hdiag), self._p(y), nb, n, k, p, T_len,
                    _lib.SOLVER_CODES[solver], float(tol), int(max_iter), float(jitter), float(missing_fill_value),
                    S.ctypes.data, len(S), Lc.ctypes.data, len(Lc), U.ctypes.data, len(U), self._p(logp), status.data_ptr(),
                    None, None, None, None, None, None, None if stage_ms is None else __import__("ctypes").addressof(stage_ms),
                    self._stream(),
                )
            )
        return logp, status

    def profile_kernels(self, A, B, C, D, Q, Z, y, d=None, Hdiag=None, q_mode=None, solver="cycle_reduction",
                        tol=1e-6, max_iter=50, jitter=JITTER_DEFAULT, missing_fill_value=MISSING_FILL, reps=5,
                        n_state_hint=0, z_selector_hint=0, n_lead_hint=0):
        """Average per-kernel durations (ms) measured with HIP events on the launch stream:
        dict(solver=, assemble=, kalman=)."""
        import ctypes

        torch = self.torch
        nb, n, k, p, T_len, qm, zb, db, hb = self._pack(A, B, C, D, Q, Z, y, d, Hdiag, q_mode)
        logp = torch.empty(nb, dtype=torch.float64, device=self.device)
        status = torch.empty(nb, dtype=torch.int32, device=self.device)
        ms = (ctypes.c_float * 3)()
        _lib.check(
            self.lib.dsge_profile_pipeline(
                self._p(A), self._p(B), self._p(C), self._p(D), self._p(Q), qm, self._p(Z), zb, self._p(d), db,
                self._p(Hdiag), hb, self._p(y), nb, n, k, p, T_len, _lib.SOLVER_CODES[solver], float(tol),
                int(max_iter), float(jitter), float(missing_fill_value), int(n_state_hint), int(z_selector_hint),
                int(n_lead_hint), self._p(logp), status.data_ptr(), int(reps), ctypes.addressof(ms), self._stream(),
            )
        )
        return dict(solver=float(ms[0]), assemble=float(ms[1]), kalman=float(ms[2]))


class ShardedLogpEvaluator:
    """Draw-sharded evaluation across the ranks of a ``torch.distributed`` group.

    Rank r owns the contiguous draws ``[lo_r, hi_r)`` (``workloads.shard_bounds``), evaluates
    them locally with ``local_eval`` and the per-draw (logp, status) are all-gathered into
    rank-ordered buffers, so ``logp[i]`` is bit-exactly draw i on every rank (stricter than
    the reference's ``imap_unordered`` pool, perturbation_diagnostics.py:484-489).  There is
    no data-path collective besides that gather.

    ``local_eval(lo, hi) -> (logp, status)`` returns 1-D tensors of length hi-lo on
    ``device``; the product path passes a closure over ``LogpEngine.solve_kalman_logp``.
    """

    def __init__(self, global_batch, local_eval, device, group=None):
        import torch.distributed as dist

        torch = _torch()
        self.torch = torch
        self.dist = dist
        self.group = group
        self.distributed = dist.is_available() and dist.is_initialized()
        self.world = dist.get_world_size(group) if self.distributed else 1
        self.rank = dist.get_rank(group) if self.distributed else 0
        self.global_batch = int(global_batch)
        self.local_eval = local_eval
        self.device = device
        self.bounds = [shard_bounds(self.global_batch, self.world, r) for r in range(self.world)]
        self.lo, self.hi = self.bounds[self.rank]
        self.max_shard = max(hi - lo for lo, hi in self.bounds)
        # ONE collective per step: each rank contributes one byte record [logp f64 x m | status i32 x m | pad]
        m = self.max_shard
        self._rec = 8 * m + 4 * m + (-(12 * m) % 8)
        # RCCL gathers device buffers in place over xGMI; a gloo group (CPU tests, several ranks sharing one device)
        # stages the 12-byte-per-draw records through the host
        self.host_staged = bool(self.distributed and dist.get_backend(group) != "nccl" and
                                torch.device(device).type != "cpu")
        buf_dev = "cpu" if self.host_staged else device
        self._l_buf = torch.zeros(self._rec, dtype=torch.uint8, device=buf_dev)
        self._g_buf = torch.empty(self.world * self._rec, dtype=torch.uint8, device=buf_dev)
        self._l_logp = self._l_buf[: 8 * m].view(torch.float64)
        self._l_stat = self._l_buf[8 * m : 12 * m].view(torch.int32)
        self._l_logp.fill_(float("nan"))
        n_loc = self.hi - self.lo
        # the rank's slice of the send record: a ``local_eval`` that writes its logp / status straight into these (device views when
        # the group is RCCL) saves the two staging copies of every step
        self.local_logp = self._l_logp[:n_loc]
        self.local_status = self._l_stat[:n_loc]
        self._even = all(hi - lo == m for lo, hi in self.bounds)
        self._on_device = not self.host_staged

    def step(self):
        """Evaluate the local shard and gather; returns (logp, status) for ALL draws.  Per step: the local evaluation, ONE
        ``all_gather_into_tensor`` of the packed records, and one unpacking copy per output (the gathered records interleave logp
        and status per rank); no staging copy when ``local_eval`` wrote into ``local_logp`` / ``local_status``, no device transfer
        when the group gathers on the device."""
        logp, status = self.local_eval(self.lo, self.hi)
        if self.world == 1:
            return logp, status
        if logp.data_ptr() != self.local_logp.data_ptr():
            self.local_logp.copy_(logp)
        if status.data_ptr() != self.local_status.data_ptr():
            self.local_status.copy_(status)
        self.dist.all_gather_into_tensor(self._g_buf, self._l_buf, group=self.group)
        torch = self.torch
        m = self.max_shard
        recs = self._g_buf.view(self.world, self._rec)
        g_logp = recs[:, : 8 * m].view(torch.float64)        # (world, m) strided views of the gathered records
        g_stat = recs[:, 8 * m : 12 * m].view(torch.int32)
        if self._even:
            out_l, out_s = g_logp.reshape(-1), g_stat.reshape(-1)
        else:
            out_l = torch.cat([g_logp[r, : hi - lo] for r, (lo, hi) in enumerate(self.bounds)])
            out_s = torch.cat([g_stat[r, : hi - lo] for r, (lo, hi) in enumerate(self.bounds)])
        if self._on_device:
            return out_l, out_s
        return out_l.to(self.device), out_s.to(self.device)
