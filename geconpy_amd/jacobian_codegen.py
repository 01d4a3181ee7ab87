"""On-device evaluation of the linearised-model Jacobians  theta -> A, B, C, D (, q).

SURVEY.md section 8(f1).  In the reference the entries of A, B, C, D are sympy expressions of the
deep parameters: ``build_symbolic_jacobians`` differentiates the model once, runs ONE shared
``sp.cse`` pass over all four matrices (gEconpy/model/compile.py:163-222) and hands the result to
pytensor; every logp call then re-evaluates that scalar graph on the host
(gEconpy/model/perturbation.py:160-190) and, for this engine, would ship 3 n^2 + n k doubles per
draw over PCIe.  This module does the same shared-CSE pass and prints the result as a HIP kernel
instead -- one thread per parameter draw, the draw's ``npar`` parameters in, the structurally
non-zero entries out (the matrices are zero-filled by ``hipMemsetAsync`` first) -- so that a batch
of draws crosses the bus as ``batch x npar`` doubles (RBC: 56 B per draw instead of 1.6 KB;
SW-sized models: ~300 B instead of 40 KB) and the matrices are born in HBM.

The generated code is model specific, so it lives in its own small shared library with a plain
C ABI (the same conventions as include/dsge_hip.h):

    int dsge_jac_dims(int* n, int* k, int* npar, int* has_q);
    int dsge_jac_launch(const double* theta, int batch,      /* device, [batch][npar]            */
                        double* A, double* B, double* C,     /* device, [batch][n][n]            */
                        double* D, double* q,                /* device, [batch][n][k], [batch][k] */
                        void* stream);                       /* enqueue only                     */
    int dsge_jac_vjp_launch(const double* theta, int batch,  /* pullback: cotangents of A,B,C,D,q ...   */
                            const double* A_bar, const double* B_bar, const double* C_bar,
                            const double* D_bar, const double* q_bar,
                            double* theta_bar, void* stream);    /* ... -> [batch][npar]                */

compiled for gfx950 with hipcc (``build()``), cached under ``geconpy_amd/_jit/`` by source hash and
loaded with ctypes.  There is no host evaluator here: ``lambdify`` of the same expressions is what
the tests use as the checker.
"""
from __future__ import annotations

import ctypes as C
import hashlib
import os
import shutil
import subprocess

PKG = os.path.dirname(os.path.abspath(__file__))
JIT_DIR = os.path.join(PKG, "_jit")


def _hipcc():
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: cannot build the Jacobian kernel")


class JacobianProgram:
    """Shared-CSE HIP program for ``theta -> A, B, C, D (, q)``.

    params : sequence of sympy Symbols, the column order of ``theta``
    A, B, C : sympy (n, n) matrices;  D : (n, k);  q : optional length-k sequence of shock variances
    """

    def __init__(self, name, params, A, B, C_, D, q=None, Z=None, d=None):
        """``Z`` (p, n) / ``d`` (p,): optional parameter-dependent observation equation -- the linearised design rows of
        ``_make_design_matrix`` (gEconpy/model/statespace.py:298-332) and the steady-state intercept of
        ``_make_obs_intercept`` (:334-388) -- evaluated by a second generated kernel (``launch_obs``)."""
        import sympy as sp

        self.name = str(name)
        self.params = list(params)
        self.mats = [sp.Matrix(M) for M in (A, B, C_, D)]
        self.n = self.mats[0].shape[0]
        self.k = self.mats[3].shape[1]
        for M, shape in zip(self.mats, [(self.n, self.n)] * 3 + [(self.n, self.k)]):
            if M.shape != shape:
                raise ValueError(f"expected shape {shape}, got {M.shape}")
        self.q = None if q is None else [sp.sympify(x) for x in q]
        if self.q is not None and len(self.q) != self.k:
            raise ValueError("q must have one entry per shock")
        self.Z = None if Z is None else sp.Matrix(Z)
        self.d = None if d is None else [sp.sympify(x) for x in d]
        if self.Z is not None and self.Z.shape[1] != self.n:
            raise ValueError("Z must have one column per model variable")
        self.p = self.Z.shape[0] if self.Z is not None else (len(self.d) if self.d is not None else 0)
        if self.Z is not None and self.d is not None and len(self.d) != self.p:
            raise ValueError("d must have one entry per row of Z")
        free = set().union(*(M.free_symbols for M in self.mats))
        if self.q is not None:
            free |= set().union(*(x.free_symbols for x in self.q))
        if self.Z is not None:
            free |= self.Z.free_symbols
        if self.d is not None:
            free |= set().union(*(sp.sympify(x).free_symbols for x in self.d))
        unknown = free - set(self.params)
        if unknown:
            raise ValueError(f"expressions use symbols that are not parameters: {sorted(map(str, unknown))}")
        self._lib = None
        self._source = None

    # -- code generation -------------------------------------------------------------------
    def nonzero_entries(self):
        """[(matrix index 0..3 or 4 for q, flat index, expression)] of the structurally non-zero entries."""
        out = []
        for mi, M in enumerate(self.mats):
            cols = M.shape[1]
            for r in range(M.shape[0]):
                for c in range(cols):
                    if M[r, c] != 0:
                        out.append((mi, r * cols + c, M[r, c]))
        if self.q is not None:
            out += [(4, j, e) for j, e in enumerate(self.q)]
        return out

    def source(self):
        if self._source is None:
            self._source = self._generate()
        return self._source

    def _generate(self):
        import sympy as sp
        from sympy.printing.c import C99CodePrinter

        # The parameters are printed as par0, par1, ...: a model parameter may be called theta, A, q, draw, x0 or carry
        # characters that are not valid in a C identifier, and none of that may meet the kernel's own names.
        ren = {p_: sp.Symbol(f"par{i}", **p_.assumptions0) for i, p_ in enumerate(self.params)}
        params = [ren[p_] for p_ in self.params]
        entries = [(mi, flat, e.xreplace(ren)) for mi, flat, e in self.nonzero_entries()]
        repl, reduced = sp.cse([e for _, _, e in entries], symbols=sp.numbered_symbols("x"), optimizations="basic")
        class _Printer(C99CodePrinter):
            def _print_Integer(self, expr):  # an integer beyond 32 bits is written as a double literal (found by tools/fuzz_theta.py:
                v = int(expr)                # sympy folds rational coefficients into integers clang cannot represent)
                return f"{v}.0" if abs(v) >= 2 ** 31 else super()._print_Integer(expr)

        pr = _Printer()
        npar, n, k = len(self.params), self.n, self.k
        lines = [
            "// generated by geconpy_amd/jacobian_codegen.py -- do not edit",
            f"// model {self.name}: n = {n}, k = {k}, npar = {npar}, {len(entries)} non-zero entries, {len(repl)} CSE temporaries",
            "#include <hip/hip_runtime.h>",
            "#include <stdint.h>",
            f"#define JAC_N {n}",
            f"#define JAC_K {k}",
            f"#define JAC_NPAR {npar}",
            "",
            "__global__ __launch_bounds__(256) void jac_kernel(const double* __restrict__ theta, int batch,",
            "    double* __restrict__ A, double* __restrict__ B, double* __restrict__ C, double* __restrict__ D,",
            "    double* __restrict__ q) {",
            "  const int draw = blockIdx.x * 256 + threadIdx.x;",
            "  if (draw >= batch) return;",
            "  const double* th = theta + (size_t)draw * JAC_NPAR;",
        ]
        for i, p in enumerate(params):
            lines.append(f"  const double {pr.doprint(p)} = th[{i}];")
        for sym, expr in repl:
            lines.append(f"  const double {pr.doprint(sym)} = {pr.doprint(expr)};")
        bases = ["A + (size_t)draw * JAC_N * JAC_N", "B + (size_t)draw * JAC_N * JAC_N", "C + (size_t)draw * JAC_N * JAC_N",
                 "D + (size_t)draw * JAC_N * JAC_K", "q + (size_t)draw * JAC_K"]
        names = ["Ad", "Bd", "Cd", "Dd", "qd"]
        used = sorted({mi for mi, _, _ in entries})
        for mi in used:
            lines.append(f"  double* {names[mi]} = {bases[mi]};")
        for (mi, flat, _), expr in zip(entries, reduced):
            lines.append(f"  {names[mi]}[{flat}] = {pr.doprint(expr)};")
        lines += ["}", ""]
        # ---- pullback kernel: theta_bar_i = sum_e bar_e * d expr_e / d theta_i (one shared CSE pass again)
        bar_syms = [sp.Symbol(f"g{mi}_{flat}") for mi, flat, _ in entries]
        vjp_exprs = [sp.Add(*[g * sp.diff(e, th_) for g, (_, _, e) in zip(bar_syms, entries)]) for th_ in params]
        used_bars = sorted(set().union(*(v.free_symbols for v in vjp_exprs)) & set(bar_syms), key=lambda x_: bar_syms.index(x_))
        repl2, reduced2 = sp.cse(vjp_exprs, symbols=sp.numbered_symbols("z"), optimizations="basic")
        bar_names = ["Ab", "Bb", "Cb", "Db", "qb"]
        lines += [
            "__global__ __launch_bounds__(256) void jac_vjp_kernel(const double* __restrict__ theta, int batch,",
            "    const double* __restrict__ A_bar, const double* __restrict__ B_bar, const double* __restrict__ C_bar,",
            "    const double* __restrict__ D_bar, const double* __restrict__ q_bar, double* __restrict__ theta_bar) {",
            "  const int draw = blockIdx.x * 256 + threadIdx.x;",
            "  if (draw >= batch) return;",
            "  const double* th = theta + (size_t)draw * JAC_NPAR;",
        ]
        for i, p_ in enumerate(params):
            lines.append(f"  const double {pr.doprint(p_)} = th[{i}];")
        bar_bases = ["A_bar + (size_t)draw * JAC_N * JAC_N", "B_bar + (size_t)draw * JAC_N * JAC_N",
                     "C_bar + (size_t)draw * JAC_N * JAC_N", "D_bar + (size_t)draw * JAC_N * JAC_K", "q_bar + (size_t)draw * JAC_K"]
        used_m = sorted({entries[bar_syms.index(g)][0] for g in used_bars})
        for mi in used_m:
            lines.append(f"  const double* {bar_names[mi]} = {bar_bases[mi]};")
        for g in used_bars:
            mi, flat, _ = entries[bar_syms.index(g)]
            lines.append(f"  const double {g} = {bar_names[mi]}[{flat}];")
        for sym, expr in repl2:
            lines.append(f"  const double {pr.doprint(sym)} = {pr.doprint(expr)};")
        lines.append("  double* tb = theta_bar + (size_t)draw * JAC_NPAR;")
        for i, expr in enumerate(reduced2):
            lines.append(f"  tb[{i}] = {pr.doprint(expr)};")
        lines += ["}", ""]
        has_obs = self.Z is not None or self.d is not None
        if has_obs:
            # ---- observation equation: Z [batch][p][n] (zero-filled by the launcher), d [batch][p], and the pullback of d
            zent = []
            if self.Z is not None:
                for r in range(self.p):
                    for c_ in range(n):
                        if self.Z[r, c_] != 0:
                            zent.append((0, r * n + c_, self.Z[r, c_].xreplace(ren)))
            dent = [(1, j, e.xreplace(ren)) for j, e in enumerate(self.d)] if self.d is not None else []
            oent = zent + dent
            repl3, red3 = sp.cse([e for _, _, e in oent], symbols=sp.numbered_symbols("w"), optimizations="basic")
            lines += [
                f"#define JAC_P {self.p}",
                "__global__ __launch_bounds__(256) void jac_obs_kernel(const double* __restrict__ theta, int batch,",
                "    double* __restrict__ Z, double* __restrict__ d) {",
                "  const int draw = blockIdx.x * 256 + threadIdx.x;",
                "  if (draw >= batch) return;",
                "  const double* th = theta + (size_t)draw * JAC_NPAR;",
            ]
            for i, p_ in enumerate(params):
                lines.append(f"  const double {pr.doprint(p_)} = th[{i}];")
            for sym, expr in repl3:
                lines.append(f"  const double {pr.doprint(sym)} = {pr.doprint(expr)};")
            for (mi, flat, _), expr in zip(oent, red3):
                tgt = f"Z[(size_t)draw * JAC_P * JAC_N + {flat}]" if mi == 0 else f"d[(size_t)draw * JAC_P + {flat}]"
                lines.append(f"  {tgt} = {pr.doprint(expr)};")
            lines += ["}", ""]
            if self.d is not None:
                gs = [sp.Symbol(f"gd_{j}") for j in range(self.p)]
                dv = [sp.Add(*[g * sp.diff(e.xreplace(ren), th_) for g, e in zip(gs, self.d)]) for th_ in params]
                repl4, red4 = sp.cse(dv, symbols=sp.numbered_symbols("v"), optimizations="basic")
                lines += [
                    "// theta_bar += (d d / d theta)' d_bar",
                    "__global__ __launch_bounds__(256) void jac_obs_vjp_kernel(const double* __restrict__ theta, int batch,",
                    "    const double* __restrict__ d_bar, double* __restrict__ theta_bar) {",
                    "  const int draw = blockIdx.x * 256 + threadIdx.x;",
                    "  if (draw >= batch) return;",
                    "  const double* th = theta + (size_t)draw * JAC_NPAR;",
                ]
                for i, p_ in enumerate(params):
                    lines.append(f"  const double {pr.doprint(p_)} = th[{i}];")
                for j, g in enumerate(gs):
                    lines.append(f"  const double {g} = d_bar[(size_t)draw * JAC_P + {j}];")
                for sym, expr in repl4:
                    lines.append(f"  const double {pr.doprint(sym)} = {pr.doprint(expr)};")
                lines.append("  double* tb = theta_bar + (size_t)draw * JAC_NPAR;")
                for i, expr in enumerate(red4):
                    lines.append(f"  tb[{i}] += {pr.doprint(expr)};")
                lines += ["}", ""]
            if zent:
                gz = [sp.Symbol(f"gz_{flat}") for _, flat, _ in zent]
                zv_ = [sp.Add(*[g * sp.diff(e, th_) for g, (_, _, e) in zip(gz, zent)]) for th_ in params]
                repl5, red5 = sp.cse(zv_, symbols=sp.numbered_symbols("u"), optimizations="basic")
                used_gz = sorted(set().union(*(v.free_symbols for v in zv_)) & set(gz), key=lambda x_: gz.index(x_))
                lines += [
                    "// theta_bar += (d Z / d theta)' Z_bar   (Z_bar: [batch][p][n], the cotangent of every entry of Z)",
                    "__global__ __launch_bounds__(256) void jac_obs_z_vjp_kernel(const double* __restrict__ theta, int batch,",
                    "    const double* __restrict__ Z_bar, double* __restrict__ theta_bar) {",
                    "  const int draw = blockIdx.x * 256 + threadIdx.x;",
                    "  if (draw >= batch) return;",
                    "  const double* th = theta + (size_t)draw * JAC_NPAR;",
                ]
                for i, p_ in enumerate(params):
                    lines.append(f"  const double {pr.doprint(p_)} = th[{i}];")
                for g in used_gz:
                    flat = zent[gz.index(g)][1]
                    lines.append(f"  const double {g} = Z_bar[(size_t)draw * JAC_P * JAC_N + {flat}];")
                for sym, expr in repl5:
                    lines.append(f"  const double {pr.doprint(sym)} = {pr.doprint(expr)};")
                lines.append("  double* tb = theta_bar + (size_t)draw * JAC_NPAR;")
                for i, expr in enumerate(red5):
                    if expr != 0:
                        lines.append(f"  tb[{i}] += {pr.doprint(expr)};")
                lines += ["}", ""]
        lines += [
            'extern "C" {',
            "int dsge_jac_dims(int* n, int* k, int* npar, int* has_q) {",
            f"  *n = JAC_N; *k = JAC_K; *npar = JAC_NPAR; *has_q = {0 if self.q is None else 1};",
            "  return 0;",
            "}",
            "// device pointers; enqueues the zero fill and the kernel on `stream`; 0 = success",
            "int dsge_jac_launch(const double* theta, int batch, double* A, double* B, double* C, double* D, double* q,",
            "                    void* stream) {",
            "  hipStream_t st = (hipStream_t)stream;",
            "  if (batch < 0 || !theta || !A || !B || !C || !D) return 1;",
            f"  if ({0 if self.q is None else 1} && !q) return 1;",
            "  if (batch == 0) return 0;",
            "  const size_t nn = (size_t)batch * JAC_N * JAC_N * sizeof(double);",
            "  if (hipMemsetAsync(A, 0, nn, st) != hipSuccess) return 2;",
            "  if (hipMemsetAsync(B, 0, nn, st) != hipSuccess) return 2;",
            "  if (hipMemsetAsync(C, 0, nn, st) != hipSuccess) return 2;",
            "  if (hipMemsetAsync(D, 0, (size_t)batch * JAC_N * JAC_K * sizeof(double), st) != hipSuccess) return 2;",
            "  hipLaunchKernelGGL(jac_kernel, dim3((batch + 255) / 256), dim3(256), 0, st, theta, batch, A, B, C, D, q);",
            "  return hipGetLastError() == hipSuccess ? 0 : 2;",
            "}",
            "// pullback: theta_bar = J' (A_bar, B_bar, C_bar, D_bar, q_bar); device pointers; 0 = success",
            "int dsge_jac_vjp_launch(const double* theta, int batch, const double* A_bar, const double* B_bar,",
            "                        const double* C_bar, const double* D_bar, const double* q_bar, double* theta_bar,",
            "                        void* stream) {",
            "  if (batch < 0 || !theta || !A_bar || !B_bar || !C_bar || !D_bar || !theta_bar) return 1;",
            f"  if ({0 if self.q is None else 1} && !q_bar) return 1;",
            "  if (batch == 0) return 0;",
            "  hipLaunchKernelGGL(jac_vjp_kernel, dim3((batch + 255) / 256), dim3(256), 0, (hipStream_t)stream, theta, batch,",
            "                     A_bar, B_bar, C_bar, D_bar, q_bar, theta_bar);",
            "  return hipGetLastError() == hipSuccess ? 0 : 2;",
            "}",
        ]
        if has_obs:
            lines += [
                "// observation equation: Z [batch][p][n] (NULL if the program has none), d [batch][p] (NULL if none); 0 = success",
                "int dsge_jac_obs_dims(int* p, int* has_Z, int* has_d) {",
                f"  *p = JAC_P; *has_Z = {0 if self.Z is None else 1}; *has_d = {0 if self.d is None else 1};",
                "  return 0;",
                "}",
                "int dsge_jac_obs_launch(const double* theta, int batch, double* Z, double* d, void* stream) {",
                "  hipStream_t st = (hipStream_t)stream;",
                f"  if (batch < 0 || !theta || ({0 if self.Z is None else 1} && !Z) || ({0 if self.d is None else 1} && !d)) return 1;",
                "  if (batch == 0) return 0;",
                "  if (Z && hipMemsetAsync(Z, 0, (size_t)batch * JAC_P * JAC_N * sizeof(double), st) != hipSuccess) return 2;",
                "  hipLaunchKernelGGL(jac_obs_kernel, dim3((batch + 255) / 256), dim3(256), 0, st, theta, batch, Z, d);",
                "  return hipGetLastError() == hipSuccess ? 0 : 2;",
                "}",
            ]
            if self.d is not None:
                lines += [
                    "int dsge_jac_obs_vjp_launch(const double* theta, int batch, const double* d_bar, double* theta_bar, void* stream) {",
                    "  if (batch < 0 || !theta || !d_bar || !theta_bar) return 1;",
                    "  if (batch == 0) return 0;",
                    "  hipLaunchKernelGGL(jac_obs_vjp_kernel, dim3((batch + 255) / 256), dim3(256), 0, (hipStream_t)stream, theta, batch,",
                    "                     d_bar, theta_bar);",
                    "  return hipGetLastError() == hipSuccess ? 0 : 2;",
                    "}",
                ]
        if has_obs and self.Z is not None:
            lines += [
                "int dsge_jac_obs_z_vjp_launch(const double* theta, int batch, const double* Z_bar, double* theta_bar, void* stream) {",
                "  if (batch < 0 || !theta || !Z_bar || !theta_bar) return 1;",
                "  if (batch == 0) return 0;",
            ]
            if zent:  # (a design matrix without a non-zero entry has no pullback kernel: nothing to add)
                lines += [
                    "  hipLaunchKernelGGL(jac_obs_z_vjp_kernel, dim3((batch + 255) / 256), dim3(256), 0, (hipStream_t)stream, theta, batch,",
                    "                     Z_bar, theta_bar);",
                ]
            lines += [
                "  return hipGetLastError() == hipSuccess ? 0 : 2;",
                "}",
            ]
        lines += ["}", ""]
        return "\n".join(lines)

    # -- build / load ----------------------------------------------------------------------
    def lib_path(self):
        h = hashlib.sha256(self.source().encode()).hexdigest()[:16]
        safe = "".join(ch if (ch.isalnum() and ch.isascii()) or ch == "_" else "_" for ch in self.name)  # model names are free text
        return os.path.join(JIT_DIR, f"libjac_{safe}_{h}.so")

    def build(self, force=False, verbose=False):
        """Compile for gfx950 (cross-compiles without a GPU); returns the path of the shared library."""
        path = self.lib_path()
        if os.path.exists(path) and not force:
            return path
        os.makedirs(JIT_DIR, exist_ok=True)
        src = path[:-3] + ".hip"
        with open(src, "w") as fh:
            fh.write(self.source())
        cmd = [_hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-o", path, src]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)
        return path

    def load(self):
        if self._lib is None:
            path = self.lib_path()
            if not os.path.exists(path):
                path = self.build()
            try:
                import torch  # noqa: F401  (one HIP runtime per process, see _lib.load)
            except ImportError:
                pass
            lib = C.CDLL(path)
            lib.dsge_jac_dims.argtypes = [C.POINTER(C.c_int)] * 4
            lib.dsge_jac_launch.argtypes = [C.c_void_p, C.c_int] + [C.c_void_p] * 6
            lib.dsge_jac_vjp_launch.argtypes = [C.c_void_p, C.c_int] + [C.c_void_p] * 7
            if self.Z is not None or self.d is not None:
                lib.dsge_jac_obs_launch.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
                if self.d is not None:
                    lib.dsge_jac_obs_vjp_launch.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
            dims = [C.c_int() for _ in range(4)]
            lib.dsge_jac_dims(*[C.byref(d) for d in dims])
            if (dims[0].value, dims[1].value, dims[2].value) != (self.n, self.k, len(self.params)):
                raise RuntimeError("generated Jacobian library does not match the program")
            self._lib = lib
        return self._lib

    def launch(self, theta_ptr, batch, A_ptr, B_ptr, C_ptr, D_ptr, q_ptr, stream):
        rc = self.load().dsge_jac_launch(theta_ptr, int(batch), A_ptr, B_ptr, C_ptr, D_ptr, q_ptr, stream)
        if rc != 0:
            raise RuntimeError(f"dsge_jac_launch failed with code {rc}")


    def launch_obs(self, theta_ptr, batch, Z_ptr, d_ptr, stream):
        """theta -> Z [batch][p][n], d [batch][p] (device pointers; the one the program lacks may be None)."""
        rc = self.load().dsge_jac_obs_launch(theta_ptr, int(batch), Z_ptr, d_ptr, stream)
        if rc != 0:
            raise RuntimeError(f"dsge_jac_obs_launch failed with code {rc}")

    def launch_obs_vjp(self, theta_ptr, batch, d_bar, theta_bar, stream):
        """theta_bar += (d d / d theta)' d_bar."""
        rc = self.load().dsge_jac_obs_vjp_launch(theta_ptr, int(batch), d_bar, theta_bar, stream)
        if rc != 0:
            raise RuntimeError(f"dsge_jac_obs_vjp_launch failed with code {rc}")

    def launch_obs_z_vjp(self, theta_ptr, batch, Z_bar, theta_bar, stream):
        """theta_bar += (d Z / d theta)' Z_bar  (programs built with ``Z=``)."""
        lib = self.load()
        lib.dsge_jac_obs_z_vjp_launch.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        rc = lib.dsge_jac_obs_z_vjp_launch(theta_ptr, int(batch), Z_bar, theta_bar, stream)
        if rc != 0:
            raise RuntimeError(f"dsge_jac_obs_z_vjp_launch failed with code {rc}")

    def launch_vjp(self, theta_ptr, batch, A_bar, B_bar, C_bar, D_bar, q_bar, theta_bar, stream):
        rc = self.load().dsge_jac_vjp_launch(theta_ptr, int(batch), A_bar, B_bar, C_bar, D_bar, q_bar, theta_bar, stream)
        if rc != 0:
            raise RuntimeError(f"dsge_jac_vjp_launch failed with code {rc}")


def rbc_linearized_program():
    """The reference's ``tests/_resources/test_gcns/rbc_linearized.gcn`` (8 linear equations :22-49, steady
    state :6-20) as sympy Jacobians of F = LHS - RHS; variable order [A, C, I, K, L, R, W, Y], parameters
    (sigma, phi, alpha, beta, delta, rho_A, sigma_A), q = [sigma_A^2].  Symbolic twin of
    ``workloads.rbc_linearized_jacobians`` (the tests check one against the other)."""
    import sympy as sp

    sigma, phi, alpha, beta, delta, rho_A, sigma_A = params = sp.symbols("sigma phi alpha beta delta rho_A sigma_A",
                                                                         positive=True)
    R = 1 / beta - (1 - delta)
    W = (1 - alpha) ** (1 / (1 - alpha)) * (alpha / R) ** (alpha / (1 - alpha))
    Y = (R / (R - delta * alpha)) ** (sigma / (sigma + phi)) * ((1 - alpha) ** (-phi) * W ** (1 + phi)) ** (1 / (sigma + phi))
    K = alpha * Y / R
    I = delta * K  # noqa: E741
    Cs = Y - I
    iA, iC, iI, iK, iL, iR, iW, iY = range(8)
    A, B, Cm, D = sp.zeros(8, 8), sp.zeros(8, 8), sp.zeros(8, 8), sp.zeros(8, 1)
    B[0, iW], B[0, iC], B[0, iL] = 1, -sigma, -phi                       # W = sigma C + phi L
    Cm[1, iC], B[1, iC], Cm[1, iR] = sigma / beta, -sigma / beta, -R      # sigma/beta (C[1] - C) = R_ss R[1]
    B[2, iK], A[2, iK], B[2, iI] = 1, -(1 - delta), -delta                # K = (1-delta) K[-1] + delta I
    B[3, iY], B[3, iA], A[3, iK], B[3, iL] = 1, -1, -alpha, -(1 - alpha)  # Y = A + alpha K[-1] + (1-alpha) L
    B[4, iR], B[4, iY], A[4, iK] = 1, -1, 1                               # R = Y - K[-1]
    B[5, iW], B[5, iY], B[5, iL] = 1, -1, 1                               # W = Y - L
    B[6, iY], B[6, iC], B[6, iI] = Y, -Cs, -I                             # Y_ss Y = C_ss C + I_ss I
    B[7, iA], A[7, iA], D[7, 0] = 1, -rho_A, -1                           # A = rho_A A[-1] + eps
    return JacobianProgram("rbc_linearized", params, A, B, Cm, D, q=[sigma_A**2])


SW_THETA_SCALE = 0.02


def sw_shaped_program(seed=None):
    """A parameterisation of the SW-shaped workload (BASELINE configs[2]; the reference ships no Smets-Wouters model, so
    there is no economic theta behind those systems): the base system is draw 0 of ``workloads.sw_shaped_batch`` and

        A[:, S_j] = A0[:, S_j] (1 + 0.02 a_j)   (18 state columns)      B = B0      D = D0
        C[:, L_j] = C0[:, L_j] (1 + 0.02 c_j)   (12 forward-looking columns)         q_j = sigma_j^2   (7 shocks)

    theta = (a_0..a_17, c_0..c_11, sigma_0..sigma_6): 37 parameters, every entry affine in theta, the zero-column
    structure (states / leads / static variables) that of the base system for every theta.  Goes through the same
    shared-CSE code generator as the RBC model: 2 600 non-zero entries, one thread per draw.  ``workloads.sw_theta_draws``
    draws a, c ~ U(-1, 1) and sigma ~ U(0.005, 0.02); within that box the systems keep a unique stable solution."""
    import numpy as np
    import sympy as sp

    from . import workloads as wl

    A0, B0, C0, D0, _ = wl.sw_shaped_system(wl.SW_SEED0 if seed is None else seed)
    n, k = D0.shape
    S = np.flatnonzero((A0 != 0).any(axis=0))
    Lc = np.flatnonzero((C0 != 0).any(axis=0))
    a = sp.symbols(f"a0:{len(S)}", real=True)
    c = sp.symbols(f"c0:{len(Lc)}", real=True)
    sg = sp.symbols(f"sigma0:{k}", positive=True)
    sc = sp.Float(SW_THETA_SCALE)

    def flt(x):
        return sp.Float(repr(float(x)), 17)

    A, B, Cm, D = sp.zeros(n, n), sp.zeros(n, n), sp.zeros(n, n), sp.zeros(n, k)
    for j, col in enumerate(S):
        for r in range(n):
            if A0[r, col] != 0:
                A[r, col] = flt(A0[r, col]) * (1 + sc * a[j])
    for j, col in enumerate(Lc):
        for r in range(n):
            if C0[r, col] != 0:
                Cm[r, col] = flt(C0[r, col]) * (1 + sc * c[j])
    for r in range(n):
        for col in range(n):
            if B0[r, col] != 0:
                B[r, col] = flt(B0[r, col])
        for col in range(k):
            if D0[r, col] != 0:
                D[r, col] = flt(D0[r, col])
    return JacobianProgram("sw_shaped_theta", list(a) + list(c) + list(sg), A, B, Cm, D, q=[s_ ** 2 for s_ in sg])
