"""oracle.py -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Python face of the CPU oracle:

* ctypes binding of ``libpsa_oracle.so`` (``psa_oracle.c``: scalar C99
  restatement of the reference's RK4 / Yaman hot path, OpenMP over points);
* ``np_*``: a structurally faithful NumPy per-point restatement (same Python
  loop over z, 4-element complex128 arrays) -- used to cross-check the C port
  and as the like-for-like "reference-equivalent on this host" CPU timing;
* ``gain_from_summary``: the sweep drivers' per-point reduction
  (scan_mismtach.py:376-389 / 723-734).

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s cpu_baseline
leg may import this module.  Parity status: PINNED by golden vectors generated
from the reference itself (tests/golden/gen_golden.py, tests/test_oracle_golden.py).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

_f64p = np.ctypeslib.ndpointer(dtype=np.float64, flags="C_CONTIGUOUS")
_i64p = np.ctypeslib.ndpointer(dtype=np.int64, flags="C_CONTIGUOUS")


def build(force: bool = False) -> str:
    so = os.path.join(_HERE, "libpsa_oracle.so")
    src = os.path.join(_HERE, "psa_oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B"])
    return so


def lib() -> C.CDLL:
    global _LIB
    if _LIB is None:
        L = C.CDLL(build())
        L.psa_oracle_n_steps.restype = C.c_int64
        L.psa_oracle_n_steps.argtypes = [C.c_double, C.c_double]
        L.psa_oracle_rhs4.restype = None
        L.psa_oracle_rhs4.argtypes = [C.c_double, _f64p, C.c_double, C.c_double, C.c_double,
                                      _f64p, _f64p, _f64p, _f64p]
        L.psa_oracle_integrate.restype = C.c_int64
        L.psa_oracle_integrate.argtypes = [C.c_int, C.c_double, C.c_int64, C.c_int64, C.c_int, C.c_double,
                                           C.c_double, C.c_double, C.c_double, _f64p, _f64p, _f64p,
                                           C.POINTER(C.c_int64)]
        L.psa_oracle_sweep.restype = C.c_int
        L.psa_oracle_sweep.argtypes = [C.c_int, C.c_int64, C.c_double, C.c_int64, C.c_int64, C.c_int,
                                       _f64p, C.c_void_p, _f64p, C.c_int, _f64p, C.c_int, _f64p, C.c_int,
                                       _f64p, _f64p, _f64p, _i64p, C.c_int]
        L.psa_oracle_max_threads.restype = C.c_int
        _LIB = L
    return _LIB


def n_steps(z_max: float, dz: float) -> int:
    """integrators.py:194 ``int(round(z_max / dz))`` (round-half-even)."""
    return int(lib().psa_oracle_n_steps(float(z_max), float(dz)))


def _c128_as_f64(a: np.ndarray) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=np.complex128).view(np.float64)


def rhs4(z, a, gamma, alpha, dbeta):
    """yaman_model.py:10-52 for one point -> (rhs, linear, kerr, fwm) complex128 (4,)."""
    outs = [np.empty(8) for _ in range(4)]
    lib().psa_oracle_rhs4(float(z), _c128_as_f64(a).reshape(-1), float(gamma), float(alpha), float(dbeta), *outs)
    return tuple(o.view(np.complex128) for o in outs)


def integrate(a0, *, z_max, dz=None, n=None, save_every=1, check_nan=True, gamma, alpha, dbeta, dbeta2=0.0):
    """integrate_interval (integrators.py:150-204) with the Yaman RHS for one point.

    Returns (z_out, A[n_rows, n_waves], first_bad_step).  With ``check_nan`` and a
    non-finite state the reference raises FloatingPointError at ``first_bad_step``;
    here the rows saved before that step are returned together with the index.
    """
    a0 = np.asarray(a0, dtype=np.complex128)
    nw = a0.shape[0]
    if n is None:
        n = n_steps(z_max, dz)
    n_saved = n // save_every + 1
    z_out = np.empty(n_saved)
    y_out = np.empty(n_saved * nw * 2)
    rows = C.c_int64(0)
    bad = lib().psa_oracle_integrate(nw, float(z_max), int(n), int(save_every), int(bool(check_nan)),
                                     float(gamma), float(alpha), float(dbeta), float(dbeta2),
                                     _c128_as_f64(a0).reshape(-1), z_out, y_out, C.byref(rows))
    r = rows.value
    return z_out[:r], y_out.view(np.complex128).reshape(n_saved, nw)[:r], int(bad)


def sweep(dbeta, *, z_max, n, save_every=10, check_nan=True, gamma, alpha, a0, dbeta2=None, threads=0):
    """N independent points (the body of scan_mismtach.py:357-392 without plotting).

    gamma/alpha: scalar or (N,); a0: (n_waves,) or (N, n_waves) complex.
    Returns dict(a_end (N,nw) c128, p_end, p_max, first_bad_step).
    """
    dbeta = np.ascontiguousarray(dbeta, dtype=np.float64)
    N = dbeta.shape[0]
    gamma = np.ascontiguousarray(np.atleast_1d(gamma), dtype=np.float64)
    alpha = np.ascontiguousarray(np.atleast_1d(alpha), dtype=np.float64)
    a0 = np.ascontiguousarray(a0, dtype=np.complex128)
    nw = a0.shape[-1]
    gs = 1 if gamma.shape[0] == N and N > 1 else 0
    als = 1 if alpha.shape[0] == N and N > 1 else 0
    a0s = 1 if a0.ndim == 2 and a0.shape[0] == N and N > 1 else 0
    if a0.ndim == 2 and not a0s:
        a0 = np.ascontiguousarray(a0[0])
    a_end = np.empty(N * nw * 2)
    p_end = np.empty(N)
    p_max = np.empty(N)
    bad = np.empty(N, dtype=np.int64)
    d2 = None
    if dbeta2 is not None:
        d2a = np.ascontiguousarray(dbeta2, dtype=np.float64)
        d2 = d2a.ctypes.data_as(C.c_void_p)
    rc = lib().psa_oracle_sweep(nw, N, float(z_max), int(n), int(save_every), int(bool(check_nan)), dbeta, d2,
                                gamma, gs, alpha, als, _c128_as_f64(a0).reshape(-1), a0s,
                                a_end, p_end, p_max, bad, int(threads))
    if rc != 0:
        raise ValueError(f"psa_oracle_sweep rc={rc}")
    return dict(a_end=a_end.view(np.complex128).reshape(N, nw), p_end=p_end, p_max=p_max, first_bad_step=bad)


def max_threads() -> int:
    return int(lib().psa_oracle_max_threads())


def gain_from_summary(p_max, first_bad_step, p0_sig, unit="db"):
    """Per-point reduction of the sweep drivers (scan_mismtach.py:376-389).

    g = max_rows |A3|^2 / p_in[2]; non-finite or <= 0 -> NaN; any exception in the
    run (FloatingPointError from check_nan) -> NaN (scan_mismtach.py:391-392).
    """
    p_max = np.asarray(p_max, dtype=float)
    with np.errstate(all="ignore"):
        g = p_max / float(p0_sig)
        ok = np.isfinite(p_max) & np.isfinite(g) & (g > 0.0) & (np.asarray(first_bad_step) < 0)
        out = np.where(ok, g if unit == "linear" else 10.0 * np.log10(np.where(ok, g, 1.0)), np.nan)
    return out


# --------------------------------------------------------------------------
# Structurally faithful NumPy per-point restatement (Python loop, (4,) arrays)
# --------------------------------------------------------------------------
def np_rhs(z, a, gamma, alpha, dbeta):
    """yaman_model.py:10-52 with 4-element arrays (same expression order)."""
    lin = np.zeros_like(a) if alpha == 0.0 else (-0.5 * alpha) * a
    p = np.abs(a) ** 2
    f = np.array([p[0] + 2.0 * (p[1] + p[2] + p[3]), p[1] + 2.0 * (p[0] + p[2] + p[3]),
                  p[2] + 2.0 * (p[0] + p[1] + p[3]), p[3] + 2.0 * (p[0] + p[1] + p[2])])
    kerr = (1j * gamma) * (f * a)
    ep = np.exp(1j * dbeta * z)
    es = np.exp(-1j * dbeta * z)
    fwm = (1j * gamma * 2.0) * np.array([ep * (np.conj(a[1]) * a[2] * a[3]), ep * (np.conj(a[0]) * a[2] * a[3]),
                                         es * (np.conj(a[3]) * a[0] * a[1]), es * (np.conj(a[2]) * a[0] * a[1])])
    return lin + kerr + fwm


def np_integrate(a0, *, z_max, dz, save_every, check_nan, gamma, alpha, dbeta):
    """integrators.py:68-204 (linspace grid, per-step dz, strided save)."""
    n = int(round(z_max / dz))
    zg = np.linspace(0.0, z_max, n + 1)
    y = np.array(a0, dtype=np.complex128)
    n_saved = n // save_every + 1
    z_out = np.empty(n_saved)
    y_out = np.empty((n_saved, y.size), dtype=np.complex128)
    z_out[0] = zg[0]
    y_out[0] = y
    k = 1
    for i in range(n):
        z = zg[i]
        h = zg[i + 1] - zg[i]
        k1 = np_rhs(z, y, gamma, alpha, dbeta)
        k2 = np_rhs(z + 0.5 * h, y + 0.5 * h * k1, gamma, alpha, dbeta)
        k3 = np_rhs(z + 0.5 * h, y + 0.5 * h * k2, gamma, alpha, dbeta)
        k4 = np_rhs(z + h, y + h * k3, gamma, alpha, dbeta)
        y = y + (h / 6.0) * (k1 + 2.0 * k2 + 2.0 * k3 + k4)
        if check_nan and not np.all(np.isfinite(y)):
            raise FloatingPointError(f"NaN or Inf detected at step {i}, z = {z}")
        if (i + 1) % save_every == 0:
            z_out[k] = zg[i + 1]
            y_out[k] = y
            k += 1
    return z_out[:k], y_out[:k]


def np_rhs6(z, a, gamma, alpha, dbeta1, dbeta2):
    """Independent NumPy statement of the BUILD-DEFINED 6-wave model (no reference counterpart), waves
    [p1, p2, s1, i1, s2, i2]:

        dA_j/dz = -alpha/2 A_j + i*gamma*(|A_j|^2 + 2*sum_{k != j} |A_k|^2) A_j + 2i*gamma*M_j
        M_p1 = conj(A_p2) * (A_s1 A_i1 e^{+i db1 z} + A_s2 A_i2 e^{+i db2 z})      M_p2: p1 <-> p2
        M_s1 = conj(A_i1) * A_p1 A_p2 e^{-i db1 z}    M_i1 = conj(A_s1) * A_p1 A_p2 e^{-i db1 z}     (pair 2 alike)

    Same conventions as the 4-wave reference model (yaman_model.py:148-151, :174-186); with pair 2 dark it IS that model.
    """
    a = np.asarray(a, dtype=np.complex128)
    P = np.abs(a) ** 2
    f = P + 2.0 * (P.sum() - P)
    e1, e2 = np.exp(1j * dbeta1 * z), np.exp(1j * dbeta2 * z)
    pumps = a[2] * a[3] * e1 + a[4] * a[5] * e2
    q12 = a[0] * a[1]
    M = np.array([np.conj(a[1]) * pumps, np.conj(a[0]) * pumps,
                  np.conj(a[3]) * q12 * np.conj(e1), np.conj(a[2]) * q12 * np.conj(e1),
                  np.conj(a[5]) * q12 * np.conj(e2), np.conj(a[4]) * q12 * np.conj(e2)])
    return (-0.5 * alpha) * a + 1j * gamma * f * a + 2j * gamma * M


def np_integrate6(a0, *, z_max, n, gamma, alpha, dbeta1, dbeta2):
    """Plain RK4 (integrators.py:25-61) on np_rhs6 over np.linspace(0, z_max, n + 1); returns the final state."""
    zg = np.linspace(0.0, z_max, n + 1)
    y = np.array(a0, dtype=np.complex128)
    for i in range(n):
        z, h = zg[i], zg[i + 1] - zg[i]
        k1 = np_rhs6(z, y, gamma, alpha, dbeta1, dbeta2)
        k2 = np_rhs6(z + 0.5 * h, y + 0.5 * h * k1, gamma, alpha, dbeta1, dbeta2)
        k3 = np_rhs6(z + 0.5 * h, y + 0.5 * h * k2, gamma, alpha, dbeta1, dbeta2)
        k4 = np_rhs6(z + h, y + h * k3, gamma, alpha, dbeta1, dbeta2)
        y = y + (h / 6.0) * (k1 + 2.0 * k2 + 2.0 * k3 + k4)
    return y


# --------------------------------------------------------------------------
# CPU timing helpers for bench.py's cpu_baseline: the reference-shaped loop on every core,
# and a batched-NumPy form (arrays over sweep points) for context.
# --------------------------------------------------------------------------
def _np_point_job(args):
    a0, z_max, dz, dbeta = args
    z, A = np_integrate(a0, z_max=z_max, dz=dz, save_every=10, check_nan=True, gamma=0.0115, alpha=1.15e-4, dbeta=dbeta)
    return A[-1]


def np_integrate_all_cores(a0, *, z_max, dz, dbetas, procs):
    """One reference-shaped run per process (multiprocessing.Pool), as BASELINE.md section 2 did for the reference."""
    import multiprocessing as mp
    with mp.get_context("fork").Pool(procs) as pool:
        return pool.map(_np_point_job, [(a0, z_max, dz, float(d)) for d in dbetas])


def np_sweep_batched(dbeta, *, z_max, n, gamma, alpha, a0):
    """Batched NumPy RK4: the same algorithm with arrays over the sweep points (shape (4, N)); final state (N, 4)."""
    dbeta = np.asarray(dbeta, dtype=float)
    y = np.repeat(np.asarray(a0, dtype=np.complex128)[:, None], dbeta.size, axis=1)

    def rhs(z, a):
        P = (a.real ** 2 + a.imag ** 2)
        f = 2.0 * P.sum(0) - P
        e = np.exp(1j * dbeta * z)
        q12, q34 = a[0] * a[1], a[2] * a[3]
        fw = np.stack([np.conj(a[1]) * q34 * e, np.conj(a[0]) * q34 * e,
                       np.conj(a[3]) * q12 * np.conj(e), np.conj(a[2]) * q12 * np.conj(e)])
        return (-0.5 * alpha) * a + 1j * gamma * (f * a) + 2j * gamma * fw

    h = z_max / n
    for i in range(n):
        z = i * h
        k1 = rhs(z, y)
        k2 = rhs(z + 0.5 * h, y + 0.5 * h * k1)
        k3 = rhs(z + 0.5 * h, y + 0.5 * h * k2)
        k4 = rhs(z + h, y + h * k3)
        y = y + (h / 6.0) * (k1 + 2.0 * k2 + 2.0 * k3 + k4)
    return y.T
