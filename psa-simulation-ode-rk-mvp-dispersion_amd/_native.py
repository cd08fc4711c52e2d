"""ctypes binding of ``libpsa_hip.so`` (the C-ABI declared in ``include/psa_rk4.h``).

This is the ONLY compute path of the package: there is no CPU fallback.  If the
shared library is missing, or no gfx950 device is visible when a compute entry
point is called, a ``NativeUnavailableError`` / ``PsaNativeError`` is raised.

Two faces:

* ``sweep_host`` / ``yaman_rhs_host`` / ``gain_summary_host`` take NumPy arrays
  (host buffers, blocking) -- what the reference-shaped Python API uses;
* ``sweep_device`` takes raw device pointers + a hipStream_t handle (e.g. from
  torch tensors) -- what ``bench.py`` and the multi-GPU driver use.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

import numpy as np

_PKG_DIR = os.path.dirname(os.path.abspath(__file__))
# PSA_HIP_LIB: developer hook to load an A/B build of the same library (tools/ab_build.sh); never a different backend
LIB_PATH = os.environ.get("PSA_HIP_LIB") or os.path.join(_PKG_DIR, "libpsa_hip.so")

# flags (mirror include/psa_rk4.h)
BCAST_GAMMA = 1 << 0
BCAST_ALPHA = 1 << 1
BCAST_A0 = 1 << 2
OPT_CHECK_NAN = 1 << 8
OPT_EXACT_STEP = 1 << 9
OPT_LDS_STAGING = 1 << 10
OPT_BLOCK64 = 1 << 11
OPT_F32_SCALAR = 1 << 12
OPT_F32_PACKED = 1 << 13
OPT_LOSSLESS = 1 << 14
OPT_SPLIT_POINT = 1 << 15
OPT_ONE_LANE = 1 << 16
OPT_TRAJ_LD = 1 << 17
OPT_QUAD_POINT = 1 << 18
MAX_POINTS = 2**31 - 256          # PSA_MAX_POINTS: the most points one launch takes

# every symbol the header declares, with (restype, argtypes)
_P = C.c_void_p
_SIGS = {
    "psa_device_count": (C.c_int, []),
    "psa_last_error": (C.c_char_p, []),
    "psa_version": (C.c_char_p, []),
    "psa_n_saved": (C.c_int64, [C.c_int64, C.c_int32]),
    "psa_release_cache": (C.c_int, []),
    "psa_traj_ld": (C.c_int64, [C.c_int64, C.c_int32]),
    "psa_rk4_sweep_f64": (C.c_int, [C.c_int, C.c_int, C.c_int64, C.c_int64, C.c_double, C.c_int32,
                                    _P, _P, _P, _P, _P, C.c_uint32, _P, _P, _P, _P, _P, _P]),
    "psa_rk4_sweep_f32": (C.c_int, [C.c_int, C.c_int, C.c_int64, C.c_int64, C.c_double, C.c_int32,
                                    _P, _P, _P, _P, _P, C.c_uint32, _P, _P, _P, _P, _P, _P]),
    "psa_rk4_sweep_f64_dev": (C.c_int, [_P, C.c_int, C.c_int64, C.c_int64, C.c_double, C.c_int32,
                                        _P, _P, _P, _P, _P, C.c_uint32, _P, _P, _P, _P, _P]),
    "psa_rk4_sweep_f32_dev": (C.c_int, [_P, C.c_int, C.c_int64, C.c_int64, C.c_double, C.c_int32,
                                        _P, _P, _P, _P, _P, C.c_uint32, _P, _P, _P, _P, _P]),
    "psa_yaman_rhs_f64": (C.c_int, [C.c_int, C.c_int64, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "psa_gain_summary_f64": (C.c_int, [C.c_int, C.c_int64, _P, _P, C.c_double, C.c_int, _P, _P, _P, _P]),
    "psa_gain_summary_f64_dev": (C.c_int, [_P, C.c_int64, _P, _P, C.c_double, C.c_int, _P, _P, _P, _P, _P]),
    "psa_gain_summary_workspace_bytes": (C.c_int64, [C.c_int64]),
    "psa_gain_summary_f32": (C.c_int, [C.c_int, C.c_int64, _P, _P, C.c_double, C.c_int, _P, _P, _P, _P]),
    "psa_gain_summary_f32_dev": (C.c_int, [_P, C.c_int64, _P, _P, C.c_double, C.c_int, _P, _P, _P, _P, _P]),
    # (stream|device, method, orders, n_orders, max_order, beta, n_beta, omega_ref, two_pi_c, atol, rtol, lambda1,
    #  axis2, n2, axis3, n3, first, n, out, valid)
    "psa_dbeta_grid_f64_dev": (C.c_int, [_P, C.c_int, _P, C.c_int, C.c_int, _P, C.c_int] + [C.c_double] * 5
                               + [_P, C.c_int64, _P, C.c_int64, C.c_int64, C.c_int64, _P, _P]),
    "psa_dbeta_grid_f32_dev": (C.c_int, [_P, C.c_int, _P, C.c_int, C.c_int, _P, C.c_int] + [C.c_double] * 5
                               + [_P, C.c_int64, _P, C.c_int64, C.c_int64, C.c_int64, _P, _P]),
    "psa_dbeta_grid_f64": (C.c_int, [C.c_int, C.c_int, _P, C.c_int, C.c_int, _P, C.c_int] + [C.c_double] * 5
                           + [_P, C.c_int64, _P, C.c_int64, C.c_int64, C.c_int64, _P, _P]),
    # (stream|device, orders, n_orders, beta, n_beta, omega_d, axis1, n1, axis2, n2, first, n, out1, out2)
    "psa_dbeta_pairs_f64_dev": (C.c_int, [_P, _P, C.c_int, _P, C.c_int, C.c_double, _P, C.c_int64, _P, C.c_int64,
                                          C.c_int64, C.c_int64, _P, _P]),
    "psa_dbeta_pairs_f32_dev": (C.c_int, [_P, _P, C.c_int, _P, C.c_int, C.c_double, _P, C.c_int64, _P, C.c_int64,
                                          C.c_int64, C.c_int64, _P, _P]),
    "psa_dbeta_pairs_f64": (C.c_int, [C.c_int, _P, C.c_int, _P, C.c_int, C.c_double, _P, C.c_int64, _P, C.c_int64,
                                      C.c_int64, C.c_int64, _P, _P]),
}
DBETA_SYMMETRIC_EVEN, DBETA_GENERAL_TAYLOR = 0, 1
DBETA_MAX_ORDER = 8
EXPORTED_SYMBOLS = tuple(_SIGS)


class NativeUnavailableError(RuntimeError):
    """libpsa_hip.so cannot be loaded (not built, or the HIP runtime is missing)."""


class PsaNativeError(RuntimeError):
    """A C-ABI call returned non-zero (argument error < 0, hipError_t > 0)."""

    def __init__(self, code: int, message: str):
        super().__init__(f"libpsa_hip rc={code}: {message}")
        self.code = code


_LIB: Optional[C.CDLL] = None


def _pin_hip_runtime() -> None:
    """One HIP runtime per process.  PyTorch-ROCm wheels bundle their own ``libamdhip64.so`` with the same SONAME
    as /opt/rocm's; whichever is mapped first serves every later dlopen of that SONAME.  If torch is installed but
    not imported yet, map ITS copy now, so that a later ``import torch`` (bench.py, the multi-GPU driver) and this
    library share device pointers and streams whatever the import order.  PSA_HIP_RUNTIME=system skips this."""
    import sys
    if "torch" in sys.modules or os.environ.get("PSA_HIP_RUNTIME", "") == "system":
        return
    try:
        import importlib.util
        spec = importlib.util.find_spec("torch")
        if spec is None or not spec.submodule_search_locations:
            return
        cand = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
        if os.path.exists(cand):
            C.CDLL(cand, mode=C.RTLD_GLOBAL)
    except Exception:
        pass  # fall through to the system runtime; a mismatch would surface as a loud HIP error, not a wrong result


def lib() -> C.CDLL:
    """Load the shared library once; raise loudly if it is not there."""
    global _LIB
    if _LIB is None:
        _pin_hip_runtime()
        if not os.path.exists(LIB_PATH):
            raise NativeUnavailableError(
                f"{LIB_PATH} not found: build it with `python __graft_entry__.py` "
                "(or `make -C psa-simulation-ode-rk-mvp-dispersion_amd/csrc`). There is no CPU fallback: "
                "the sweep only runs in the HIP library.")
        try:
            L = C.CDLL(LIB_PATH)
        except OSError as e:  # missing libamdhip64 etc.
            raise NativeUnavailableError(f"cannot load {LIB_PATH}: {e}") from e
        for name, (res, args) in _SIGS.items():
            fn = getattr(L, name)  # AttributeError if the build is stale
            fn.restype = res
            fn.argtypes = args
        _LIB = L
    return _LIB


def device_count() -> int:
    return int(lib().psa_device_count())


def version() -> str:
    return lib().psa_version().decode()


def traj_ld(n_points: int, dtype=np.float64) -> int:
    """Leading dimension of a device trajectory buffer launched with OPT_TRAJ_LD (psa_traj_ld)."""
    return int(lib().psa_traj_ld(int(n_points), int(np.dtype(dtype).itemsize)))


def release_cache() -> int:
    """Destroy the idle per-device call contexts (stream, events, scratch) the host-buffer entry points keep; -> how many."""
    return int(lib().psa_release_cache())


def _check(rc: int) -> None:
    if rc != 0:
        raise PsaNativeError(rc, lib().psa_last_error().decode(errors="replace"))


def _ptr(a: Optional[np.ndarray]):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _prep(x, dtype, n_points: int, name: str):
    """-> (contiguous 1-D array, is_broadcast)."""
    arr = np.ascontiguousarray(np.atleast_1d(np.asarray(x)), dtype=dtype)
    if arr.ndim != 1:
        raise ValueError(f"{name} must be a scalar or 1-D array")
    if arr.shape[0] == n_points and n_points != 1:
        return arr, False
    if arr.shape[0] == 1:
        return arr, True
    raise ValueError(f"{name} must have 1 or {n_points} entries, got {arr.shape[0]}")


def sweep_host(dbeta, *, n_steps: int, z_max: float, save_every: int, gamma, alpha, a0, dbeta2=None,
               check_nan: bool = True, exact_step: Optional[bool] = None, want_traj: bool = False, dtype=np.float64,
               device: int = 0, extra_flags: int = 0) -> dict:
    """Run N independent RK4 propagations on the GPU (host buffers in, host buffers out).

    exact_step: None = exact first_bad_step in float64 (free there: block test + replay of a failing block) and the
    per-save-block index in float32; True / False force either.

    dbeta (N,); gamma/alpha scalar or (N,); a0 (n_waves,) or (N, n_waves) complex.
    Returns a_end (N, n_waves) complex, p_end, p_max (N,), first_bad_step (N,) int64,
    traj (N, n_saved, n_waves) complex or None, elapsed_ms (kernel only).
    """
    dtype = np.dtype(dtype)
    if dtype not in (np.dtype(np.float64), np.dtype(np.float32)):
        raise ValueError("dtype must be float64 or float32")
    cdt = np.complex128 if dtype == np.float64 else np.complex64
    dbeta = np.ascontiguousarray(np.atleast_1d(np.asarray(dbeta)), dtype=dtype)
    if dbeta.ndim != 1:
        raise ValueError("dbeta must be 1-D")
    N = int(dbeta.shape[0])
    a0 = np.ascontiguousarray(np.asarray(a0), dtype=cdt)
    if a0.ndim == 1:
        a0 = a0[None, :]
    if a0.ndim != 2 or a0.shape[1] not in (4, 6):
        raise ValueError("a0 must have shape (n_waves,) or (N, n_waves) with n_waves in (4, 6)")
    nw = int(a0.shape[1])
    flags = int(extra_flags)
    if a0.shape[0] == 1:
        flags |= BCAST_A0
    elif a0.shape[0] != N:
        raise ValueError(f"a0 must have 1 or {N} rows, got {a0.shape[0]}")
    gamma, gb = _prep(gamma, dtype, N, "gamma")
    alpha, ab = _prep(alpha, dtype, N, "alpha")
    if gb:
        flags |= BCAST_GAMMA
    if ab:
        flags |= BCAST_ALPHA
    if check_nan:
        flags |= OPT_CHECK_NAN
        if exact_step or (exact_step is None and dtype == np.float64):
            flags |= OPT_EXACT_STEP
    d2 = None
    if nw == 6:
        if dbeta2 is None:
            raise ValueError("n_waves == 6 needs dbeta2")
        d2 = np.ascontiguousarray(np.atleast_1d(np.asarray(dbeta2)), dtype=dtype)
        if d2.shape != dbeta.shape:
            raise ValueError("dbeta2 must match dbeta")
    elif dbeta2 is not None:
        raise ValueError("dbeta2 is only meaningful for 6 waves")

    n_saved = int(n_steps) // int(save_every) + 1 if save_every > 0 else 0
    a_end = np.empty((N, nw), dtype=cdt)
    p_end = np.empty(N, dtype=dtype)
    p_max = np.empty(N, dtype=dtype)
    bad = np.empty(N, dtype=np.int64)
    traj = np.empty((N, n_saved, nw), dtype=cdt) if want_traj else None
    ms = C.c_double(0.0)
    fn = lib().psa_rk4_sweep_f64 if dtype == np.float64 else lib().psa_rk4_sweep_f32
    _check(fn(int(device), nw, N, int(n_steps), float(z_max), int(save_every), _ptr(dbeta), _ptr(d2), _ptr(gamma),
              _ptr(alpha), _ptr(a0), flags, _ptr(a_end), _ptr(p_end), _ptr(p_max), _ptr(bad), _ptr(traj),
              C.cast(C.byref(ms), C.c_void_p)))
    return dict(a_end=a_end, p_end=p_end, p_max=p_max, first_bad_step=bad, traj=traj, elapsed_ms=ms.value)


def sweep_device(*, stream: int, n_waves: int, n_points: int, n_steps: int, z_max: float, save_every: int,
                 d_dbeta: int, d_dbeta2: int, d_gamma: int, d_alpha: int, d_a0_soa: int, flags: int,
                 d_a_end_soa: int, d_p_end: int, d_p_max: int, d_first_bad: int, d_traj_soa: int = 0,
                 dtype=np.float64) -> None:
    """Asynchronous launch on device pointers (ints), SoA layout -- see psa_rk4_sweep_f64_dev."""
    fn = lib().psa_rk4_sweep_f64_dev if np.dtype(dtype) == np.float64 else lib().psa_rk4_sweep_f32_dev
    _check(fn(stream or None, int(n_waves), int(n_points), int(n_steps), float(z_max), int(save_every),
              d_dbeta or None, d_dbeta2 or None, d_gamma or None, d_alpha or None, d_a0_soa or None, int(flags),
              d_a_end_soa or None, d_p_end or None, d_p_max or None, d_first_bad or None, d_traj_soa or None))


def yaman_rhs_host(z, a, gamma, alpha, dbeta, *, terms: bool = False, device: int = 0):
    """Batched yaman_model.rhs_yaman_simplified on the GPU: a (N,4) complex -> (N,4) complex."""
    a = np.ascontiguousarray(np.asarray(a), dtype=np.complex128)
    if a.ndim == 1:
        a = a[None, :]
    if a.ndim != 2 or a.shape[1] != 4:
        raise ValueError("a_arr must have shape (4,)")
    N = a.shape[0]
    bc = lambda x: np.ascontiguousarray(np.broadcast_to(np.asarray(x, dtype=np.float64), (N,)))  # noqa: E731
    z, gamma, alpha, dbeta = bc(z), bc(gamma), bc(alpha), bc(dbeta)
    out = np.empty((N, 4), dtype=np.complex128)
    parts = [np.empty((N, 4), dtype=np.complex128) for _ in range(3)] if terms else [None, None, None]
    _check(lib().psa_yaman_rhs_f64(int(device), N, _ptr(z), _ptr(a), _ptr(gamma), _ptr(alpha), _ptr(dbeta),
                                   _ptr(out), _ptr(parts[0]), _ptr(parts[1]), _ptr(parts[2])))
    return (out, *parts) if terms else out


def gain_summary_host(p_metric, first_bad_step, p0_sig: float, *, gain_db: bool = True, device: int = 0):
    """Per-point gain (NaN on failure) + (argmax, max, #finite) over the sweep, reduced on the GPU.
    A float32 ``p_metric`` (a float32 sweep) goes through ``psa_gain_summary_f32`` and returns float32 gains."""
    p = np.asarray(p_metric)
    f32 = p.dtype == np.float32
    p = np.ascontiguousarray(p, dtype=np.float32 if f32 else np.float64)
    bad = None if first_bad_step is None else np.ascontiguousarray(np.asarray(first_bad_step), dtype=np.int64)
    N = p.shape[0]
    gain = np.empty(N, dtype=p.dtype)
    bi = C.c_int64(-1)
    bg = C.c_double(float("nan"))
    nf = C.c_int64(0)
    fn = lib().psa_gain_summary_f32 if f32 else lib().psa_gain_summary_f64
    _check(fn(int(device), N, _ptr(p), _ptr(bad), float(p0_sig), int(bool(gain_db)), _ptr(gain),
              C.cast(C.byref(bi), _P), C.cast(C.byref(bg), _P), C.cast(C.byref(nf), _P)))
    return gain, int(bi.value), float(bg.value), int(nf.value)


def gain_summary_device(*, stream: int, n_points: int, d_p_metric: int, d_first_bad: int, p0_sig: float, gain_db: bool,
                        d_gain: int, d_best_index: int, d_best_gain: int, d_n_finite: int, d_workspace: int,
                        dtype=np.float64) -> None:
    """Asynchronous gain reduction on device pointers (ints) -- see psa_gain_summary_f64_dev / _f32_dev."""
    fn = lib().psa_gain_summary_f64_dev if np.dtype(dtype) == np.float64 else lib().psa_gain_summary_f32_dev
    _check(fn(stream or None, int(n_points), d_p_metric or None, d_first_bad or None, float(p0_sig),
              int(bool(gain_db)), d_gain or None, d_best_index or None, d_best_gain or None, d_n_finite or None,
              d_workspace or None))


def gain_summary_workspace_bytes(n_points: int) -> int:
    return int(lib().psa_gain_summary_workspace_bytes(int(n_points)))


# ---- device-side dbeta producer (psa_dbeta_grid_* / psa_dbeta_pairs_*) -------------------------------------------
def dbeta_model(disp, pm_cfg=None, *, even_orders=None) -> dict:
    """The C-ABI's description of (DispersionParams, PhaseMatchingConfig): method, even orders / max order, beta_0..beta_8,
    omega_ref, tolerances.  ``pm_cfg=None`` + ``even_orders`` describes the symmetric closed form alone (six-wave grid).
    Raises ValueError for what the device producer does not cover (PROVIDED, orders above 8)."""
    from .phase_matching import PhaseMatchingMethod
    if disp is None:
        raise ValueError("disp must be provided unless method == 'provided'")
    top = DBETA_MAX_ORDER
    if disp.extra is not None and any(int(k) > top and v != 0.0 for k, v in disp.extra.items()):
        raise ValueError(f"the device dbeta producer covers dispersion orders up to {top}")
    beta = np.array([disp.get_beta_n(n) for n in range(top + 1)], dtype=np.float64)
    if pm_cfg is None:
        method, orders, max_order, atol, rtol = DBETA_SYMMETRIC_EVEN, tuple(even_orders or (2, 4)), 0, 0.0, 1e-12
    elif pm_cfg.method == PhaseMatchingMethod.SYMMETRIC_EVEN:
        method, orders, max_order, atol, rtol = DBETA_SYMMETRIC_EVEN, tuple(pm_cfg.even_orders), 0, pm_cfg.atol, pm_cfg.rtol
    elif pm_cfg.method == PhaseMatchingMethod.GENERAL_TAYLOR:
        method, orders, max_order, atol, rtol = DBETA_GENERAL_TAYLOR, (), int(pm_cfg.max_order), pm_cfg.atol, pm_cfg.rtol
    else:
        raise ValueError("a PROVIDED dbeta is an input, not something to generate")
    if method == DBETA_SYMMETRIC_EVEN and (not 1 <= len(orders) <= 4 or any(n > top for n in orders)):
        raise ValueError(f"the device dbeta producer takes 1..4 even orders up to {top}")
    if method == DBETA_GENERAL_TAYLOR and max_order > top:
        raise ValueError(f"the device dbeta producer covers max_order up to {top}")
    from . import constants
    return dict(method=method, orders=np.asarray(orders, dtype=np.int32), max_order=max_order, beta=beta,
                omega_ref=float(disp.omega_ref), two_pi_c=2.0 * np.pi * constants.c, atol=float(atol), rtol=float(rtol))


def _model_head(m: dict):
    return (int(m["method"]), _ptr(m["orders"]) if m["orders"].size else None, int(m["orders"].size), int(m["max_order"]),
            _ptr(m["beta"]), int(m["beta"].size), m["omega_ref"], m["two_pi_c"], m["atol"], m["rtol"])


def dbeta_grid_host(model: dict, lambda1_m: float, lambda2_axis, lambda3_axis, *, first: int = 0,
                    n_points: Optional[int] = None, device: int = 0):
    """dbeta (and validity) of points [first, first + n_points) of the flattened lambda2 x lambda3 grid, on the GPU."""
    ax2 = np.ascontiguousarray(np.atleast_1d(lambda2_axis), dtype=np.float64)
    ax3 = np.ascontiguousarray(np.atleast_1d(lambda3_axis), dtype=np.float64)
    n = ax2.size * ax3.size - int(first) if n_points is None else int(n_points)
    out = np.empty(n, dtype=np.float64)
    valid = np.empty(n, dtype=np.uint8)
    _check(lib().psa_dbeta_grid_f64(int(device), *_model_head(model), float(lambda1_m), _ptr(ax2), ax2.size, _ptr(ax3),
                                    ax3.size, int(first), n, _ptr(out), _ptr(valid)))
    return out, valid.astype(bool)


def dbeta_grid_device(model: dict, lambda1_m: float, *, stream: int, d_lambda2_axis: int, n2: int, d_lambda3_axis: int,
                      n3: int, first: int, n_points: int, d_dbeta: int, d_valid: int = 0, dtype=np.float64) -> None:
    fn = lib().psa_dbeta_grid_f64_dev if np.dtype(dtype) == np.float64 else lib().psa_dbeta_grid_f32_dev
    _check(fn(stream or None, *_model_head(model), float(lambda1_m), d_lambda2_axis or None, int(n2),
              d_lambda3_axis or None, int(n3), int(first), int(n_points), d_dbeta or None, d_valid or None))


def dbeta_pairs_host(model: dict, omega_d: float, Omega1_axis, Omega2_axis, *, first: int = 0,
                     n_points: Optional[int] = None, device: int = 0):
    ax1 = np.ascontiguousarray(np.atleast_1d(Omega1_axis), dtype=np.float64)
    ax2 = np.ascontiguousarray(np.atleast_1d(Omega2_axis), dtype=np.float64)
    n = ax1.size * ax2.size - int(first) if n_points is None else int(n_points)
    o1, o2 = np.empty(n, dtype=np.float64), np.empty(n, dtype=np.float64)
    _check(lib().psa_dbeta_pairs_f64(int(device), _ptr(model["orders"]), int(model["orders"].size), _ptr(model["beta"]),
                                     int(model["beta"].size), float(omega_d), _ptr(ax1), ax1.size, _ptr(ax2), ax2.size,
                                     int(first), n, _ptr(o1), _ptr(o2)))
    return o1, o2


def dbeta_pairs_device(model: dict, omega_d: float, *, stream: int, d_Omega1_axis: int, n1: int, d_Omega2_axis: int,
                       n2: int, first: int, n_points: int, d_dbeta1: int, d_dbeta2: int, dtype=np.float64) -> None:
    fn = lib().psa_dbeta_pairs_f64_dev if np.dtype(dtype) == np.float64 else lib().psa_dbeta_pairs_f32_dev
    _check(fn(stream or None, _ptr(model["orders"]), int(model["orders"].size), _ptr(model["beta"]),
              int(model["beta"].size), float(omega_d), d_Omega1_axis or None, int(n1), d_Omega2_axis or None, int(n2),
              int(first), int(n_points), d_dbeta1 or None, d_dbeta2 or None))
