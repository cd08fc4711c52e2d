"""Fixed-step RK4 integration -- the operator API of the reference's integrators.py
(``RHSFunction`` :18, ``rk4_step`` :25-61, ``integrate_fixed_step`` :68-142, ``integrate_interval`` :150-204).

Dispatch
--------
* ``f`` is a NATIVE operator handle (``f.native_kind == "yaman4"``, i.e. ``yaman_model.rhs_yaman_simplified``):
  the whole z-loop -- four RHS evaluations per step, save stride, per-step NaN test -- runs inside ONE launch of
  the HIP kernel ``psa_rk4_sweep_f64`` with N = 1 and the trajectory output enabled.  No Python code runs per
  step.  If ``libpsa_hip.so`` or a GPU is missing this raises; it never degrades to a host loop.
* ``f`` is an arbitrary Python callable (what the reference's own tests pass: ``lambda z, y, p: y``): the callback
  has to execute on the host by definition, so the stepping loop around it does too (``_step_callable`` below).
  This is the reference's generic-operator contract, not a fallback for the Yaman path.

Semantics kept from the reference: n = int(round(z_max/dz)) and the grid np.linspace(0, z_max, n+1) (R7);
rows only at multiples of ``save_every`` plus z = 0, ``n_saved = n // save_every + 1`` (R8); ValueError for
bad z_max / dz / save_every / z_grid.ndim; FloatingPointError("NaN or Inf detected at step {i}, z = {z}")
when ``check_nan`` and the state after step i is not finite; silent NaNs otherwise.
"""
from __future__ import annotations

from typing import Callable, Tuple

import numpy as np

from . import _native

RHSFunction = Callable[[float, np.ndarray, object], np.ndarray]

__all__ = ["RHSFunction", "rk4_step", "integrate_fixed_step", "integrate_interval"]


def _is_native(f) -> bool:
    return getattr(f, "native_kind", None) == "yaman4"


def rk4_step(f: RHSFunction, z: float, y: np.ndarray, dz: float, params: object) -> np.ndarray:
    """One classic RK4 step; ``f`` may be a Python callable or the native RHS handle (4 batched-kernel calls)."""
    half = 0.5 * dz
    k1 = f(z, y, params)
    k2 = f(z + half, y + half * k1, params)
    k3 = f(z + half, y + half * k2, params)
    k4 = f(z + dz, y + dz * k3, params)
    return y + (dz / 6.0) * (k1 + 2.0 * k2 + 2.0 * k3 + k4)


def _step_callable(f, z_grid, y0, params, save_every, check_nan):
    """Host stepping loop for user-supplied Python right-hand sides (see module docstring)."""
    n = len(z_grid) - 1
    rows = n // save_every + 1
    z_out = np.empty(rows, dtype=float)
    y_out = np.empty((rows, y0.size), dtype=y0.dtype)
    y = y0.copy()
    z_out[0], y_out[0] = z_grid[0], y
    r = 1
    for i in range(n):
        y = rk4_step(f, z_grid[i], y, z_grid[i + 1] - z_grid[i], params)
        if check_nan and not np.all(np.isfinite(y)):
            raise FloatingPointError(f"NaN or Inf detected at step {i}, z = {z_grid[i]}")
        if (i + 1) % save_every == 0:
            z_out[r], y_out[r] = z_grid[i + 1], y
            r += 1
    return z_out[:r], y_out[:r]


def _run_native(f, z_max: float, n_steps: int, y0, params, save_every: int, check_nan: bool):
    """N = 1 launch of the sweep kernel with every saved row written out."""
    from .yaman_model import extract_gamma_alpha_dbeta
    gamma, alpha, dbeta = extract_gamma_alpha_dbeta(params)
    a0 = np.asarray(y0)
    if a0.shape != (4,):
        raise ValueError("a_arr must have shape (4,)")
    res = _native.sweep_host(np.array([dbeta]), n_steps=n_steps, z_max=z_max, save_every=save_every, gamma=gamma,
                             alpha=alpha, a0=a0.astype(np.complex128), check_nan=check_nan, exact_step=True,
                             want_traj=True)
    z_grid = np.linspace(0.0, z_max, n_steps + 1)
    bad = int(res["first_bad_step"][0])
    if check_nan and bad >= 0:
        raise FloatingPointError(f"NaN or Inf detected at step {bad}, z = {z_grid[bad]}")
    rows = res["traj"][0]
    z_out = z_grid[::save_every][: rows.shape[0]].copy()
    if not np.iscomplexobj(a0):  # the reference stores into an array of y0's dtype
        rows = rows.astype(a0.dtype)
    return z_out, rows


def _uniform_from_zero(z_grid: np.ndarray) -> bool:
    n = len(z_grid) - 1
    return n >= 1 and z_grid[0] == 0.0 and z_grid[-1] > 0.0 and \
        np.array_equal(z_grid, np.linspace(0.0, z_grid[-1], n + 1))


def integrate_fixed_step(f: RHSFunction, z_grid: np.ndarray, y0: np.ndarray, params: object, *,
                         save_every: int = 1, check_nan: bool = True) -> Tuple[np.ndarray, np.ndarray]:
    """RK4 over a given monotone grid -> (z_out[n_saved], y_out[n_saved, state_dim])."""
    z_grid = np.asarray(z_grid, dtype=float)
    if z_grid.ndim != 1:
        raise ValueError("z_grid must be a one-dimensional array")
    if save_every <= 0:
        raise ValueError("save_every must be a positive integer")
    y0 = np.asarray(y0)
    if _is_native(f) and _uniform_from_zero(z_grid):
        return _run_native(f, float(z_grid[-1]), len(z_grid) - 1, y0, params, int(save_every), bool(check_nan))
    # arbitrary grids / arbitrary callables: the callback contract (for the native handle each stage is one
    # call of the batched HIP RHS kernel)
    return _step_callable(f, z_grid, y0, params, int(save_every), bool(check_nan))


def integrate_interval(f: RHSFunction, z_max: float, dz: float, y0: np.ndarray, params: object, *,
                       save_every: int = 1, check_nan: bool = True) -> Tuple[np.ndarray, np.ndarray]:
    """RK4 on [0, z_max] with nominal step dz (effective step z_max / round(z_max/dz))."""
    if z_max <= 0.0:
        raise ValueError("z_max must be positive")
    if dz <= 0.0:
        raise ValueError("dz must be positive")
    if save_every <= 0:
        raise ValueError("save_every must be a positive integer")
    n_steps = int(round(z_max / dz))
    if _is_native(f):
        if n_steps < 1:
            return np.zeros(1), np.asarray(y0)[None, :].copy()
        return _run_native(f, float(z_max), n_steps, np.asarray(y0), params, int(save_every), bool(check_nan))
    return integrate_fixed_step(f, np.linspace(0.0, z_max, n_steps + 1), np.asarray(y0), params,
                                save_every=save_every, check_nan=check_nan)
