"""The batched engine behind the sweep drivers: N independent propagations in one kernel launch.

The reference runs a sweep as a Python ``for`` over points, each calling ``run_single_simulation``
(scan_mismtach.py:357-392, :694-738).  Here the per-point scalars become arrays (dbeta[N], optionally
gamma[N], alpha[N], A0[N,4]) and the loop becomes the grid of ``psa_rk4_sweep_f64``: one sweep point per
lane, the z-loop inside the kernel, only the summary (A_end, |A3|^2 at the last saved row, max over saved
rows, first non-finite step) written back.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Optional, Sequence

import numpy as np

from . import _native
from .config import n_steps_of

__all__ = ["SweepResult", "initial_amplitudes", "rk4_sweep"]


def initial_amplitudes(p_in, phase_in=None) -> np.ndarray:
    """A0 = sqrt(P) * exp(i*phi) for p_in (..., n_waves); the phase factor is skipped when every phase is 0
    (simulation.py:120-123, so sqrt(P) stays bit-exact in the common seed-phase-zero case)."""
    p = np.asarray(p_in, dtype=float)
    amp = np.sqrt(p).astype(np.complex128)
    if phase_in is not None:
        ph = np.asarray(phase_in, dtype=float)
        if np.any(ph != 0.0):
            amp = amp * np.exp(1j * ph)
    return amp


@dataclass
class SweepResult:
    a_end: np.ndarray            # (N, n_waves) complex: state at the last SAVED row
    p_end: np.ndarray            # (N,) |A_signal|^2 there
    p_max: np.ndarray            # (N,) max over saved rows (z = 0 included), NaN-propagating
    first_bad_step: np.ndarray   # (N,) int64, -1 = finite everywhere (or check_nan off)
    n_steps: int
    save_every: int
    elapsed_ms: float            # kernel time (hipEvents)
    traj: Optional[np.ndarray] = None   # (N, n_saved, n_waves) when requested

    def gain(self, p0_sig: float, *, mode: str = "max", unit: str = "dB", device: int = 0) -> np.ndarray:
        """Per-point signal gain with the drivers' NaN rules (scan_mismtach.py:376-392); reduced on the GPU."""
        return self.summary(p0_sig, mode=mode, unit=unit, device=device)[0]

    def summary(self, p0_sig: float, *, mode: str = "max", unit: str = "dB", device: int = 0):
        """(gain[N], best_index, best_gain, n_finite) -- gain_mode "end" | "max" (scan_mismtach.py:27-40)."""
        if mode not in ("end", "max"):
            raise ValueError(f"Unknown gain_mode={mode!r}. Use 'end' or 'max'.")
        u = str(unit).strip().lower()
        if u not in ("db", "linear"):
            raise ValueError("gain_unit must be 'dB' or 'linear'")
        metric = self.p_max if mode == "max" else self.p_end
        # a float32 sweep keeps float32 gains (psa_gain_summary_f32); everything else is reduced in float64
        metric = np.asarray(metric)
        if metric.dtype != np.float32:
            metric = metric.astype(np.float64, copy=False)
        return _native.gain_summary_host(metric, self.first_bad_step, float(p0_sig), gain_db=(u == "db"), device=device)


def _cut(x, lo: int, hi: int, n: int, per_point_ndim: int):
    """The [lo, hi) block of a per-point argument (leading dimension n); scalars / single rows pass through."""
    if x is None:
        return None
    x = np.asarray(x)
    return x[lo:hi] if (x.ndim == per_point_ndim and x.shape[0] == n and n > 1) else x


def _sweep_over_devices(devices, dbeta, **kw) -> dict:
    """One host thread per device, each integrating a contiguous block through ``psa_rk4_sweep_*(device=k)`` (ctypes
    drops the GIL for the duration of the call; the C-ABI is thread-safe for distinct devices).  Blocks are the same
    split the multi-process path uses (``distributed.shard_bounds``): the first ``N % len(devices)`` get one point more."""
    from concurrent.futures import ThreadPoolExecutor
    n, k = int(dbeta.shape[0]), len(devices)
    base, rem = divmod(n, k)
    bounds, lo = [], 0
    for r in range(k):
        hi = lo + base + (1 if r < rem else 0)
        bounds.append((lo, hi))
        lo = hi

    def run(r):
        lo, hi = bounds[r]
        if hi == lo:
            return None
        sub = dict(kw)
        for name, nd in (("gamma", 1), ("alpha", 1), ("dbeta2", 1), ("a0", 2)):
            sub[name] = _cut(kw.get(name), lo, hi, n, nd)
        return _native.sweep_host(dbeta[lo:hi], device=int(devices[r]), **sub)

    with ThreadPoolExecutor(max_workers=k) as pool:
        parts = [p for p in pool.map(run, range(k)) if p is not None]
    out = {key: np.concatenate([p[key] for p in parts]) for key in ("a_end", "p_end", "p_max", "first_bad_step")}
    out["traj"] = np.concatenate([p["traj"] for p in parts]) if parts[0]["traj"] is not None else None
    out["elapsed_ms"] = max(p["elapsed_ms"] for p in parts)
    return out


def rk4_sweep(dbeta, *, z_max: float, dz: Optional[float] = None, n_steps: Optional[int] = None,
              save_every: int = 10, check_nan: bool = True, gamma, alpha, a0, dbeta2=None, dtype=np.float64,
              device: int = 0, exact_step: Optional[bool] = None, want_traj: bool = False,
              devices: Optional[Sequence[int]] = None) -> SweepResult:
    """Propagate N points.  ``dz`` gives n = int(round(z_max/dz)) as integrators.py:194; or pass ``n_steps``.
    ``devices=[0, 1, ...]`` splits the points over several GPUs of this process (one thread per device)."""
    if z_max <= 0.0:
        raise ValueError("z_max must be positive")
    if n_steps is None:
        if dz is None or dz <= 0.0:
            raise ValueError("dz must be positive")
        n_steps = n_steps_of(z_max, dz)
    if save_every <= 0:
        raise ValueError("save_every must be a positive integer")
    if n_steps < 1:
        raise ValueError("z_max / dz rounds to zero steps")
    kw = dict(n_steps=int(n_steps), z_max=float(z_max), save_every=int(save_every), gamma=gamma, alpha=alpha, a0=a0,
              dbeta2=dbeta2, check_nan=check_nan, exact_step=exact_step, want_traj=want_traj, dtype=dtype)
    devs = None if devices is None else [int(d) for d in devices]
    if devs is not None and len(devs) == 0:
        raise ValueError("devices must name at least one GPU")
    db = np.atleast_1d(np.asarray(dbeta))
    if devs is not None and len(devs) > 1 and db.ndim == 1 and db.shape[0] > 1:
        r = _sweep_over_devices(devs, db, **kw)
    else:
        r = _native.sweep_host(dbeta, device=(devs[0] if devs else device), **kw)
    return SweepResult(r["a_end"], r["p_end"], r["p_max"], r["first_bad_step"], int(n_steps), int(save_every),
                       r["elapsed_ms"], r["traj"])
