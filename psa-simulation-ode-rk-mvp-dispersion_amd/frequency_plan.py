"""Frequency plans for dual-pump FWM: [pump1, pump2, signal, idler] <-> [w1, w2, w3, w4].

Scalar call surface of the reference's frequency_plan.py (conversions :77-98, energy check :112-131,
SymmetricPlan :134-199, builders :202-327) re-expressed on top of ARRAY kernels, because the sweep
drivers here build the plan of every sweep point at once (``plan_from_wavelengths_batch``) instead of
once per Python loop iteration (scan_mismtach.py:360, :697).  A scalar call is the N = 1 case and raises
what the reference raises; the batch call returns a validity mask instead (the drivers turn invalid
points into NaN, scan_mismtach.py:391-392).
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Optional, Tuple

import numpy as np

from . import constants

_TWO_PI = 2.0 * np.pi
_LABELS = ("pump1", "pump2", "signal", "idler")


def _finite_scalar(x, name: str) -> float:
    try:
        v = float(x)
    except Exception as e:
        raise TypeError(f"{name}: not a real scalar ({type(x).__name__})") from e
    if not np.isfinite(v):
        raise ValueError(f"{name}: not finite ({v!r})")
    return v


def _positive_scalar(x, name: str, unit: str) -> float:
    v = _finite_scalar(x, name)
    if v <= 0.0:
        raise ValueError(f"{name}: needs a positive value in {unit} ({v!r})")
    return v


# ---- conversions (frequency_plan.py:77-98) -----------------------------------------------------
def omega_from_f(f_hz: float) -> float:
    return _TWO_PI * _positive_scalar(f_hz, "f_hz", "Hz")


def f_from_omega(omega: float) -> float:
    return _positive_scalar(omega, "omega", "rad/s") / _TWO_PI


def omega_from_lambda(lambda_m: float) -> float:
    return _TWO_PI * constants.c / _positive_scalar(lambda_m, "lambda_m", "m")


def lambda_from_omega(omega: float) -> float:
    return _TWO_PI * constants.c / _positive_scalar(omega, "omega", "rad/s")


def omega_from_lambda_array(lambda_m) -> np.ndarray:
    """Vector form of ``omega_from_lambda`` (no validation; same operation order: (2*pi*c)/lambda)."""
    return _TWO_PI * constants.c / np.asarray(lambda_m, dtype=float)


def _omega4(om) -> np.ndarray:
    arr = np.asarray(list(om), dtype=float)
    if arr.shape != (4,):
        raise ValueError(f"omega: four entries expected, shape is {arr.shape}")
    if not np.all(np.isfinite(arr)):
        raise ValueError("omega: non-finite entry")
    if np.any(arr <= 0.0):
        raise ValueError("omega: entries must be > 0 rad/s")
    return arr


def _conserves(lhs, rhs, atol, rtol):
    # np.isclose(lhs, rhs): |lhs - rhs| <= atol + rtol * |rhs|
    return np.abs(lhs - rhs) <= (atol + rtol * np.abs(rhs))


def enforce_energy_conservation(omega, *, atol: float = 0.0, rtol: float = 1e-12) -> None:
    """w1 + w2 == w3 + w4 within tolerance, else ValueError (frequency_plan.py:112-131)."""
    om = _omega4(omega)
    lhs, rhs = om[0] + om[1], om[2] + om[3]
    if not bool(_conserves(lhs, rhs, atol, rtol)):
        raise ValueError(f"w1 + w2 = {lhs:.16e} but w3 + w4 = {rhs:.16e} (difference {lhs - rhs:.3e}): "
                         "photon energy is not conserved by this plan")


# ---- symmetric representation (frequency_plan.py:134-199) ------------------------------------------
@dataclass(frozen=True)
class SymmetricPlan:
    """(omega_c, omega_d, Omega): w1,2 = omega_c +- omega_d ; w3,4 = omega_c +- Omega  [rad/s]."""
    omega_c: float
    omega_d: float
    Omega: float

    def __post_init__(self) -> None:
        oc = _positive_scalar(self.omega_c, "omega_c", "rad/s")
        od = _finite_scalar(self.omega_d, "omega_d")
        Om = _finite_scalar(self.Omega, "Omega")
        if abs(od) >= oc:
            raise ValueError(f"|omega_d| = {abs(od)!r} >= omega_c = {oc!r}: a pump frequency would be <= 0")
        for k, v in (("omega_c", oc), ("omega_d", od), ("Omega", Om)):
            object.__setattr__(self, k, v)

    omega1 = property(lambda self: self.omega_c + self.omega_d)
    omega2 = property(lambda self: self.omega_c - self.omega_d)
    omega3 = property(lambda self: self.omega_c + self.Omega)
    omega4 = property(lambda self: self.omega_c - self.Omega)

    def omegas(self) -> np.ndarray:
        om = np.array([self.omega1, self.omega2, self.omega3, self.omega4], dtype=float)
        if np.any(om <= 0.0):
            raise ValueError(f"signal/idler frequency <= 0 for this (omega_c, Omega): {om.tolist()}")
        enforce_energy_conservation(om)
        return om


def plan_from_symmetry(omega_c: float, omega_d: float, Omega: float) -> np.ndarray:
    return SymmetricPlan(omega_c=omega_c, omega_d=omega_d, Omega=Omega).omegas()


# ---- array kernels -------------------------------------------------------------------------------------
def symmetry_arrays(w1, w2, w3, w4, *, atol: float = 0.0, rtol: float = 1e-12
                    ) -> Tuple[np.ndarray, np.ndarray, np.ndarray, np.ndarray]:
    """Vector form of ``infer_symmetry_from_omegas`` with w4 GIVEN (frequency_plan.py:238-253).

    Returns (omega_c, omega_d, Omega, valid).  ``valid`` is False wherever the scalar function would raise:
    non-finite / non-positive omegas, energy conservation, |omega_d| >= omega_c, a non-positive
    regenerated omega, or the final omega4 consistency check.
    """
    w1, w2, w3, w4 = (np.asarray(w, dtype=float) for w in (w1, w2, w3, w4))
    with np.errstate(all="ignore"):
        ok = np.isfinite(w1) & np.isfinite(w2) & np.isfinite(w3) & np.isfinite(w4)
        ok &= (w1 > 0.0) & (w2 > 0.0) & (w3 > 0.0) & (w4 > 0.0)
        ok &= _conserves(w1 + w2, w3 + w4, atol, rtol)
        oc = 0.5 * (w1 + w2)
        od = 0.5 * (w1 - w2)
        Om = w3 - oc
        ok &= np.isfinite(oc) & np.isfinite(od) & np.isfinite(Om) & (oc > 0.0) & (np.abs(od) < oc)
        r1, r2, r3, r4 = oc + od, oc - od, oc + Om, oc - Om        # SymmetricPlan.omegas()
        ok &= (r1 > 0.0) & (r2 > 0.0) & (r3 > 0.0) & (r4 > 0.0)
        ok &= _conserves(r1 + r2, r3 + r4, 0.0, 1e-12)
        ok &= _conserves(r4, w4, atol, rtol)
    return oc, od, Om, ok


def plan_from_wavelengths_batch(lambda1_m, lambda2_m, lambda3_m, *, atol: float = 0.0, rtol: float = 1e-12
                                ) -> Tuple[np.ndarray, np.ndarray]:
    """All sweep points at once: (omega[N,4], valid[N]); broadcasting over the three inputs.

    Restates ``plan_from_wavelengths(l1, l2, l3, lambda4_m=None)`` (frequency_plan.py:291-327):
    w_j = 2*pi*c / lambda_j, w4 = w1 + w2 - w3; invalid where a wavelength is non-finite or <= 0,
    where the inferred w4 <= 0, or where energy conservation fails.
    """
    l1, l2, l3 = np.broadcast_arrays(*(np.asarray(x, dtype=float) for x in (lambda1_m, lambda2_m, lambda3_m)))
    with np.errstate(all="ignore"):
        ok = np.isfinite(l1) & np.isfinite(l2) & np.isfinite(l3) & (l1 > 0.0) & (l2 > 0.0) & (l3 > 0.0)
        w1, w2, w3 = (_TWO_PI * constants.c / l for l in (l1, l2, l3))
        w4 = w1 + w2 - w3
        ok &= np.isfinite(w4) & (w4 > 0.0)
        ok &= _conserves(w1 + w2, w3 + w4, atol, rtol)
    return np.stack([w1, w2, w3, w4], axis=-1), ok


# ---- scalar builders (N = 1 views of the kernels above, raising like the reference) ---------------
def infer_symmetry_from_omegas(omega1: float, omega2: float, omega3: float, omega4: Optional[float] = None, *,
                               atol: float = 0.0, rtol: float = 1e-12) -> SymmetricPlan:
    w1 = _positive_scalar(omega1, "omega1", "rad/s")
    w2 = _positive_scalar(omega2, "omega2", "rad/s")
    w3 = _positive_scalar(omega3, "omega3", "rad/s")
    if omega4 is None:
        w4 = _positive_scalar(w1 + w2 - w3, "omega4(inferred)", "rad/s")
    else:
        w4 = _positive_scalar(omega4, "omega4", "rad/s")
        enforce_energy_conservation(np.array([w1, w2, w3, w4]), atol=atol, rtol=rtol)
    oc = 0.5 * (w1 + w2)
    sp = SymmetricPlan(omega_c=oc, omega_d=0.5 * (w1 - w2), Omega=w3 - oc)
    back = sp.omegas()
    if not bool(_conserves(back[3], w4, atol, rtol)):
        raise ValueError(f"symmetric form gives w4 = {back[3]:.16e}, expected {w4:.16e}")
    return sp


def plan_from_omegas(omega1: float, omega2: float, omega3: float, omega4: Optional[float] = None, *,
                     atol: float = 0.0, rtol: float = 1e-12) -> np.ndarray:
    w1 = _positive_scalar(omega1, "omega1", "rad/s")
    w2 = _positive_scalar(omega2, "omega2", "rad/s")
    w3 = _positive_scalar(omega3, "omega3", "rad/s")
    w4 = _positive_scalar(w1 + w2 - w3 if omega4 is None else omega4,
                          "omega4(inferred)" if omega4 is None else "omega4", "rad/s")
    om = np.array([w1, w2, w3, w4], dtype=float)
    enforce_energy_conservation(om, atol=atol, rtol=rtol)
    return om


def plan_from_wavelengths(lambda1_m: float, lambda2_m: float, lambda3_m: float, lambda4_m: Optional[float] = None, *,
                          atol: float = 0.0, rtol: float = 1e-12) -> np.ndarray:
    w1 = omega_from_lambda(_positive_scalar(lambda1_m, "lambda1_m", "m"))
    w2 = omega_from_lambda(_positive_scalar(lambda2_m, "lambda2_m", "m"))
    w3 = omega_from_lambda(_positive_scalar(lambda3_m, "lambda3_m", "m"))
    if lambda4_m is None:
        w4 = _positive_scalar(w1 + w2 - w3, "omega4(inferred)", "rad/s")
    else:
        w4 = omega_from_lambda(_positive_scalar(lambda4_m, "lambda4_m", "m"))
    om = np.array([w1, w2, w3, w4], dtype=float)
    enforce_energy_conservation(om, atol=atol, rtol=rtol)
    return om


def describe_plan(omega) -> str:
    om = _omega4(omega)
    rows = ["Frequency plan (wave order: pump1, pump2, signal, idler):"]
    for lab, w in zip(_LABELS, om):
        rows.append(f"  {lab:6s}: omega={w: .16e} rad/s, f={f_from_omega(w): .16e} Hz, "
                    f"lambda={lambda_from_omega(w): .16e} m")
    rows.append(f"  Check: omega1+omega2 - (omega3+omega4) = {(om[0] + om[1]) - (om[2] + om[3]): .16e} rad/s")
    return "\n".join(rows)
