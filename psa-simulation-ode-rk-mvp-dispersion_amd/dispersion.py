"""Taylor dispersion model and phase mismatch (host-side feeder of the sweep kernel's dbeta[N]).

Call surface of the reference's dispersion.py: unit converters (:72-99), beta_n from D/S (:102-139),
``DispersionParams`` (:142-230), ``beta_taylor`` (:233-279), ``delta_beta_from_omegas`` (:282-318),
``delta_beta_symmetric`` (:321-372), ``dispersion_params_from_D_S`` (:375-466).  Every function accepts
arrays for the frequency arguments, so one call serves a whole sweep; arithmetic ORDER follows the
reference expression by expression (the results feed exp(i*dbeta*z) over ~1e5 steps, so the last bits
matter for 1e-9 parity).

Known reference behaviour that parity depends on and that is reproduced on purpose:
``dispersion_params_from_D_S`` passes dS/dlambda in the D slot of ``beta4_from_D_S``
(dispersion.py:455), so with dS/dlambda = 0 one gets beta4 = pref * 6 * lambda * S.
"""
from __future__ import annotations

from dataclasses import dataclass
from math import factorial
from typing import Dict, Iterable, Optional, Sequence, Tuple, Union

import numpy as np

from . import constants

_TWO_PI = 2.0 * np.pi
ArrayLike = Union[float, np.ndarray]


def _real(x, name: str) -> float:
    try:
        v = float(x)
    except Exception as e:
        raise TypeError(f"{name}: not a real scalar ({type(x).__name__})") from e
    if not np.isfinite(v):
        raise ValueError(f"{name}: not finite ({v!r})")
    return v


def _positive(x, name: str) -> float:
    v = _real(x, name)
    if v <= 0.0:
        raise ValueError(f"{name}: needs a positive value ({v!r})")
    return v


# ---- engineering units -> SI ---------------------------------------------------------------------------
def D_ps_nm_km_to_SI(D_ps_nm_km: float) -> float:
    return _real(D_ps_nm_km, "D_ps_nm_km") * 1e-6          # ps/(nm km) -> s/m^2


def S_ps_nm2_km_to_SI(S_ps_nm2_km: float) -> float:
    return _real(S_ps_nm2_km, "S_ps_nm2_km") * 1e3         # ps/(nm^2 km) -> s/m^3


def dSdlmbd_ps_nm3_km_to_SI(dSdlmbd_ps_nm3_km: float) -> float:
    return _real(dSdlmbd_ps_nm3_km, "dSdlmbd_ps_nm3_km") * 1e12   # ps/(nm^3 km) -> s/m^4


# ---- beta_n at a reference wavelength --------------------------------------------------------------------
def beta2_from_D(lambda_ref_m: float, D_SI: float) -> float:
    lam, D = _positive(lambda_ref_m, "lambda_ref_m"), _real(D_SI, "D_SI")
    return -((lam * lam) / (_TWO_PI * constants.c)) * D


def beta3_from_D_S(lambda_ref_m: float, D_SI: float, S_SI: float) -> float:
    lam, D, S = _positive(lambda_ref_m, "lambda_ref_m"), _real(D_SI, "D_SI"), _real(S_SI, "S_SI")
    pref = (lam**4) / ((2.0 * np.pi)**2 * constants.c**2)
    return pref * (S + 2.0 * D / lam)


def beta4_from_D_S(lambda_ref_m: float, D_SI: float, S_SI: float, dSdlmbd_SI: float) -> float:
    lam, D = _positive(lambda_ref_m, "lambda_ref_m"), _real(D_SI, "D_SI")
    S, dS = _real(S_SI, "S_SI"), _real(dSdlmbd_SI, "dSdlmbd_SI")
    pref = -(lam**4) / (2.0 * np.pi * constants.c)**3
    return pref * (6 * D + 6 * lam * S + lam**2 * dS)


@dataclass(frozen=True)
class DispersionParams:
    """beta(w) = sum_n beta_n/n! (w - omega_ref)^n ; beta_n in s^n per length unit; missing orders are 0."""
    omega_ref: float
    beta0: float = 0.0
    beta1: float = 0.0
    beta2: float = 0.0
    beta3: float = 0.0
    beta4: float = 0.0
    extra: Optional[Dict[int, float]] = None   # {order: beta_order}; overrides the named fields

    def __post_init__(self) -> None:
        object.__setattr__(self, "omega_ref", _positive(self.omega_ref, "omega_ref"))
        for n in range(5):
            object.__setattr__(self, f"beta{n}", _real(getattr(self, f"beta{n}"), f"beta{n}"))
        if self.extra is not None:
            if not isinstance(self.extra, dict):
                raise TypeError("extra: a dict order -> coefficient, or None")
            cleaned = {}
            for k, v in self.extra.items():
                if not isinstance(k, int):
                    raise TypeError(f"extra: order {k!r} is not an int")
                if k < 0:
                    raise ValueError(f"extra: negative order {k}")
                cleaned[k] = _real(v, f"extra[{k}]")
            object.__setattr__(self, "extra", cleaned)

    def get_beta_n(self, n: int) -> float:
        if not isinstance(n, int):
            raise TypeError("n must be int")
        if n < 0:
            raise ValueError("n must be >= 0")
        if self.extra is not None and n in self.extra:
            return float(self.extra[n])
        return getattr(self, f"beta{n}") if n <= 4 else 0.0

    def available_orders(self) -> Tuple[int, ...]:
        found = {n for n in range(5) if self.get_beta_n(n) != 0.0}
        if self.extra is not None:
            found |= {n for n, v in self.extra.items() if v != 0.0}
        return tuple(sorted(found))

    def scaled(self, length_scale: float) -> "DispersionParams":
        """Coefficients per (length unit / length_scale): every beta_n divided by the scale
        (what simulation.py:126-150 does for km -> m)."""
        s = float(length_scale)
        if s == 1.0:
            return self
        ex = None if self.extra is None else {int(k): float(v) / s for k, v in self.extra.items()}
        return DispersionParams(self.omega_ref, self.beta0 / s, self.beta1 / s, self.beta2 / s, self.beta3 / s,
                                self.beta4 / s, ex)


def beta_taylor(omega: ArrayLike, disp: DispersionParams, *, max_order: int = 4) -> ArrayLike:
    """beta(w) up to ``max_order``; scalar in -> float out, array in -> array out."""
    if not isinstance(max_order, int):
        raise TypeError("max_order must be int")
    if max_order < 0:
        raise ValueError("max_order must be >= 0")
    w = np.asarray(omega, dtype=float)
    if not np.all(np.isfinite(w)):
        raise ValueError("omega: non-finite")
    if np.any(w <= 0.0):
        raise ValueError("omega: must be > 0 rad/s")
    acc = _taylor_sum(w - disp.omega_ref, disp, max_order)
    return float(acc.item()) if np.isscalar(omega) else acc


def _taylor_sum(dw: np.ndarray, disp: DispersionParams, max_order: int) -> np.ndarray:
    acc = np.zeros_like(dw, dtype=float)
    for n in range(max_order + 1):
        bn = disp.get_beta_n(n)
        if bn != 0.0:
            acc = acc + bn * (dw**n) / float(factorial(n))
    return acc


def delta_beta_from_omegas_array(omega: np.ndarray, disp: DispersionParams, *, max_order: int = 4) -> np.ndarray:
    """dbeta = (beta(w3) + beta(w4)) - (beta(w1) + beta(w2)) for omega[..., 4]; no validation."""
    om = np.asarray(omega, dtype=float)
    b = _taylor_sum(om - disp.omega_ref, disp, max_order)
    return (b[..., 2] + b[..., 3]) - (b[..., 0] + b[..., 1])


def delta_beta_from_omegas(omegas: Sequence[float], disp: DispersionParams, *, max_order: int = 4,
                           atol: float = 0.0, rtol: float = 1e-12) -> float:
    om = np.asarray(list(omegas), dtype=float)
    if om.shape != (4,):
        raise ValueError(f"omegas: four entries expected, shape is {om.shape}")
    if not np.all(np.isfinite(om)):
        raise ValueError("omegas: non-finite entry")
    if np.any(om <= 0.0):
        raise ValueError("omegas: entries must be > 0 rad/s")
    lhs, rhs = om[0] + om[1], om[2] + om[3]
    if not np.isclose(lhs, rhs, atol=atol, rtol=rtol):
        raise ValueError(f"w1 + w2 = {lhs:.16e} but w3 + w4 = {rhs:.16e}: photon energy is not conserved")
    if not isinstance(max_order, int):
        raise TypeError("max_order must be int")
    if max_order < 0:
        raise ValueError("max_order must be >= 0")
    # one 0-d evaluation per wave (not the 4-vector kernel): keeps libm's scalar pow, bit-for-bit with the reference
    b1, b2, b3, b4 = (beta_taylor(om[j], disp, max_order=max_order) for j in range(4))
    return float((b3 + b4) - (b1 + b2))


def _check_even_orders(even_orders: Iterable[int]) -> list:
    evens = list(even_orders)
    if not evens:
        raise ValueError("even_orders: empty")
    for n in evens:
        if not isinstance(n, int):
            raise TypeError("even_orders: ints only")
        if n < 2:
            raise ValueError(f"even_orders: {n} < 2")
        if n % 2:
            raise ValueError(f"even_orders: {n} is odd")
    return evens


def delta_beta_symmetric_array(omega_d, Omega, disp: DispersionParams, *, even_orders: Iterable[int] = (2, 4)):
    """dbeta = sum_{n even} beta_n * (Omega^n - omega_d^n) * 2 / n!  for arrays (or scalars) of
    (omega_d, Omega); accumulated in the order given, as dispersion.py:365-370 does."""
    od = np.asarray(omega_d, dtype=float)
    Om = np.asarray(Omega, dtype=float)
    acc = np.zeros(np.broadcast(od, Om).shape, dtype=float)
    for n in _check_even_orders(even_orders):
        bn = disp.get_beta_n(n)
        if bn != 0.0:
            acc = acc + bn * (Om**n - od**n) * 2.0 / float(factorial(n))
    return acc


def delta_beta_symmetric(omega_c: float, omega_d: float, Omega: float, disp: DispersionParams, *,
                         even_orders: Iterable[int] = (2, 4)) -> float:
    _positive(omega_c, "omega_c")   # disp.omega_ref is NOT required to equal omega_c (dispersion.py:349-352)
    od, Om = _real(omega_d, "omega_d"), _real(Omega, "Omega")
    # Python-float arithmetic, exactly as the reference's scalar loop (x**n on floats)
    total = 0.0
    for n in _check_even_orders(even_orders):
        bn = disp.get_beta_n(n)
        if bn != 0.0:
            total += bn * (Om**n - od**n) * 2.0 / float(factorial(n))
    return float(total)


def dispersion_params_from_D_S(lambda_ref_m: float, D: float, S: Optional[float] = None,
                               dSdlmbd: Optional[float] = None, *, D_units: str = "SI", S_units: str = "SI",
                               dSdlmbd_units: str = "SI", omega_ref: Optional[float] = None, beta0: float = 0.0,
                               beta1: float = 0.0, extra: Optional[Dict[int, float]] = None) -> DispersionParams:
    """DispersionParams at ``lambda_ref_m`` from D, S, dS/dlambda (units "SI" or the ps/nm/km family)."""
    lam = _positive(lambda_ref_m, "lambda_ref_m")
    wref = _TWO_PI * constants.c / lam if omega_ref is None else _positive(omega_ref, "omega_ref")

    def to_si(value, units, name, si_fn, eng_units):
        if value is None:
            return 0
        if units == "SI":
            return _real(value, name)
        if units == eng_units:
            return si_fn(value)
        raise ValueError(f"Unknown {name}_units={units!r}. Use 'SI' or {eng_units!r}.")

    if D_units not in ("SI", "ps/nm/km"):
        raise ValueError(f"Unknown D_units={D_units!r}. Use 'SI' or 'ps/nm/km'.")
    D_SI = _real(D, "D") if D_units == "SI" else D_ps_nm_km_to_SI(D)   # D is mandatory
    S_SI = to_si(S, S_units, "S", S_ps_nm2_km_to_SI, "ps/nm^2/km")
    dS_SI = to_si(dSdlmbd, dSdlmbd_units, "dSdlmbd", dSdlmbd_ps_nm3_km_to_SI, "ps/nm^3/km")
    return DispersionParams(
        omega_ref=wref, beta0=beta0, beta1=beta1,
        beta2=beta2_from_D(lam, D_SI),
        beta3=beta3_from_D_S(lam, D_SI, S_SI),
        # reference quirk kept for parity (dispersion.py:455): dS/dlambda sits in the D slot
        beta4=beta4_from_D_S(lam, dS_SI, S_SI, dS_SI),
        extra=extra)
