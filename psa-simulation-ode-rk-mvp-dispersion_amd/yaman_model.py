"""The 4-wave Agrawal-Yaman right-hand side as an OPERATOR HANDLE.

In the reference, ``rhs_yaman_simplified(z, a_arr, params)`` (yaman_model.py:10-52) is a Python function
that the integrator calls four times per step.  Here the same name is a callable object that

* carries ``native_kind = "yaman4"``: ``integrators.integrate_interval`` / ``integrate_fixed_step``
  recognise it and run the WHOLE z-loop inside one HIP kernel (psa_rk4_sweep_f64) instead of calling back;
* can still be called like the reference function (single point or a batch of points): the evaluation runs
  in the batched HIP RHS kernel ``psa_yaman_rhs_f64`` -- there is no NumPy implementation of the physics in
  this package.

    dA_j/dz = -(alpha/2) A_j + i*gamma*(|A_j|^2 + 2*sum_{k!=j}|A_k|^2) A_j + 2i*gamma*(FWM term)*exp(+-i*dbeta*z)
"""
from __future__ import annotations

from typing import Tuple

import numpy as np

from . import _native

__all__ = ["rhs_yaman_simplified", "extract_gamma_alpha_dbeta", "yaman_terms"]


def extract_gamma_alpha_dbeta(params) -> Tuple[float, float, float]:
    """(gamma, alpha, dbeta) from a ModelParams-like object, with the reference's fallbacks
    (yaman_model.py:59-116): gamma_W_m | gamma; alpha_1_m | alpha | 0; cache.delta_beta_1_m, else
    beta_legacy_1_m | beta -> (b3 + b4) - (b1 + b2)."""
    if not hasattr(params, "fiber"):
        raise ValueError("params must have attribute 'fiber'")
    fiber = params.fiber
    for name in ("gamma_W_m", "gamma"):
        if hasattr(fiber, name):
            gamma = float(getattr(fiber, name))
            break
    else:
        raise ValueError("Fiber parameters must contain gamma_W_m (new) or gamma (legacy).")
    alpha = 0.0
    for name in ("alpha_1_m", "alpha"):
        if hasattr(fiber, name):
            alpha = float(getattr(fiber, name))
            break
    cache = getattr(params, "cache", None)
    dbeta = getattr(cache, "delta_beta_1_m", None) if cache is not None else None
    if dbeta is None:
        betas = None
        for name in ("beta_legacy_1_m", "beta"):
            if getattr(fiber, name, None) is not None:
                betas = np.asarray(getattr(fiber, name), dtype=float)
                break
        if betas is None:
            raise ValueError("Phase mismatch dbeta is not available. Expected params.cache.delta_beta_1_m to be set "
                             "(preferred), or fiber.beta_legacy_1_m / fiber.beta to exist for fallback.")
        if betas.shape != (4,):
            raise ValueError("Fallback betas must have shape (4,)")
        dbeta = float((betas[2] + betas[3]) - (betas[0] + betas[1]))
    return gamma, alpha, float(dbeta)


_extract_gamma_alpha_dbeta = extract_gamma_alpha_dbeta  # the reference's private name


class _YamanRHS:
    """Operator handle for the 4-wave RHS; see the module docstring."""
    native_kind = "yaman4"
    n_waves = 4
    __name__ = "rhs_yaman_simplified"

    def __call__(self, z, a_arr, params) -> np.ndarray:
        a = np.asarray(a_arr)
        if a.shape != (4,):
            raise ValueError("a_arr must have shape (4,)")
        gamma, alpha, dbeta = extract_gamma_alpha_dbeta(params)
        return _native.yaman_rhs_host(float(z), a.astype(np.complex128, copy=False), gamma, alpha, dbeta)[0]

    @staticmethod
    def batch(z, a, gamma, alpha, dbeta) -> np.ndarray:
        """N evaluations at once: z, gamma, alpha, dbeta scalars or (N,); a (N, 4) complex."""
        return _native.yaman_rhs_host(z, a, gamma, alpha, dbeta)

    def __repr__(self) -> str:
        return "<psa_amd native RHS 'yaman4'>"


rhs_yaman_simplified = _YamanRHS()


def yaman_terms(z, a, gamma, alpha, dbeta):
    """(linear, kerr, fwm) of yaman_model.py:123-132 / :135-156 / :159-186 for a batch, from the GPU kernel."""
    _, lin, kerr, fwm = _native.yaman_rhs_host(z, a, gamma, alpha, dbeta, terms=True)
    return lin, kerr, fwm


def _linear_loss_terms(a_arr, alpha):
    return yaman_terms(0.0, a_arr, 0.0, alpha, 0.0)[0][0]


def _kerr_terms(a_arr, gamma):
    return yaman_terms(0.0, a_arr, gamma, 0.0, 0.0)[1][0]


def _fwm_terms(z, a_arr, gamma, dbeta):
    return yaman_terms(z, a_arr, gamma, 0.0, dbeta)[2][0]
