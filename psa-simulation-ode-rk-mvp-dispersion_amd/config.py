"""Numerical run configuration -- same field names and factories as the reference's config.py
(SimulationConfig config.py:6-30, factories :33-70, validate_config :73-93), so existing call
sites keep working.  Lengths are in the caller's ``length_unit`` (see simulation.run_single_simulation).
"""
from __future__ import annotations

from dataclasses import dataclass

__all__ = ["SimulationConfig", "default_simulation_config", "custom_simulation_config", "validate_config",
           "n_steps_of"]


@dataclass(frozen=True)
class SimulationConfig:
    z_max: float      # propagation length
    dz: float         # nominal step; the effective step is z_max / round(z_max / dz)
    integrator: str   # only "rk4"
    save_every: int   # keep every save_every-th step (row 0 is z = 0)
    check_nan: bool   # per-step NaN/Inf detection -> FloatingPointError (single run) / NaN gain (sweeps)
    verbose: bool     # accepted for compatibility; never read (as in the reference)


def custom_simulation_config(*, z_max=1.0, dz=1e-3, integrator="rk4", save_every=10, check_nan=True,
                             verbose=False) -> SimulationConfig:
    return SimulationConfig(z_max, dz, integrator, save_every, check_nan, verbose)


def default_simulation_config() -> SimulationConfig:
    return custom_simulation_config(z_max=0.5)


def validate_config(cfg: SimulationConfig) -> None:
    """Raise ValueError for the same five conditions as the reference (config.py:80-93)."""
    problems = (
        (cfg.z_max <= 0.0, "z_max must be positive"),
        (cfg.dz <= 0.0, "dz must be positive"),
        (cfg.dz > cfg.z_max, "dz must be smaller than z_max"),
        (str(cfg.integrator).lower() != "rk4", f"Unsupported integrator: {cfg.integrator}"),
        (cfg.save_every <= 0, "save_every must be a positive integer"),
    )
    for bad, msg in problems:
        if bad:
            raise ValueError(msg)


def n_steps_of(z_max: float, dz: float) -> int:
    """``int(round(z_max / dz))`` -- the reference's step count (integrators.py:194), round-half-even."""
    return int(round(z_max / dz))
