"""psa-simulation-ode-rk-mvp-dispersion_amd -- MI355X-native RK4 sweep for the Agrawal-Yaman 4-wave ODE.

The directory name (mandated by the build contract) is not a valid Python identifier, so the package is
imported under the alias ``psa_amd`` (see ``psa_amd.py`` at the repository root)::

    import psa_amd
    from psa_amd.simulation import run_single_simulation
    from psa_amd.scan_mismtach import plot_max_gain_and_dbeta_vs_lambda_signal

Module names mirror the reference (config, simulation, integrators, yaman_model, scan_mismtach,
frequency_plan, phase_matching, dispersion, parameters, constants).  All numerics run in
``libpsa_hip.so`` (hand-written HIP for gfx950) through ``_native``; there is no CPU fallback.
"""
__version__ = "0.3.0"
