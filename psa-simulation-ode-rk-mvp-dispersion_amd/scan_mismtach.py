"""Sweep drivers with the reference's call surface (the module keeps the upstream spelling ``scan_mismtach``).

* ``plot_max_signal_gain_vs_lambda_signal(...) -> (x, gain_max)``            reference scan_mismtach.py:262-430
* ``plot_max_gain_and_dbeta_vs_lambda_signal(...) -> (x, gain_max, dbeta)``  reference scan_mismtach.py:588-783
* ``plot_dbeta_vs_lambda_signal(...) -> (x, dbeta)``: what reference scan_mismtach.py:473-585 sets out to return (there every
  point comes back NaN: its ``_omega0_from_dispersion`` looks for a field the dataclass does not have)
* ``scan_dbeta_seeded_signal(...)``: a working direct-dbeta scan with gain_mode "end" | "max" and the
  argmax-over-sweep summary -- what the reference's dead ``scan_mismatch_seeded_signal`` (:43-259) set out to do.
* ``scan_gain_grid(...)``: the same sweep over a 2-D (pump-2 wavelength x signal wavelength) grid in one launch
  (BASELINE config 3's shape: 1024 x 1024 points); the reference has no grid builder.
* ``scan_six_wave_grid(...)``: BASELINE config 5's shape -- a grid over the detunings (Omega1, Omega2) of two
  signal/idler pairs sharing the two pumps, on the build-defined 6-wave model (no reference counterpart).

Where the reference loops over lambda3 in Python and calls ``run_single_simulation`` per point, these drivers
build every plan and every dbeta at once on the host (``plan_from_wavelengths_batch``,
``compute_phase_mismatch_batch``), launch ONE HIP sweep over all valid points and reduce the gain on the GPU.

Several GPUs (the reference's loop over points is embarrassingly parallel, scan_mismtach.py:357-392, :694-738):
* under a ``torch.distributed`` process group (one process per GPU, ``torchrun``) every driver splits its points into
  contiguous blocks, each rank produces the phase mismatch of ITS block (on its GPU with the device producer), integrates it
  and contributes the block's output record to ONE all_gather (RCCL under ``nccl``); every rank returns the full arrays;
* a plain Python caller passes ``devices=[0, 1, ...]``: one host thread per GPU over ``psa_rk4_sweep_f64(device=k)``.

Failure conventions kept from the reference: malformed arguments raise ``ValueError`` up front
(:315-349, :630-671); anything that would raise INSIDE the per-point ``try`` (an impossible plan, a bad cfg,
FloatingPointError from ``check_nan``) never raises -- the point's gain (and dbeta) is NaN (:391-392, :736-738).
Plotting is presentation only: it happens after the numbers exist and only if matplotlib is importable.
"""
from __future__ import annotations

from typing import Literal, Optional, Sequence, Tuple

import numpy as np

from .config import SimulationConfig, custom_simulation_config, n_steps_of  # noqa: F401
from .dispersion import DispersionParams, delta_beta_symmetric_array
from .frequency_plan import plan_from_wavelengths_batch
from .phase_matching import PhaseMatchingConfig, PhaseMatchingMethod, compute_phase_mismatch_batch
from .simulation import _prepare, make_initial_amplitudes
from .sweep import SweepResult, rk4_sweep

GainMode = Literal["end", "max"]


def _select_power_metric(Pz: np.ndarray, mode: GainMode) -> float:
    """P(z_max) ("end") or max_z P(z) ("max") of one power trace (scan_mismtach.py:27-40)."""
    Pz = np.asarray(Pz)
    if Pz.ndim != 1:
        raise ValueError("Pz must be a 1D array of power versus z.")
    if mode == "end":
        return float(Pz[-1])
    if mode == "max":
        return float(np.max(Pz))
    raise ValueError(f"Unknown gain_mode={mode!r}. Use 'end' or 'max'.")


# ---- argument checks shared by the two lambda3 drivers ---------------------------------------------------
def _check_sweep_inputs(lambda_signal_m, p_in, phase_in):
    lam3 = np.asarray(list(lambda_signal_m), dtype=float)
    if lam3.ndim != 1 or lam3.size == 0:
        raise ValueError("lambda_signal_m must be a non-empty 1D sequence")
    if not np.all(np.isfinite(lam3)) or np.any(lam3 <= 0.0):
        raise ValueError("lambda_signal_m must contain finite positive wavelengths (m)")
    p0 = np.asarray(list(p_in), dtype=float)
    if p0.shape != (4,):
        raise ValueError(f"p_in must have shape (4,), got {p0.shape}")
    if not np.all(np.isfinite(p0)) or np.any(p0 < 0.0):
        raise ValueError("p_in must contain finite non-negative powers")
    if p0[2] <= 0.0:
        raise ValueError("p_in[2] (signal seed power) must be > 0 to define gain")
    ph0 = None
    if phase_in is not None:
        ph0 = np.asarray(list(phase_in), dtype=float)
        if ph0.shape != (4,):
            raise ValueError(f"phase_in must have shape (4,), got {ph0.shape}")
        if not np.all(np.isfinite(ph0)):
            raise ValueError("phase_in must contain finite values")
    return lam3, p0, ph0


def _norm_choice(value, name, allowed):
    v = str(value).strip().lower()
    if v not in allowed:
        pretty = " or ".join(f"'{a}'" for a in allowed) if name != "gain_unit" else "'dB' or 'linear'"
        raise ValueError(f"{name} must be {pretty}")
    return v


def _wavelength_axis(lam3, unit):
    u = unit.strip().lower()
    if u == "nm":
        return lam3 * 1e9, r"Signal wavelength $\lambda_3$ (nm)"
    if u == "m":
        return lam3, r"Signal wavelength $\lambda_3$ (m)"
    raise ValueError("return_wavelength_unit must be 'm' or 'nm'")


# ---- how a driver call's points are divided --------------------------------------------------------------------
class _Shard:
    """This process's share [lo, hi) of a driver call over n points: everything when no process group is up."""

    def __init__(self, n: int, device: Optional[int]):
        import sys
        self.n, self.world, self.rank, self.group = int(n), 1, 0, None
        td = sys.modules.get("torch.distributed")      # a caller who initialised a process group has imported it
        if td is not None and td.is_available() and td.is_initialized() and td.get_world_size() > 1:
            self.world, self.rank = td.get_world_size(), td.get_rank()
        if self.world > 1:
            from .distributed import local_device, shard_bounds
            self.lo, self.hi = shard_bounds(self.n, self.world, self.rank)
            self.device = local_device() if device is None else int(device)
        else:
            self.lo, self.hi = 0, self.n
            self.device = 0 if device is None else int(device)
        self.width = -(-self.n // self.world)            # widest block

    @property
    def sharded(self) -> bool:
        return self.world > 1

    def block(self, x, per_point_ndim: int = 1):
        """This rank's rows of a per-point argument; scalars and single rows pass through."""
        x = np.asarray(x)
        return x[self.lo:self.hi] if (x.ndim == per_point_ndim and x.shape[0] == self.n and self.n > 1) else x


NEVER_RAN = -2    # first_bad_step of a point whose plan / dbeta was invalid: it never reached the kernel


def _run_block(shard: _Shard, ok_blk, dbeta_blk, *, dbeta2_blk=None, extras=(), n_waves=4, dtype=np.float64, devices=None,
               **run_kw):
    """Integrate the valid points of this process's block and, when the call is sharded over a process group, exchange
    the blocks: ONE all_gather of [output record | extras], every rank ends up with the whole sweep.

    ok_blk, dbeta_blk (, dbeta2_blk): the block's validity mask and per-metre mismatch; ``extras``: per-point float64
    arrays of the block that travel with the record (the caller-unit dbeta).  run_kw: rk4_sweep's arguments, per-point
    ones already cut to the block.  Returns (SweepResult over the valid points of the WHOLE sweep in order | None,
    ok[n], [extras over n])."""
    ok_blk = np.asarray(ok_blk, dtype=bool)
    idx = np.flatnonzero(ok_blk)
    res = None
    if idx.size:
        kw = dict(run_kw)
        for name, nd in (("gamma", 1), ("alpha", 1), ("a0", 2)):
            v = np.asarray(kw[name])
            if v.ndim == nd and v.shape[0] == ok_blk.size and ok_blk.size > 1:
                kw[name] = v[idx]
        res = rk4_sweep(np.asarray(dbeta_blk)[idx], dbeta2=(None if dbeta2_blk is None else np.asarray(dbeta2_blk)[idx]),
                        dtype=dtype, device=shard.device, devices=(None if shard.sharded else devices), **kw)
    if not shard.sharded:
        return res, ok_blk, [np.asarray(e) for e in extras]

    from .distributed import RecordLayout, all_gather_host_words, shard_bounds
    layout = RecordLayout(n_waves, dtype)
    nb = ok_blk.size
    a_end = np.full((nb, n_waves), np.nan, dtype=layout.cdtype)
    p_end, p_max = np.full(nb, np.nan, dtype=layout.dtype), np.full(nb, np.nan, dtype=layout.dtype)
    bad = np.full(nb, NEVER_RAN, dtype=np.int64)
    if res is not None:
        a_end[idx], p_end[idx], p_max[idx], bad[idx] = res.a_end, res.p_end, res.p_max, res.first_bad_step
    parts = [layout.pack(a_end, p_end, p_max, bad, pad_to=shard.width)]
    for e in extras:
        buf = np.zeros(shard.width, dtype=np.float64)
        buf[:nb] = e
        parts.append(buf.view(np.int64))
    gathered = all_gather_host_words(np.concatenate(parts), shard.group, device=shard.device)
    nrec = layout.words(shard.width)
    cols, ext = [[], [], [], []], [[] for _ in extras]
    for r in range(shard.world):
        lo, hi = shard_bounds(shard.n, shard.world, r)
        for c, part in zip(cols, layout.unpack(gathered[r, :nrec], hi - lo)):
            c.append(part)
        for k in range(len(extras)):
            ext[k].append(gathered[r, nrec + k * shard.width: nrec + k * shard.width + (hi - lo)].view(np.float64))
    a_end, p_end, p_max, bad = (np.concatenate(c) for c in cols)
    ok = bad != NEVER_RAN
    full = None
    if ok.any():
        full = SweepResult(a_end[ok], p_end[ok], p_max[ok], bad[ok], int(run_kw["n_steps"]), int(run_kw["save_every"]),
                           0.0 if res is None else res.elapsed_ms)
    return full, ok, [np.concatenate(e) for e in ext]


# ---- the engine call shared by the drivers ------------------------------------------------------------------
def _grid_dbeta(lam1, lam2_axis, lam3_axis, disp, pm_cfg, producer, device, lo=0, hi=None):
    """dbeta and validity of points [lo, hi) of the flattened lambda_p2 x lambda_signal grid (row-major; default: all).
    producer "host": the NumPy array restatement (frequency_plan / phase_matching ``*_batch``);
    producer "device": the same operations on the GPU (psa_dbeta_grid_f64, csrc/psa_dbeta.hip) -- a rank of a sharded sweep
    produces exactly its own block, so no per-point input ever travels."""
    ax2, ax3 = np.atleast_1d(lam2_axis), np.atleast_1d(lam3_axis)
    hi = ax2.size * ax3.size if hi is None else hi
    if producer == "device":
        from . import _native
        if hi == lo:
            return np.zeros(0), np.zeros(0, dtype=bool)
        return _native.dbeta_grid_host(_native.dbeta_model(disp, pm_cfg), float(lam1), ax2, ax3, first=lo, n_points=hi - lo,
                                       device=device)
    if producer != "host":
        raise ValueError("dbeta_producer must be 'host' or 'device'")
    i = np.arange(lo, hi)
    omega, ok = plan_from_wavelengths_batch(float(lam1), ax2[i // ax3.size], ax3[i % ax3.size])
    db, ok_db = compute_phase_mismatch_batch(omega, disp, pm_cfg)
    ok = ok & ok_db
    return np.where(ok, db, np.nan), ok


def _pick_producer(choice, n_points, disp, pm_cfg, even_orders=None):
    """"host" | "device" | "auto": auto takes the device producer for grids of 4 096 points or more when the model is one
    it covers (the NumPy producer costs ~0.2 us per point, 230 ms on BASELINE config 3's 1024 x 1024 grid against 19 ms
    of host time with the device producer; the two agree bit for bit on the reference's vectors).  The
    choice follows the size of the WHOLE sweep, so a sharded call picks what the unsharded one would."""
    if choice in ("host", "device"):
        return choice
    if choice != "auto":
        raise ValueError("dbeta_producer must be 'auto', 'host' or 'device'")
    if n_points < 4096 or disp is None:
        return "host"
    try:
        from . import _native
        _native.dbeta_model(disp, pm_cfg, even_orders=even_orders)
        return "device"
    except ValueError:
        return "host"


def _sweep_gain(*, cfg, lam1, grid_axes, gamma, alpha, p0, ph0, dispersion, pm_cfg, length_unit, gain_unit,
                gain_mode="max", device=None, devices=None, dbeta_producer="host", caller_dbeta=None):
    """Everything the reference does inside its per-point ``try``, for all points of the lambda_p2 x lambda_signal grid
    ``grid_axes`` at once (a lambda3 sweep is its 1 x N case).

    ``caller_dbeta = (dispersion as given, pm_cfg)`` also produces the drivers' returned dbeta (1/length_unit, computed
    from the UNSCALED dispersion like scan_mismtach.py:700-706) block by block.  Returns (gain[N], dbeta_caller[N] | None,
    SweepResult | None).  Never raises for per-point or cfg problems: those become NaN, as ``except Exception`` does upstream.
    """
    if dbeta_producer not in ("auto", "host", "device"):       # a caller error, not a per-point failure
        raise ValueError("dbeta_producer must be 'auto', 'host' or 'device'")
    ax2, ax3 = np.atleast_1d(grid_axes[0]), np.atleast_1d(grid_axes[1])
    N = ax2.size * ax3.size
    shard = _Shard(N, device)
    gain = np.full(N, np.nan)
    nb = shard.hi - shard.lo
    try:
        # plan-independent part of run_single_simulation (validation, unit scaling, containers)
        pre = _prepare(cfg, gamma=gamma, alpha=alpha, dispersion=dispersion, phase_matching_cfg=pm_cfg,
                       beta_legacy=None, length_unit=length_unit)
        a0 = make_initial_amplitudes(p0, ph0)
        fiber, grid, pm = pre["fiber"], pre["grid"], pre["pm"].config
        producer = _pick_producer(dbeta_producer, N, fiber.dispersion, pm)
        dbeta_m, ok = _grid_dbeta(lam1, ax2, ax3, fiber.dispersion, pm, producer, shard.device, shard.lo, shard.hi)
        n_steps = n_steps_of(fiber.length_m, grid.dz_m)
        if n_steps < 1:
            raise ValueError("no steps")
        run_kw = dict(z_max=fiber.length_m, n_steps=n_steps, save_every=cfg.save_every, check_nan=bool(cfg.check_nan),
                      gamma=fiber.gamma_W_m, alpha=fiber.alpha_1_m, a0=a0)
    except Exception:
        # a bad cfg fails identically on every rank (same arguments), so nobody enters the collective: the returned dbeta
        # (which upstream is set before the run is attempted) is then produced for the whole grid by each rank itself
        cd = None
        if caller_dbeta is not None:
            try:
                cd, _ = _grid_dbeta(lam1, ax2, ax3, caller_dbeta[0], caller_dbeta[1],
                                    _pick_producer(dbeta_producer, N, caller_dbeta[0], caller_dbeta[1]), shard.device)
            except Exception:
                cd = np.full(N, np.nan)
        return gain, cd, None
    extras = []
    if caller_dbeta is not None:
        try:
            cd, _ = _grid_dbeta(lam1, ax2, ax3, caller_dbeta[0], caller_dbeta[1],
                                _pick_producer(dbeta_producer, N, caller_dbeta[0], caller_dbeta[1]), shard.device,
                                shard.lo, shard.hi)
        except Exception:
            cd = np.full(nb, np.nan)
        extras = [cd]
    res, ok_full, extras_full = _run_block(shard, ok, dbeta_m, extras=extras, devices=devices, **run_kw)
    if res is not None:
        gain[ok_full] = res.gain(p0[2], mode=gain_mode, unit=gain_unit, device=shard.device)
    return gain, (extras_full[0] if extras_full else None), res


def _maybe_plot(draw, save_path, show):
    """Presentation tail (scan_mismtach.py:412-428, :753-781); skipped when there is nothing to show or save."""
    if save_path is None and not show:
        return
    try:
        import matplotlib.pyplot as plt
    except Exception:  # plotting is optional here
        return
    fig = draw(plt)
    if save_path is not None:
        fig.savefig(save_path, dpi=200, bbox_inches="tight")
    if show:
        plt.show()
    else:
        plt.close(fig)


def plot_max_signal_gain_vs_lambda_signal(*, cfg: SimulationConfig, lambda_p1_m: float, lambda_p2_m: float,
                                          lambda_signal_m: Sequence[float], gamma: float, alpha: float,
                                          p_in: Sequence[float], phase_in: Optional[Sequence[float]] = None,
                                          dispersion: Optional[DispersionParams] = None,
                                          phase_matching_cfg: Optional[PhaseMatchingConfig] = None,
                                          length_unit: str = "m", return_wavelength_unit: str = "nm",
                                          gain_unit: str = "dB", xscale: str = "linear", yscale: str = "linear",
                                          show_progress: bool = True, tqdm_desc: str = "Sweeping λ3",
                                          save_path: Optional[str] = None, show: bool = True,
                                          devices: Optional[Sequence[int]] = None) -> Tuple[np.ndarray, np.ndarray]:
    """Max-over-z signal gain versus lambda3 -> (x_wavelength, gain_max); NaN where a point failed.

    ``show_progress`` / ``tqdm_desc`` are accepted for compatibility: the sweep is a single kernel launch.
    ``devices`` (not upstream): GPUs of this process to split the points over; under a process group the ranks split them.
    """
    lam1, lam2 = float(lambda_p1_m), float(lambda_p2_m)
    lam3, p0, ph0 = _check_sweep_inputs(lambda_signal_m, p_in, phase_in)
    unit = _norm_choice(gain_unit, "gain_unit", ("db", "linear"))
    xs = _norm_choice(xscale, "xscale", ("linear", "log"))
    ys = _norm_choice(yscale, "yscale", ("linear", "log"))
    if ys == "log" and unit == "db":
        raise ValueError("yscale='log' is not supported with gain_unit='dB'. Use gain_unit='linear'.")
    _wavelength_axis(lam3, return_wavelength_unit)   # the reference raises this only AFTER its sweep; here before any device work

    # a lambda3 sweep is the 1 x N case of the grid: sweeps of 4 096 points or more get their dbeta from the device producer
    gain, _, _ = _sweep_gain(cfg=cfg, lam1=lam1, grid_axes=(np.array([lam2]), lam3), gamma=gamma, alpha=alpha, p0=p0, ph0=ph0,
                             dispersion=dispersion, pm_cfg=phase_matching_cfg, length_unit=length_unit,
                             gain_unit=unit, dbeta_producer="auto", devices=devices)
    x, x_label = _wavelength_axis(lam3, return_wavelength_unit)

    def draw(plt):
        fig = plt.figure()
        plt.plot(x, gain, marker="o")
        plt.xlabel(x_label)
        plt.ylabel(r"Max signal gain $G_{\max}$ (linear)" if unit == "linear" else r"Max signal gain $G_{\max}$ (dB)")
        plt.title("Maximum signal gain vs signal wavelength")
        plt.grid(True, which="both")
        plt.xscale(xs)
        plt.yscale(ys)
        return fig

    _maybe_plot(draw, save_path, show)
    return x, gain


def plot_max_gain_and_dbeta_vs_lambda_signal(*, cfg: SimulationConfig, lambda_p1_m: float, lambda_p2_m: float,
                                             lambda_signal_m: Sequence[float], gamma: float, alpha: float,
                                             p_in: Sequence[float], phase_in: Optional[Sequence[float]] = None,
                                             dispersion: DispersionParams,
                                             phase_matching_cfg: Optional[PhaseMatchingConfig] = None,
                                             length_unit: str = "m", return_wavelength_unit: str = "nm",
                                             gain_unit: str = "dB", xscale: str = "linear",
                                             yscale_gain: str = "linear", yscale_dbeta: str = "linear",
                                             show_progress: bool = True,
                                             tqdm_desc: str = "Sweeping λ3 (gain + dBeta)",
                                             save_path: Optional[str] = None, show: bool = True,
                                             devices: Optional[Sequence[int]] = None
                                             ) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
    """One sweep that returns both the max signal gain and dbeta(lambda3) -> (x, gain_max, dbeta).

    ``dbeta`` is in 1/length_unit, computed from the dispersion AS GIVEN (scan_mismtach.py:700-706); the kernel
    uses the per-metre value derived from the scaled dispersion (simulation.py:340), exactly as upstream.
    """
    lam1, lam2 = float(lambda_p1_m), float(lambda_p2_m)
    lam3, p0, ph0 = _check_sweep_inputs(lambda_signal_m, p_in, phase_in)
    if dispersion is None:
        raise ValueError("dispersion must be provided to compute dBeta(λ3)")
    unit = _norm_choice(gain_unit, "gain_unit", ("db", "linear"))
    xs = _norm_choice(xscale, "xscale", ("linear", "log"))
    ysg = _norm_choice(yscale_gain, "yscale_gain", ("linear", "log"))
    ysd = _norm_choice(yscale_dbeta, "yscale_dbeta", ("linear", "log"))
    if ysg == "log" and unit == "db":
        raise ValueError("yscale_gain='log' is not supported with gain_unit='dB'. Use gain_unit='linear'.")
    pm_cfg = phase_matching_cfg if phase_matching_cfg is not None else PhaseMatchingConfig(
        method=PhaseMatchingMethod.SYMMETRIC_EVEN, max_order=4, even_orders=(2, 4), atol=0.0, rtol=1e-12)
    _wavelength_axis(lam3, return_wavelength_unit)   # validated up front (upstream: after the sweep)

    # dbeta in the caller's units travels with the sweep: per point, NaN where the plan or the mismatch is invalid (a lambda3
    # sweep is the 1 x N case of the grid: 4 096 points or more go through the device producer)
    gain, dbeta, _ = _sweep_gain(cfg=cfg, lam1=lam1, grid_axes=(np.array([lam2]), lam3), gamma=gamma, alpha=alpha, p0=p0,
                                 ph0=ph0, dispersion=dispersion, pm_cfg=pm_cfg, length_unit=length_unit, gain_unit=unit,
                                 dbeta_producer="auto", caller_dbeta=(dispersion, pm_cfg), devices=devices)
    gain = np.where(np.isnan(dbeta), np.nan, gain)   # a point whose dbeta failed never reaches the run upstream
    x, x_label = _wavelength_axis(lam3, return_wavelength_unit)
    ref_line = -float(gamma) * float(p0[0] + p0[1])

    def draw(plt):
        fig, (ax1, ax2) = plt.subplots(2, 1, sharex=True, figsize=(9, 7))
        ax1.plot(x, gain, marker="o")
        ax1.set_ylabel("Max signal gain (linear)" if unit == "linear" else "Max signal gain (dB)")
        ax1.grid(True, which="both", alpha=0.3)
        ax1.set_yscale(ysg)
        ax2.plot(x, dbeta, marker="o", label=r"$\Delta\beta(\lambda_3)$")
        ax2.axhline(ref_line, ls="--", lw=2, label=r"$\gamma(P_1+P_2)$")
        ax2.set_xlabel(x_label)
        ax2.set_ylabel(rf"$\Delta\beta$  [1/{length_unit}]")
        ax2.grid(True, which="both", alpha=0.3)
        ax2.set_xscale(xs)
        ax2.set_yscale(ysd)
        ax2.legend()
        fig.suptitle("Max signal gain and phase mismatch vs signal wavelength")
        fig.tight_layout()
        return fig

    _maybe_plot(draw, save_path, show)
    return x, gain, dbeta


def plot_dbeta_vs_lambda_signal(*, gamma: float, lambda_p1_m: float, lambda_p2_m: float, lambda_signal_m: Sequence[float],
                                p_in: Sequence[float], dispersion: DispersionParams, return_wavelength_unit: str = "nm",
                                xscale: str = "linear", yscale: str = "linear", length_unit: str = "m",
                                show_progress: bool = True, tqdm_desc: str = "Scanning dBeta(λ3)",
                                title: Optional[str] = None, save_path: Optional[str] = None, show: bool = True,
                                device: Optional[int] = None, dbeta_producer: str = "auto"
                                ) -> Tuple[np.ndarray, np.ndarray]:
    """dbeta(lambda3) = beta(w1) + beta(w2) - beta(w3) - beta(w4) with beta by its Taylor series through order 4, next to the
    line gamma*(P1 + P2) -> (x, dbeta); dbeta in 1/(the length unit of the dispersion coefficients).

    The call surface, argument checks and NaN-per-failed-point rule of reference scan_mismtach.py:473-585.  Upstream the
    function returns NaN for every point (its helper asks the dispersion object for ``omega0``, the field is ``omega_ref``,
    and the exception is swallowed: SURVEY R3); the quantity its docstring and ``_beta_taylor`` (:441-459) describe is the
    reference's own ``delta_beta_from_omegas`` (dispersion.py:282-318, beta0 and beta1 cancel by energy conservation), which
    is what is returned here -- through the array producer, or the device one for 4 096 points or more.
    """
    lam1, lam2 = float(lambda_p1_m), float(lambda_p2_m)
    lam3 = np.asarray(list(lambda_signal_m), dtype=float)
    if lam3.ndim != 1 or lam3.size == 0:
        raise ValueError("lambda_signal_m must be a non-empty 1D sequence")
    if not np.all(np.isfinite(lam3)) or np.any(lam3 <= 0.0):
        raise ValueError("lambda_signal_m must contain finite positive wavelengths (m)")
    p0 = np.asarray(list(p_in), dtype=float)
    if p0.shape != (4,):
        raise ValueError(f"p_in must have shape (4,), got {p0.shape}")
    if not np.all(np.isfinite(p0)) or np.any(p0 < 0.0):
        raise ValueError("p_in must contain finite non-negative powers")
    xs = _norm_choice(xscale, "xscale", ("linear", "log"))
    ys = _norm_choice(yscale, "yscale", ("linear", "log"))
    if dispersion is None:
        raise ValueError("dispersion must be provided to compute dBeta(λ3)")

    pm_cfg = PhaseMatchingConfig(method=PhaseMatchingMethod.GENERAL_TAYLOR, max_order=4, atol=0.0, rtol=1e-12)
    producer = _pick_producer(dbeta_producer, lam3.size, dispersion, pm_cfg)
    dbeta, _ = _grid_dbeta(lam1, np.array([lam2]), lam3, dispersion, pm_cfg, producer, 0 if device is None else int(device))
    x, x_label = _wavelength_axis(lam3, return_wavelength_unit)
    y_unit = "1/km" if str(length_unit).strip().lower() == "km" else "1/m"
    ref_line = float(gamma) * float(p0[0] + p0[1])
    if ys == "log" and (not np.nanmin(dbeta) > 0.0 or ref_line <= 0.0):     # all-NaN counts as "not > 0" (upstream: a warning + NaN)
        raise ValueError("yscale='log' requires dBeta and gamma*(P1+P2) to be strictly > 0.")

    def draw(plt):
        fig = plt.figure(figsize=(8.0, 5.0))
        plt.plot(x, dbeta, label=r"$d\beta(\lambda_3)$")
        plt.axhline(ref_line, linestyle="--", label=r"$\gamma(P_1+P_2)$")
        plt.xlabel(x_label)
        plt.ylabel(rf"$d\beta$ [{y_unit}]")
        plt.xscale(xs)
        plt.yscale(ys)
        if title is not None:
            plt.title(title)
        plt.grid(True, which="both", linestyle="--", alpha=0.5)
        plt.legend()
        plt.tight_layout()
        return fig

    _maybe_plot(draw, save_path, show)
    return x, dbeta


def scan_dbeta_seeded_signal(*, cfg: SimulationConfig, delta_beta: Sequence[float], gamma, alpha,
                             p_in: Sequence[float], phase_in: Optional[Sequence[float]] = None,
                             length_unit: str = "m", gain_mode: GainMode = "end", gain_unit: str = "dB",
                             dtype=np.float64, device: Optional[int] = None,
                             devices: Optional[Sequence[int]] = None) -> dict:
    """Scan the phase mismatch directly (PROVIDED dbeta per point) and summarise the signal gain.

    delta_beta: (N,) in 1/length_unit.  gamma / alpha: scalars or (N,) in per-length_unit.
    Returns dict(delta_beta, gain, best_index, best_delta_beta, best_gain, n_finite, result=SweepResult,
    points_per_s) -- gain with the reference's NaN rules, argmax/max reduced on the GPU.
    """
    if gain_mode not in ("end", "max"):
        raise ValueError(f"Unknown gain_mode={gain_mode!r}. Use 'end' or 'max'.")
    unit = _norm_choice(gain_unit, "gain_unit", ("db", "linear"))
    db = np.asarray(delta_beta, dtype=float)
    if db.ndim != 1 or db.size == 0:
        raise ValueError("delta_beta must be a non-empty 1D sequence")
    _, p0, ph0 = _check_sweep_inputs([1.0], p_in, phase_in)
    pre = _prepare(cfg, gamma=0.0, alpha=0.0, dispersion=None,
                   phase_matching_cfg=PhaseMatchingConfig(method=PhaseMatchingMethod.PROVIDED, provided_delta_beta=0.0),
                   beta_legacy=None, length_unit=length_unit)
    scale, L, dz_m = pre["scale"], pre["fiber"].length_m, pre["grid"].dz_m
    shard = _Shard(db.size, device)
    gam, alp = np.asarray(gamma, dtype=float) / scale, np.asarray(alpha, dtype=float) / scale
    for name, v in (("gamma", gam), ("alpha", alp)):
        if v.ndim > 1 or (v.ndim == 1 and v.shape[0] not in (1, db.size)):
            raise ValueError(f"{name} must be a scalar or have one entry per delta_beta point")
    res, _, _ = _run_block(shard, np.ones(shard.hi - shard.lo, dtype=bool), shard.block(db / scale), dtype=dtype,
                           devices=devices, z_max=L, n_steps=n_steps_of(L, dz_m), save_every=cfg.save_every,
                           check_nan=bool(cfg.check_nan), gamma=shard.block(gam), alpha=shard.block(alp),
                           a0=make_initial_amplitudes(p0, ph0))
    gain, bi, bg, nf = res.summary(p0[2], mode=gain_mode, unit=unit, device=shard.device)
    secs = max(res.elapsed_ms, 1e-9) * 1e-3
    return dict(delta_beta=db, gain=gain, best_index=bi, best_delta_beta=(float(db[bi]) if bi >= 0 else float("nan")),
                best_gain=bg, n_finite=nf, result=res, points_per_s=db.size / secs)


def scan_gain_grid(*, cfg: SimulationConfig, lambda_p1_m: float, lambda_p2_m: Sequence[float],
                   lambda_signal_m: Sequence[float], gamma: float, alpha: float, p_in: Sequence[float],
                   phase_in: Optional[Sequence[float]] = None, dispersion: DispersionParams,
                   phase_matching_cfg: Optional[PhaseMatchingConfig] = None, length_unit: str = "m",
                   gain_unit: str = "dB", gain_mode: GainMode = "max", device: Optional[int] = None,
                   dbeta_producer: str = "auto", devices: Optional[Sequence[int]] = None) -> dict:
    """Signal gain over the grid lambda_p2[Ny] x lambda_signal[Nx]: Ny*Nx independent runs, one kernel launch.
    ``dbeta_producer``: "host" (NumPy), "device" (the grid's phase mismatch computed on the GPU as well: same operations, see
    _grid_dbeta) or "auto" (device for grids of 4 096 points or more).

    Row iy is what ``plot_max_gain_and_dbeta_vs_lambda_signal(lambda_p2_m=lambda_p2[iy], ...)`` returns (same plans, same
    NaN rules; the same dbeta bit for bit when both calls use the same producer -- "auto" decides by the size of the whole
    call, and the two producers differ by a few ulp on ~0.2 % of points).  ``device`` / ``devices``: one GPU,
    or several of this process; under a ``torch.distributed`` process group the ranks split the grid.  Returns dict(gain (Ny, Nx), dbeta (Ny, Nx) in 1/length_unit,
    best_index (iy, ix) | None, best_gain, n_finite, result=SweepResult | None).
    """
    if gain_mode not in ("end", "max"):
        raise ValueError(f"Unknown gain_mode={gain_mode!r}. Use 'end' or 'max'.")
    unit = _norm_choice(gain_unit, "gain_unit", ("db", "linear"))
    lam3, p0, ph0 = _check_sweep_inputs(lambda_signal_m, p_in, phase_in)
    lam2 = np.asarray(list(lambda_p2_m), dtype=float)
    if lam2.ndim != 1 or lam2.size == 0 or not np.all(np.isfinite(lam2)) or np.any(lam2 <= 0.0):
        raise ValueError("lambda_p2_m must be a non-empty 1D sequence of finite positive wavelengths (m)")
    if dispersion is None:
        raise ValueError("dispersion must be provided")
    pm_cfg = phase_matching_cfg if phase_matching_cfg is not None else PhaseMatchingConfig()
    gain, dbeta, res = _sweep_gain(cfg=cfg, lam1=float(lambda_p1_m), grid_axes=(lam2, lam3), gamma=gamma, alpha=alpha, p0=p0,
                                   ph0=ph0, dispersion=dispersion, pm_cfg=pm_cfg, length_unit=length_unit, gain_unit=unit,
                                   gain_mode=gain_mode, device=device, devices=devices, dbeta_producer=dbeta_producer,
                                   caller_dbeta=(dispersion, pm_cfg))
    gain = np.where(np.isnan(dbeta), np.nan, gain)
    finite = np.isfinite(gain)
    best = None
    if finite.any():
        flat = int(np.nanargmax(gain))
        best = (flat // lam3.size, flat % lam3.size)
    return dict(gain=gain.reshape(lam2.size, lam3.size), dbeta=dbeta.reshape(lam2.size, lam3.size), best_index=best,
                best_gain=(float(gain.reshape(-1)[best[0] * lam3.size + best[1]]) if best else float("nan")),
                n_finite=int(finite.sum()), result=res)


def scan_six_wave_grid(*, cfg: SimulationConfig, lambda_p1_m: float, lambda_p2_m: float, Omega1: Sequence[float],
                       Omega2: Sequence[float], gamma: float, alpha: float, p_in: Sequence[float],
                       phase_in: Optional[Sequence[float]] = None, dispersion: DispersionParams,
                       even_orders: Tuple[int, ...] = (2, 4), length_unit: str = "m", gain_unit: str = "dB",
                       gain_mode: GainMode = "max", device: Optional[int] = None, dbeta_producer: str = "auto",
                       devices: Optional[Sequence[int]] = None) -> dict:
    """Six waves [p1, p2, s1, i1, s2, i2]: pair k sits at omega_c +- Omega_k (omega_c, omega_d from the two pumps) and
    has dbeta_k = sum_{n even} beta_n (Omega_k^n - omega_d^n) 2/n!  (the symmetric-even form, dispersion.py:321-372).
    Runs the Omega1[Ny] x Omega2[Nx] grid in ONE launch of the 6-wave kernel.

    p_in / phase_in: six entries.  Returns dict(gain (Ny, Nx) of signal 1 -- the kernel's summary wave --, dbeta1 (Ny,),
    dbeta2 (Nx,), a_end (Ny, Nx, 6), first_bad_step (Ny, Nx), result=SweepResult).  The 6-wave model is build-defined:
    with pair 2 dark every row equals the 4-wave run at dbeta1 (tested), beyond that parity is unpinned.
    """
    if gain_mode not in ("end", "max"):
        raise ValueError(f"Unknown gain_mode={gain_mode!r}. Use 'end' or 'max'.")
    unit = _norm_choice(gain_unit, "gain_unit", ("db", "linear"))
    p0 = np.asarray(list(p_in), dtype=float)
    if p0.shape != (6,) or not np.all(np.isfinite(p0)) or np.any(p0 < 0.0):
        raise ValueError("p_in must hold six finite non-negative powers [p1, p2, s1, i1, s2, i2]")
    if p0[2] <= 0.0:
        raise ValueError("p_in[2] (signal-1 seed power) must be > 0 to define gain")
    ph = None if phase_in is None else np.asarray(list(phase_in), dtype=float)
    if ph is not None and (ph.shape != (6,) or not np.all(np.isfinite(ph))):
        raise ValueError("phase_in must hold six finite phases")
    O1, O2 = np.asarray(list(Omega1), dtype=float), np.asarray(list(Omega2), dtype=float)
    if O1.ndim != 1 or O2.ndim != 1 or O1.size == 0 or O2.size == 0 or not (np.all(np.isfinite(O1)) and np.all(np.isfinite(O2))):
        raise ValueError("Omega1 and Omega2 must be non-empty 1D sequences of finite detunings (rad/s)")
    if dispersion is None:
        raise ValueError("dispersion must be provided")
    from .frequency_plan import omega_from_lambda
    w1, w2 = omega_from_lambda(lambda_p1_m), omega_from_lambda(lambda_p2_m)
    wc, wd = 0.5 * (w1 + w2), 0.5 * (w1 - w2)
    if np.any(np.abs(O1) >= wc) or np.any(np.abs(O2) >= wc):
        raise ValueError("|Omega| must stay below omega_c (sideband frequencies must be positive)")
    pre = _prepare(cfg, gamma=gamma, alpha=alpha, dispersion=dispersion, phase_matching_cfg=None, beta_legacy=None,
                   length_unit=length_unit)
    disp_m, fiber, grid = pre["fiber"].dispersion, pre["fiber"], pre["grid"]
    db1 = delta_beta_symmetric_array(wd, O1, disp_m, even_orders=even_orders)      # per metre
    db2 = delta_beta_symmetric_array(wd, O2, disp_m, even_orders=even_orders)
    dbeta_producer = _pick_producer(dbeta_producer, O1.size * O2.size, disp_m, None, even_orders=even_orders)
    shard = _Shard(O1.size * O2.size, device)
    if shard.hi == shard.lo:
        d1_blk, d2_blk = np.zeros(0), np.zeros(0)
    elif dbeta_producer == "device":     # this block's (dbeta_1, dbeta_2) from the GPU producer (psa_dbeta_pairs_f64)
        from . import _native
        d1_blk, d2_blk = _native.dbeta_pairs_host(_native.dbeta_model(disp_m, None, even_orders=even_orders), wd, O1, O2,
                                                  first=shard.lo, n_points=shard.hi - shard.lo, device=shard.device)
    else:
        i = np.arange(shard.lo, shard.hi)
        d1_blk, d2_blk = db1[i // O2.size], db2[i % O2.size]
    from .sweep import initial_amplitudes
    res, _, _ = _run_block(shard, np.ones(shard.hi - shard.lo, dtype=bool), d1_blk, dbeta2_blk=d2_blk, n_waves=6,
                           devices=devices, z_max=fiber.length_m, n_steps=n_steps_of(fiber.length_m, grid.dz_m),
                           save_every=cfg.save_every, check_nan=bool(cfg.check_nan), gamma=fiber.gamma_W_m,
                           alpha=fiber.alpha_1_m, a0=initial_amplitudes(p0, ph))
    gain = res.gain(p0[2], mode=gain_mode, unit=unit, device=shard.device)
    shape = (O1.size, O2.size)
    return dict(gain=gain.reshape(shape), dbeta1=db1 * pre["scale"], dbeta2=db2 * pre["scale"],
                a_end=res.a_end.reshape(shape + (6,)), first_bad_step=res.first_bad_step.reshape(shape), result=res)
