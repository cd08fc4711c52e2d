"""Sweeps across the GPUs of one node: shard the points, one gather at the end.

Sweep points are independent initial-value problems (the reference's loop body has no cross-iteration state,
scan_mismtach.py:694-738), so the path shards with NO data-path collective: rank r integrates the contiguous
block ``shard_bounds(N, world, r)`` on its own GPU.  The only exchange is one ``all_gather`` of the per-point
output record (RCCL over xGMI when the process group is ``nccl``; ``gloo`` on CPU tensors in the tests):

    record[r] = [Re/Im A_end (2*n_waves rows) | p_sig_end | p_sig_max | first_bad_step (int64 bits)]   x n_local

i.e. 88 B per point for 4 waves -- a few MB per rank, latency-bound on 7 x ~153 GB/s links, so one flat
all_gather is the right collective (no bucketing, no ring tuning).

One process per GPU (``torchrun`` / ``python -m torch.distributed.run``).  ``import torch`` happens before
``libpsa_hip.so`` is loaded so both share one HIP runtime in the process.
"""
from __future__ import annotations

from typing import Callable, Optional, Tuple

import numpy as np
import torch  # noqa: F401  (must precede the native library: one HIP runtime per process)
import torch.distributed as dist

from . import _native
from .sweep import SweepResult

__all__ = ["shard_bounds", "record_rows", "pack_record", "unpack_records", "sweep_sharded", "DeviceSweep"]


def shard_bounds(n_points: int, world: int, rank: int) -> Tuple[int, int]:
    """Contiguous block split; the first ``n_points % world`` ranks get one extra point."""
    if world < 1 or not 0 <= rank < world:
        raise ValueError(f"bad rank/world: {rank}/{world}")
    base, rem = divmod(int(n_points), world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def record_rows(n_waves: int) -> int:
    return 2 * n_waves + 3


def pack_record(a_end: np.ndarray, p_end: np.ndarray, p_max: np.ndarray, first_bad: np.ndarray, width: int) -> np.ndarray:
    """(rows, width) float64, zero-padded on the right; first_bad travels as raw int64 bits in the last row."""
    n, nw = a_end.shape
    rec = np.zeros((record_rows(nw), width), dtype=np.float64)
    rec[0:2 * nw:2, :n] = a_end.real.T
    rec[1:2 * nw:2, :n] = a_end.imag.T
    rec[2 * nw, :n] = p_end
    rec[2 * nw + 1, :n] = p_max
    rec[2 * nw + 2, :n] = np.ascontiguousarray(first_bad, dtype=np.int64).view(np.float64)
    return rec


def unpack_records(gathered: np.ndarray, n_points: int, world: int, n_waves: int):
    """gathered (world, rows, width) -> (a_end (N, nw) c128, p_end, p_max, first_bad int64), shards trimmed."""
    a_parts, pe, pm, fb = [], [], [], []
    for r in range(world):
        lo, hi = shard_bounds(n_points, world, r)
        n = hi - lo
        rec = gathered[r]
        a_parts.append((rec[0:2 * n_waves:2, :n] + 1j * rec[1:2 * n_waves:2, :n]).T)
        pe.append(rec[2 * n_waves, :n])
        pm.append(rec[2 * n_waves + 1, :n])
        fb.append(np.ascontiguousarray(rec[2 * n_waves + 2, :n]).view(np.int64))
    return (np.concatenate(a_parts), np.concatenate(pe), np.concatenate(pm), np.concatenate(fb))


def _native_executor(dbeta, **kw):
    return _native.sweep_host(dbeta, **kw)


def sweep_sharded(dbeta, *, n_steps: int, z_max: float, save_every: int = 10, gamma, alpha, a0, dbeta2=None,
                  check_nan: bool = True, group=None, device: Optional[int] = None,
                  executor: Optional[Callable[..., dict]] = None) -> SweepResult:
    """Every rank passes the SAME full-sweep arguments and receives the full result.

    Per-point arrays (dbeta, optionally gamma / alpha / a0 / dbeta2 with leading dimension N) are sliced to the
    rank's block; scalars and a single a0 are broadcast.  ``executor(dbeta_local, **kw) -> dict`` runs the local
    shard; the default is the HIP kernel on ``device`` (default: LOCAL_RANK-th GPU).  Tests inject the CPU oracle
    here to exercise the shard/gather logic under ``gloo`` without a GPU.
    """
    if not dist.is_initialized():
        raise RuntimeError("torch.distributed is not initialised (launch with torchrun, one process per GPU)")
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    dbeta = np.ascontiguousarray(np.asarray(dbeta, dtype=np.float64))
    N = dbeta.shape[0]
    lo, hi = shard_bounds(N, world, rank)

    def cut(x, per_point_ndim):
        x = np.asarray(x)
        return x[lo:hi] if (x.ndim == per_point_ndim and x.shape[0] == N and N > 1) else x

    a0 = np.asarray(a0)
    nw = int(a0.shape[-1])
    kw = dict(n_steps=int(n_steps), z_max=float(z_max), save_every=int(save_every), gamma=cut(gamma, 1),
              alpha=cut(alpha, 1), a0=cut(a0, 2), check_nan=bool(check_nan))
    if dbeta2 is not None:
        kw["dbeta2"] = cut(dbeta2, 1)
    if executor is None:
        executor = _native_executor
        if device is None:
            import os
            device = int(os.environ.get("LOCAL_RANK", rank))
        kw["device"] = int(device) % max(1, _native.device_count())
    local = executor(dbeta[lo:hi], **kw) if hi > lo else dict(
        a_end=np.zeros((0, nw), complex), p_end=np.zeros(0), p_max=np.zeros(0), first_bad_step=np.zeros(0, np.int64))

    width = (N + world - 1) // world                     # widest shard; shorter ones are zero-padded
    rec = pack_record(local["a_end"], local["p_end"], local["p_max"], local["first_bad_step"], width)
    backend = dist.get_backend(group)
    dev = torch.device("cuda", torch.cuda.current_device()) if backend == "nccl" else torch.device("cpu")
    t_local = torch.from_numpy(rec).to(dev)
    # shipped as int64 words: a pure bit copy (first_bad_step = -1 is a NaN pattern when read as float64)
    t_all = torch.empty(world * t_local.numel(), dtype=torch.int64, device=dev)   # flat: rank-major concatenation
    dist.all_gather_into_tensor(t_all, t_local.reshape(-1).view(torch.int64), group=group)   # the single collective
    gathered = t_all.view(torch.float64).cpu().numpy().reshape((world,) + tuple(t_local.shape))
    a_end, p_end, p_max, first_bad = unpack_records(gathered, N, world, nw)
    return SweepResult(a_end, p_end, p_max, first_bad, int(n_steps), int(save_every),
                       float(local.get("elapsed_ms", 0.0)))


class DeviceSweep:
    """A rank's shard kept resident in HBM (torch tensors), launched on torch's current stream.

    Used by ``bench.py`` and by callers that chain sweeps without host round trips.  Layout is the SoA device
    layout of ``psa_rk4_sweep_f64_dev``; ``record`` is the (2*nw + 3, n_local) float64 tensor that
    ``all_gather_into_tensor`` ships (first_bad_step in the last row as int64 bits).
    """

    def __init__(self, dbeta_local: np.ndarray, *, n_steps: int, z_max: float, save_every: int, gamma: float,
                 alpha: float, a0: np.ndarray, check_nan: bool = True, exact_step: bool = False,
                 device: Optional[torch.device] = None, extra_flags: int = 0):
        if not torch.cuda.is_available():
            raise RuntimeError("DeviceSweep needs a GPU: libpsa_hip has no CPU fallback")
        self.device = device or torch.device("cuda", torch.cuda.current_device())
        a0 = np.asarray(a0, dtype=np.complex128)
        if a0.ndim != 1 or a0.shape[0] != 4:
            raise ValueError("DeviceSweep: a0 must be one (4,) complex vector (broadcast to every point)")
        self.n_waves, self.n_local = 4, int(np.asarray(dbeta_local).shape[0])
        self.n_steps, self.z_max, self.save_every = int(n_steps), float(z_max), int(save_every)
        f64 = dict(dtype=torch.float64, device=self.device)
        self.dbeta = torch.as_tensor(np.ascontiguousarray(dbeta_local, dtype=np.float64)).to(self.device)
        self.gamma = torch.tensor([float(gamma)], **f64)
        self.alpha = torch.tensor([float(alpha)], **f64)
        self.a0_soa = torch.tensor(np.stack([a0.real, a0.imag], 1).reshape(-1, 1), **f64).contiguous()   # [8][1]
        self.record = torch.zeros((record_rows(4), self.n_local), **f64)
        self.traj = None    # optional [n_saved][4][n_local][2] trajectory buffer (enable_trajectory)
        self.flags = (_native.BCAST_GAMMA | _native.BCAST_ALPHA | _native.BCAST_A0 | int(extra_flags)
                      | (_native.OPT_CHECK_NAN if check_nan else 0) | (_native.OPT_EXACT_STEP if exact_step else 0)
                      | (_native.OPT_LOSSLESS if float(alpha) == 0.0 else 0))

    def launch(self) -> None:
        """Asynchronous: enqueue the sweep kernel on torch's current stream."""
        r, n = self.record, self.n_local
        es = r.element_size()
        base = r.data_ptr()
        _native.sweep_device(stream=torch.cuda.current_stream(self.device).cuda_stream, n_waves=4, n_points=n,
                             n_steps=self.n_steps, z_max=self.z_max, save_every=self.save_every,
                             d_dbeta=self.dbeta.data_ptr(), d_dbeta2=0, d_gamma=self.gamma.data_ptr(),
                             d_alpha=self.alpha.data_ptr(), d_a0_soa=self.a0_soa.data_ptr(), flags=self.flags,
                             d_a_end_soa=base, d_p_end=base + 8 * n * es, d_p_max=base + 9 * n * es,
                             d_first_bad=base + 10 * n * es,
                             d_traj_soa=(self.traj.data_ptr() if self.traj is not None else 0))

    def summarize(self, p0_sig: float, *, mode: str = "max", gain_db: bool = True) -> None:
        """Enqueue the gain reduction of the sweep drivers (scan_mismtach.py:376-389 + argmax) on the same stream:
        fills ``self.gain`` (n_local,), ``self.best`` = [best_index, n_finite] (int64) and ``self.best_gain`` (1,)."""
        if not hasattr(self, "gain"):
            f64 = dict(dtype=torch.float64, device=self.device)
            self.gain = torch.empty(self.n_local, **f64)
            self.best = torch.zeros(2, dtype=torch.int64, device=self.device)
            self.best_gain = torch.zeros(1, **f64)
            self._ws = torch.empty(_native.gain_summary_workspace_bytes(self.n_local), dtype=torch.uint8, device=self.device)
        es, n, base = self.record.element_size(), self.n_local, self.record.data_ptr()
        row = 9 if mode == "max" else 8                      # p_sig_max | p_sig_end row of the record
        _native.gain_summary_device(stream=torch.cuda.current_stream(self.device).cuda_stream, n_points=n,
                                    d_p_metric=base + row * n * es, d_first_bad=base + 10 * n * es, p0_sig=p0_sig,
                                    gain_db=gain_db, d_gain=self.gain.data_ptr(), d_best_index=self.best.data_ptr(),
                                    d_best_gain=self.best_gain.data_ptr(), d_n_finite=self.best.data_ptr() + 8,
                                    d_workspace=self._ws.data_ptr())

    def enable_trajectory(self) -> int:
        """Allocate the trajectory buffer [n_saved][n_waves][n_local][2] ((re, im) pairs) in HBM; returns its bytes."""
        n_saved = self.n_steps // self.save_every + 1
        self.traj = torch.empty((n_saved, self.n_waves, self.n_local, 2), dtype=torch.float64, device=self.device)
        return self.traj.numel() * self.traj.element_size()

    def gather(self, group=None) -> torch.Tensor:
        """One all_gather of the record over the process group -> (world, rows, n_local) on this GPU."""
        world = dist.get_world_size(group)
        out = torch.empty(world * self.record.numel(), dtype=torch.int64, device=self.device)
        dist.all_gather_into_tensor(out, self.record.reshape(-1).view(torch.int64), group=group)   # bit copy
        return out.view(torch.float64).reshape((world,) + tuple(self.record.shape))

    def result(self) -> SweepResult:
        rec = self.record.cpu().numpy()
        a, pe, pm, fb = unpack_records(rec[None], self.n_local, 1, 4)
        return SweepResult(a, pe, pm, fb, self.n_steps, self.save_every, 0.0)
