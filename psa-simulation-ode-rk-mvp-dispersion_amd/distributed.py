"""Sweeps across the GPUs of one node: shard the points, one gather at the end.

Sweep points are independent initial-value problems (the reference's loop body has no cross-iteration state,
scan_mismtach.py:694-738), so the path shards with NO data-path collective: rank r integrates the contiguous
block ``shard_bounds(N, world, r)`` on its own GPU.  The only exchange is one ``all_gather`` of the per-point
output record (RCCL over xGMI when the process group is ``nccl``; ``gloo`` on CPU tensors in the tests).

The record is the flat byte image the sweep kernel itself writes (``RecordLayout``), for n points of a sweep with
n_waves waves in float64 or float32 (element size es):

    [ A_end SoA: 2*n_waves rows x n x es | p_sig_end n x es | p_sig_max n x es | first_bad_step n x int64 ]

i.e. 88 B per point for 4 waves in float64, 48 B in float32, 120 B for 6 waves -- a few MB per rank, latency-bound on
7 x ~153 GB/s links, so one flat all_gather is the right collective (no bucketing, no ring tuning).  It travels as
int64 words (a pure bit copy: first_bad_step = -1 would be a NaN pattern as a float).  Ranks whose block is one point
shorter (N % world != 0) pad their image with zero words to the widest block, so every rank contributes equally.

Inputs need not travel at all: ``DeviceSweep.fill_dbeta_grid`` / ``fill_dbeta_pairs`` make each rank generate the dbeta
of its own block on its own GPU from the grid definition (two short axes), see csrc/psa_dbeta.hip.

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

__all__ = ["shard_bounds", "RecordLayout", "unpack_gathered", "all_gather_host_words", "local_device", "sweep_sharded",
           "DeviceSweep"]


def shard_bounds(n_points: int, world: int, rank: int) -> Tuple[int, int]:
    """Contiguous block split; the first ``n_points % world`` ranks get one extra point."""
    if world < 1 or not 0 <= rank < world:
        raise ValueError(f"bad rank/world: {rank}/{world}")
    base, rem = divmod(int(n_points), world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


class RecordLayout:
    """Byte layout of one shard's output record for (n_waves, dtype); see the module docstring."""

    def __init__(self, n_waves: int, dtype=np.float64):
        if n_waves not in (4, 6):
            raise ValueError("n_waves must be 4 or 6")
        self.n_waves = int(n_waves)
        self.dtype = np.dtype(dtype)
        if self.dtype not in (np.dtype(np.float64), np.dtype(np.float32)):
            raise ValueError("dtype must be float64 or float32")
        self.es = self.dtype.itemsize
        self.cdtype = np.complex128 if self.es == 8 else np.complex64
        self.float_rows = 2 * self.n_waves + 2

    def bytes_per_point(self) -> int:
        return self.float_rows * self.es + 8

    def nbytes(self, n: int) -> int:            # always a whole number of int64 words: (2*nw + 2) is even
        return int(n) * self.bytes_per_point()

    def words(self, n: int) -> int:
        return self.nbytes(n) // 8

    def offsets(self, n: int) -> dict:
        """Byte offsets of the parts inside a record of n points."""
        n, nw, es = int(n), self.n_waves, self.es
        return dict(a_end=0, p_end=2 * nw * n * es, p_max=(2 * nw + 1) * n * es, first_bad=(2 * nw + 2) * n * es)

    def pack(self, a_end: np.ndarray, p_end: np.ndarray, p_max: np.ndarray, first_bad: np.ndarray,
             pad_to: Optional[int] = None) -> np.ndarray:
        """Host arrays -> the int64-word image (zero-padded to the size of a ``pad_to``-point record)."""
        a_end = np.asarray(a_end)
        n = a_end.shape[0]
        if a_end.shape != (n, self.n_waves):
            raise ValueError(f"a_end must be (n, {self.n_waves}), got {a_end.shape}")
        buf = np.zeros(self.nbytes(max(n, pad_to or 0)), dtype=np.uint8)
        off = self.offsets(n)
        soa = np.empty((2 * self.n_waves, n), dtype=self.dtype)
        soa[0::2] = a_end.real.T
        soa[1::2] = a_end.imag.T
        buf[0:off["p_end"]] = soa.reshape(-1).view(np.uint8)
        buf[off["p_end"]:off["p_max"]] = np.ascontiguousarray(p_end, dtype=self.dtype).view(np.uint8)
        buf[off["p_max"]:off["first_bad"]] = np.ascontiguousarray(p_max, dtype=self.dtype).view(np.uint8)
        buf[off["first_bad"]:self.nbytes(n)] = np.ascontiguousarray(first_bad, dtype=np.int64).view(np.uint8)
        return buf.view(np.int64)

    def unpack(self, words: np.ndarray, n: int):
        """The int64-word image of an n-point record (trailing padding ignored) -> (a_end (n, nw) complex, p_end, p_max,
        first_bad int64)."""
        n = int(n)
        buf = np.ascontiguousarray(words).reshape(-1).view(np.uint8)
        off = self.offsets(n)
        soa = buf[0:off["p_end"]].view(self.dtype).reshape(2 * self.n_waves, n)
        a_end = np.empty((n, self.n_waves), dtype=self.cdtype)
        a_end.real = soa[0::2].T
        a_end.imag = soa[1::2].T
        p_end = buf[off["p_end"]:off["p_max"]].view(self.dtype).copy()
        p_max = buf[off["p_max"]:off["first_bad"]].view(self.dtype).copy()
        first_bad = buf[off["first_bad"]:self.nbytes(n)].view(np.int64).copy()
        return a_end, p_end, p_max, first_bad


def unpack_gathered(layout: RecordLayout, gathered: np.ndarray, n_points: int, world: int):
    """gathered (world, words(width)) int64 -> the whole sweep's (a_end, p_end, p_max, first_bad), shards trimmed."""
    parts = [layout.unpack(gathered[r], shard_bounds(n_points, world, r)[1] - shard_bounds(n_points, world, r)[0])
             for r in range(world)]
    return tuple(np.concatenate([p[k] for p in parts]) for k in range(4))


def _native_executor(dbeta, **kw):
    return _native.sweep_host(dbeta, **kw)


def _all_gather_words(t_local: torch.Tensor, world: int, group, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """The single collective of the path: every rank contributes the same number of int64 words."""
    if out is None:
        out = torch.empty(world * t_local.numel(), dtype=torch.int64, device=t_local.device)
    dist.all_gather_into_tensor(out.view(-1), t_local, group=group)
    return out.view(world, t_local.numel())


def all_gather_host_words(words: np.ndarray, group=None, device: Optional[int] = None) -> np.ndarray:
    """Host int64 words in, (world, n_words) host words out: the one collective of a sharded sweep for callers whose
    blocks live in host buffers (``sweep_sharded``, the ``scan_mismtach`` drivers).  Under ``nccl`` (RCCL) the words
    ride through the GPU that computed the block (``device``; default: the current one); under ``gloo`` they stay on
    the host.  Every rank must contribute the same number of words."""
    world = dist.get_world_size(group)
    t = torch.from_numpy(np.ascontiguousarray(words, dtype=np.int64))
    if dist.get_backend(group) == "nccl":
        dev = torch.device("cuda", int(device) if device is not None else torch.cuda.current_device())
        with torch.cuda.device(dev):
            return _all_gather_words(t.to(dev), world, group).cpu().numpy()
    return _all_gather_words(t, world, group).numpy()


def local_device(group=None) -> int:
    """The GPU a rank of a one-process-per-GPU job drives: LOCAL_RANK (torchrun), else rank modulo the visible GPUs."""
    import os
    n = max(1, _native.device_count())
    return int(os.environ.get("LOCAL_RANK", dist.get_rank(group))) % n


def sweep_sharded(dbeta, *, n_steps: int, z_max: float, save_every: int = 10, gamma, alpha, a0, dbeta2=None,
                  check_nan: bool = True, dtype=np.float64, group=None, device: Optional[int] = None,
                  executor: Optional[Callable[..., dict]] = None) -> SweepResult:
    """Every rank passes the SAME full-sweep arguments and receives the full result.

    Per-point arrays (dbeta, optionally gamma / alpha / a0 / dbeta2 with leading dimension N) are sliced to the
    rank's block; scalars and a single a0 are broadcast.  ``dtype`` float64 | float32 selects the kernel and the record
    (BASELINE config 4 is float32); six-column ``a0`` + ``dbeta2`` select the 6-wave model (config 5).
    ``executor(dbeta_local, **kw) -> dict`` runs the local shard; the default is the HIP kernel on ``device`` (default:
    LOCAL_RANK-th GPU).  Tests inject the CPU oracle here to exercise the shard/gather logic under ``gloo`` without a GPU.
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
    layout = RecordLayout(nw, dtype)
    kw = dict(n_steps=int(n_steps), z_max=float(z_max), save_every=int(save_every), gamma=cut(gamma, 1),
              alpha=cut(alpha, 1), a0=cut(a0, 2), check_nan=bool(check_nan))
    if dbeta2 is not None:
        kw["dbeta2"] = cut(dbeta2, 1)
    if layout.es == 4:
        kw["dtype"] = np.float32
    if executor is None:
        executor = _native_executor
        if device is None:
            import os
            device = int(os.environ.get("LOCAL_RANK", rank))
        kw["device"] = int(device) % max(1, _native.device_count())
    local = executor(dbeta[lo:hi], **kw) if hi > lo else dict(
        a_end=np.zeros((0, nw), layout.cdtype), p_end=np.zeros(0, layout.dtype), p_max=np.zeros(0, layout.dtype),
        first_bad_step=np.zeros(0, np.int64))

    width = (N + world - 1) // world                     # widest shard; shorter ones are zero-padded
    words = layout.pack(local["a_end"], local["p_end"], local["p_max"], local["first_bad_step"], pad_to=width)
    # the collective runs on the GPU that computed the shard, also for a rank whose block is empty
    gathered = all_gather_host_words(words, group, device=kw.get("device", device))
    a_end, p_end, p_max, first_bad = unpack_gathered(layout, gathered, N, world)
    return SweepResult(a_end, p_end, p_max, first_bad, int(n_steps), int(save_every),
                       float(local.get("elapsed_ms", 0.0)))


_TORCH_DTYPE = {np.dtype(np.float64): torch.float64, np.dtype(np.float32): torch.float32}


class DeviceSweep:
    """A rank's shard kept resident in HBM (torch tensors), launched on torch's current stream.

    Used by ``bench.py`` and by callers that chain sweeps without host round trips: float64 or float32, 4 or 6 waves
    (``a0`` with 6 entries + ``dbeta2_local``); gamma / alpha a scalar or one value per point, a0 one vector or one per point.  Layout is the SoA device layout of ``psa_rk4_sweep_f64_dev`` / ``_f32_dev``;
    ``record`` is the int64-word image of ``RecordLayout`` that the kernel writes into directly and that
    ``gather()`` ships -- no packing pass.  ``pad_to`` (the widest block of the sharded sweep) makes records of ragged
    shards equally long.  dbeta comes either from the host (``dbeta_local``) or is generated on this GPU
    (``n_local`` + ``fill_dbeta_grid`` / ``fill_dbeta_pairs``).
    """

    def __init__(self, dbeta_local: Optional[np.ndarray] = None, *, n_steps: int, z_max: float, save_every: int,
                 gamma, alpha, a0: np.ndarray, dbeta2_local: Optional[np.ndarray] = None,
                 n_local: Optional[int] = None, dtype=np.float64, check_nan: bool = True, exact_step: Optional[bool] = None,
                 device: Optional[torch.device] = None, extra_flags: int = 0, pad_to: Optional[int] = None):
        if not torch.cuda.is_available():
            raise RuntimeError("DeviceSweep needs a GPU: libpsa_hip has no CPU fallback")
        self.device = device or torch.device("cuda", torch.cuda.current_device())
        a0 = np.asarray(a0, dtype=np.complex128)
        if a0.ndim not in (1, 2) or a0.shape[-1] not in (4, 6):
            raise ValueError("DeviceSweep: a0 must be (n_waves,) (broadcast to every point) or (n_local, n_waves), n_waves 4 | 6")
        self.n_waves = int(a0.shape[-1])
        self.layout = RecordLayout(self.n_waves, dtype)
        self.np_dtype = self.layout.dtype
        self.tdtype = _TORCH_DTYPE[self.np_dtype]
        if dbeta_local is None and n_local is None:
            raise ValueError("give dbeta_local, or n_local and fill the mismatch on the device")
        self.n_local = int(np.asarray(dbeta_local).shape[0]) if dbeta_local is not None else int(n_local)
        self.n_steps, self.z_max, self.save_every = int(n_steps), float(z_max), int(save_every)
        opts = dict(dtype=self.tdtype, device=self.device)

        def to_dev(x):
            return torch.as_tensor(np.ascontiguousarray(x, dtype=self.np_dtype)).to(self.device)

        self.dbeta = to_dev(dbeta_local) if dbeta_local is not None else torch.zeros(self.n_local, **opts)
        self.dbeta2 = None
        if self.n_waves == 6:
            if dbeta_local is not None and dbeta2_local is None:
                raise ValueError("six waves need dbeta2_local")
            self.dbeta2 = to_dev(dbeta2_local) if dbeta2_local is not None else torch.zeros(self.n_local, **opts)
            if self.dbeta2.shape != self.dbeta.shape:
                raise ValueError("dbeta2_local must match dbeta_local")
        elif dbeta2_local is not None:
            raise ValueError("dbeta2_local is only meaningful for six waves")
        # gamma / alpha: a scalar (broadcast) or one value per point of the block; a0: one vector or one per point
        bcast = 0

        def per_point_or_scalar(x, name, flag):
            nonlocal bcast
            arr = np.atleast_1d(np.asarray(x, dtype=np.float64))
            if arr.shape == (1,):
                bcast |= flag
            elif arr.shape != (self.n_local,):
                raise ValueError(f"{name} must be a scalar or have one entry per point of the block ({self.n_local})")
            return to_dev(arr)

        self.gamma = per_point_or_scalar(gamma, "gamma", _native.BCAST_GAMMA)
        self.alpha = per_point_or_scalar(alpha, "alpha", _native.BCAST_ALPHA)
        if a0.ndim == 1:
            bcast |= _native.BCAST_A0
            a0 = a0[None, :]
        elif a0.shape[0] != self.n_local:
            raise ValueError(f"a0 must have one row per point of the block ({self.n_local})")
        soa = np.empty((2 * self.n_waves, a0.shape[0]), dtype=np.float64)       # row 2j = Re A_j, 2j+1 = Im A_j
        soa[0::2], soa[1::2] = a0.real.T, a0.imag.T
        self.a0_soa = to_dev(soa).contiguous()
        self.pad_to = max(self.n_local, int(pad_to or 0))
        # Two records, used alternately once `stage_to_host` is in play: launch k+1 writes one while the copy engine
        # still reads launch k's from the other (`record` is always the one the next / latest launch writes).
        self._records = [torch.zeros(self.layout.words(self.pad_to), dtype=torch.int64, device=self.device)]
        self._cur = 0                      # the record the NEXT launch writes
        self._last = 0                     # the record the latest launch wrote
        self._copy_stream = None
        self._host = [None, None]          # pinned images of (record | gathered records) per buffer
        self._host_gain = [None, None]
        self._copied = [None, None]        # event: the copy out of buffer b has finished
        self._gathered = [None, None]
        self._summ = [None, None]          # (gain, [best_index, n_finite], best_gain) of the latest summarize per buffer
        self._ws = None
        self.traj = None    # optional [n_saved][n_waves][n_local][2] trajectory buffer (enable_trajectory)
        lossless = bool(bcast & _native.BCAST_ALPHA) and float(np.asarray(alpha).reshape(-1)[0]) == 0.0
        self.flags = (bcast | int(extra_flags) | (_native.OPT_CHECK_NAN if check_nan else 0)
                      | (_native.OPT_EXACT_STEP if (check_nan and (exact_step or (exact_step is None and self.np_dtype == np.float64))) else 0) | (_native.OPT_LOSSLESS if lossless else 0))
        self._axes = {}

    @property
    def record(self) -> torch.Tensor:
        """The record of the latest launch (int64 words, RecordLayout)."""
        return self._records[self._last]

    # ---- record parts as device addresses -------------------------------------------------------------------
    def _part(self, name: str, which: Optional[int] = None) -> int:
        rec = self._records[self._cur if which is None else which]
        return rec.data_ptr() + self.layout.offsets(self.n_local)[name]

    def _stream(self) -> int:
        return torch.cuda.current_stream(self.device).cuda_stream

    # ---- dbeta generated on this GPU ---------------------------------------------------------------------------
    def _axis(self, key: str, values) -> torch.Tensor:
        t = torch.as_tensor(np.ascontiguousarray(np.atleast_1d(values), dtype=np.float64)).to(self.device)
        self._axes[key] = t      # kept alive: the launch is asynchronous
        return t

    def fill_dbeta_grid(self, model: dict, lambda1_m: float, lambda2_axis, lambda3_axis, *, first: int) -> None:
        """This rank's block [first, first + n_local) of the flattened lambda_p2 x lambda_signal grid, produced by
        ``psa_dbeta_grid_*_dev`` on torch's current stream (invalid plans become NaN -> a NaN gain downstream)."""
        if self.n_waves != 4:
            raise ValueError("the wavelength grid drives the 4-wave model; six waves use fill_dbeta_pairs")
        ax2, ax3 = self._axis("l2", lambda2_axis), self._axis("l3", lambda3_axis)
        _native.dbeta_grid_device(model, lambda1_m, stream=self._stream(), d_lambda2_axis=ax2.data_ptr(), n2=ax2.numel(),
                                  d_lambda3_axis=ax3.data_ptr(), n3=ax3.numel(), first=int(first), n_points=self.n_local,
                                  d_dbeta=self.dbeta.data_ptr(), dtype=self.np_dtype)

    def fill_dbeta_pairs(self, model: dict, omega_d: float, Omega1_axis, Omega2_axis, *, first: int) -> None:
        """Six waves: (dbeta_1, dbeta_2) of this rank's block of the flattened Omega1 x Omega2 grid."""
        if self.n_waves != 6:
            raise ValueError("fill_dbeta_pairs is for the six-wave model")
        ax1, ax2 = self._axis("O1", Omega1_axis), self._axis("O2", Omega2_axis)
        _native.dbeta_pairs_device(model, omega_d, stream=self._stream(), d_Omega1_axis=ax1.data_ptr(), n1=ax1.numel(),
                                   d_Omega2_axis=ax2.data_ptr(), n2=ax2.numel(), first=int(first), n_points=self.n_local,
                                   d_dbeta1=self.dbeta.data_ptr(), d_dbeta2=self.dbeta2.data_ptr(), dtype=self.np_dtype)

    # ---- the sweep ------------------------------------------------------------------------------------------------
    def launch(self) -> None:
        """Asynchronous: enqueue the sweep kernel on torch's current stream."""
        self._last = self._cur
        _native.sweep_device(stream=self._stream(), n_waves=self.n_waves, n_points=self.n_local,
                             n_steps=self.n_steps, z_max=self.z_max, save_every=self.save_every,
                             d_dbeta=self.dbeta.data_ptr(), d_dbeta2=(self.dbeta2.data_ptr() if self.dbeta2 is not None else 0),
                             d_gamma=self.gamma.data_ptr(), d_alpha=self.alpha.data_ptr(), d_a0_soa=self.a0_soa.data_ptr(),
                             flags=self.flags, d_a_end_soa=self._part("a_end"), d_p_end=self._part("p_end"),
                             d_p_max=self._part("p_max"), d_first_bad=self._part("first_bad"),
                             d_traj_soa=(self._traj_full.data_ptr() if self.traj is not None else 0), dtype=self.np_dtype)

    def summarize(self, p0_sig: float, *, mode: str = "max", gain_db: bool = True) -> None:
        """Enqueue the gain reduction of the sweep drivers (scan_mismtach.py:376-389 + argmax) on the same stream:
        fills ``self.gain`` (n_local,), ``self.best`` = [best_index, n_finite] (int64) and ``self.best_gain`` (1,)."""
        if mode not in ("end", "max"):
            raise ValueError(f"Unknown gain_mode={mode!r}. Use 'end' or 'max'.")
        b = self._last
        if self._summ[b] is None:
            self._summ[b] = (torch.empty(self.n_local, dtype=self.tdtype, device=self.device),
                             torch.zeros(2, dtype=torch.int64, device=self.device),
                             torch.zeros(1, dtype=torch.float64, device=self.device))
        if self._ws is None:
            self._ws = torch.empty(_native.gain_summary_workspace_bytes(self.n_local), dtype=torch.uint8, device=self.device)
        self.gain, self.best, self.best_gain = self._summ[b]
        _native.gain_summary_device(stream=self._stream(), n_points=self.n_local,
                                    d_p_metric=self._part("p_max" if mode == "max" else "p_end", b),
                                    d_first_bad=self._part("first_bad", b), p0_sig=p0_sig, gain_db=gain_db,
                                    d_gain=self.gain.data_ptr(), d_best_index=self.best.data_ptr(),
                                    d_best_gain=self.best_gain.data_ptr(), d_n_finite=self.best.data_ptr() + 8,
                                    d_workspace=self._ws.data_ptr(), dtype=self.np_dtype)

    def enable_trajectory(self) -> int:
        """Allocate the trajectory buffer [n_saved][n_waves][ld][2] ((re, im) pairs) in HBM, ld = psa_traj_ld(n_local) (the
        point count, padded off a 2 MiB stride); ``self.traj`` is its [n_saved][n_waves][n_local][2] view.  Returns the
        bytes of the rows proper."""
        import os
        n_saved = self.n_steps // self.save_every + 1
        dense = os.environ.get("PSA_TRAJ_DENSE", "0") == "1"      # A/B hook (tools/ab_traj_padding.sh): ld = n_local
        ld = self.n_local if dense else _native.traj_ld(self.n_local, self.np_dtype)
        self._traj_full = torch.empty((n_saved, self.n_waves, ld, 2), dtype=self.tdtype, device=self.device)
        self.traj = self._traj_full[:, :, :self.n_local, :]
        if not dense:
            self.flags |= _native.OPT_TRAJ_LD
        return n_saved * self.n_waves * self.n_local * 2 * self._traj_full.element_size()

    def gather(self, group=None) -> torch.Tensor:
        """One all_gather of the latest record over the process group -> (world, words(pad_to)) int64 on this GPU.  Every
        rank must have been built with the same ``pad_to`` (the widest block); trim with ``unpack_gathered``.  The result
        lives in a buffer owned by this object (one per record), reused by the launch after next."""
        world = dist.get_world_size(group)
        b = self._last
        if self._gathered[b] is None or self._gathered[b].shape[0] != world:
            self._gathered[b] = torch.empty((world, self.record.numel()), dtype=torch.int64, device=self.device)
        if dist.get_backend(group) != "nccl":
            # rehearsal backend (gloo: several ranks sharing one GPU, or CPU-only hosts): the same words, staged through
            # the host because gloo has no device-side all_gather_into_tensor
            self._gathered[b].copy_(_all_gather_words(self.record.cpu(), world, group))
            return self._gathered[b]
        with torch.cuda.device(self.device):
            return _all_gather_words(self.record, world, group, out=self._gathered[b])

    def stage_to_host(self, gathered: Optional[torch.Tensor] = None, overlap: bool = True) -> None:
        """Enqueue the device-to-host copy of the latest pass's outputs -- the record (or ``gathered``, every rank's) and,
        if ``summarize`` ran, the gains and the argmax -- into pinned host memory on a SECOND stream, then switch to the
        other record so that the next launch can start at once: the copy of pass k rides under the kernel of pass k + 1.
        Stream order is kept by events both ways (copy after the producers of pass k; the launch that reuses a buffer
        after the copy that read it).  ``host_result()`` returns the image after a synchronize.  ``overlap=False`` puts the
        copy on the launch stream itself (A/B: no second queue, the copy then sits between two kernels)."""
        b = self._last
        cur = torch.cuda.current_stream(self.device)
        if not overlap:
            self._copy_stream = cur
        elif self._copy_stream is None or self._copy_stream == cur:
            self._copy_stream = torch.cuda.Stream(device=self.device)
        if len(self._records) == 1:
            self._records.append(torch.zeros_like(self._records[0]))
        src = (self._records[b] if gathered is None else gathered).view(-1)
        if self._host[b] is None or self._host[b].numel() != src.numel():
            self._host[b] = torch.empty(src.numel(), dtype=torch.int64, pin_memory=True)
        summ = self._summ[b]
        if summ is not None and self._host_gain[b] is None:
            self._host_gain[b] = tuple(torch.empty(t.shape, dtype=t.dtype, pin_memory=True) for t in summ)
        produced = torch.cuda.Event()
        produced.record(cur)
        self._copy_stream.wait_event(produced)
        with torch.cuda.stream(self._copy_stream):
            self._host[b].copy_(src, non_blocking=True)
            if summ is not None:
                for h, t in zip(self._host_gain[b], summ):
                    h.copy_(t, non_blocking=True)
            self._copied[b] = torch.cuda.Event()
            self._copied[b].record(self._copy_stream)
        self._cur = b ^ 1
        if self._copied[self._cur] is not None:      # the pass before last read this buffer: long done, but say so
            cur.wait_event(self._copied[self._cur])

    def reserve_staging(self, world: int = 1, with_summary: bool = True) -> None:
        """Allocate everything ``summarize`` / ``gather`` / ``stage_to_host`` would otherwise create on first use -- the
        second record, the copy stream, both gathered buffers (``world`` > 1), both pinned host images and the summary
        buffers -- so that no allocation (page-locking hundreds of MB takes ~0.1 s) falls into a timed region."""
        if len(self._records) == 1:
            self._records.append(torch.zeros_like(self._records[0]))
        if self._copy_stream is None:
            self._copy_stream = torch.cuda.Stream(device=self.device)
        n_words = self._records[0].numel() * (world if world > 1 else 1)
        for b in (0, 1):
            if world > 1 and (self._gathered[b] is None or self._gathered[b].shape[0] != world):
                self._gathered[b] = torch.empty((world, self._records[0].numel()), dtype=torch.int64, device=self.device)
            if self._host[b] is None or self._host[b].numel() != n_words:
                self._host[b] = torch.empty(n_words, dtype=torch.int64, pin_memory=True)
            if with_summary:
                if self._summ[b] is None:
                    self._summ[b] = (torch.empty(self.n_local, dtype=self.tdtype, device=self.device),
                                     torch.zeros(2, dtype=torch.int64, device=self.device),
                                     torch.zeros(1, dtype=torch.float64, device=self.device))
                if self._host_gain[b] is None:
                    self._host_gain[b] = tuple(torch.empty(t.shape, dtype=t.dtype, pin_memory=True) for t in self._summ[b])
        if with_summary and self._ws is None:
            self._ws = torch.empty(_native.gain_summary_workspace_bytes(self.n_local), dtype=torch.uint8, device=self.device)

    def host_result(self):
        """(words, (gain, best, best_gain) | None) of the latest staged pass as NumPy views of the pinned buffers.
        The caller synchronizes first (``torch.cuda.synchronize`` or the event in ``_copied``)."""
        b = self._last
        if self._copied[b] is None:
            raise RuntimeError("nothing staged: call stage_to_host() after launch()")
        self._copied[b].synchronize()
        g = self._host_gain[b]
        return self._host[b].numpy(), (None if g is None else tuple(t.numpy() for t in g))

    def result(self) -> SweepResult:
        a, pe, pm, fb = self.layout.unpack(self.record.cpu().numpy(), self.n_local)
        return SweepResult(a, pe, pm, fb, self.n_steps, self.save_every, 0.0)
