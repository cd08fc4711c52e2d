#!/usr/bin/env python3
"""bench.py -- RK4 field-point updates/s of the HIP sweep on MI355X (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config c2|c3|c4|c5]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One bench "step" = one pass of the hot path over one batch.  The default workload (the headline, `--config c2`) is
BASELINE config 2 per GPU: 65 536 dbeta sweep points x 4 fields x 100 000 z-steps, float64, dbeta = linspace(-0.05, 0.05),
gamma = 0.0115, alpha = 1.15e-4, P = (0.5, 0.5, 1e-5, 1e-5) W, L = 1000 m, save_every = 10, check_nan on (SURVEY 8d "C2").
The other BASELINE configurations run through the same code path:
    c3  1 048 576 points per GPU (lambda_p2[1024] x lambda_3[1024] grid), 4 fields, 100 000 z-steps, float64
    c4  131 072 points per GPU (one eighth of the 1024 x 1024 grid), 4 fields, 1 000 000 z-steps, float32
    c5  32 768 points per GPU (one eighth of the 512 x 512 (Omega1, Omega2) grid), 6 fields, 100 000 z-steps, float64
For c3-c5 every rank GENERATES the dbeta of its block on its own GPU (psa_dbeta_grid_*_dev / psa_dbeta_pairs_*_dev): no
per-point input is scattered.  Inputs are resident in HBM before the timed region; a step is the sweep kernel on the
rank's shard, the on-device gain reduction (per-point gain + argmax, the sweep drivers' summary) and, for N > 1, the single
RCCL all_gather of the output record.  Weak scaling: every rank owns the same number of points (rank r gets the r-th
contiguous block of a sweep that is N times as large); at N = 8 c4 and c5 are exactly BASELINE configs 4 and 5.

Prints ONE JSON line on rank 0 (metric/value/unit/... + "roofline" + "cpu_baseline").
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402  (before the native library: one HIP runtime per process)
import torch.distributed as dist  # noqa: E402

import psa_amd._native as nat  # noqa: E402
from psa_amd.distributed import DeviceSweep, shard_bounds  # noqa: E402

Z_MAX = 1000.0
SAVE_EVERY = 10
GAMMA, ALPHA = 0.0115, 1.15e-4
P_C2 = np.array([0.5, 0.5, 1e-5, 1e-5])
P_GRID = np.array([0.1, 0.1, 1e-7, 1e-7])
P_SIX = np.array([0.3, 0.25, 1e-6, 1e-6, 2e-6, 5e-7])
DBETA_RANGE = (-0.05, 0.05)

# ---- accounting agreed in BASELINE.md section 2 / SURVEY 8(d); the 6-wave count follows the same rules --
# 4 waves: SURVEY 8(d)'s count (652; the kernel executes 524 of them in 298 instructions).  6 waves: 2 flops x the 468 FP64
# instructions of the one-lane step -- the most the vector ALU can execute for it.  The rule-by-rule count
# of round 2 (1 156) credited a shared sum six times over and let a measured run reach 1.02 of the peak.
FLOPS_PER_RK4_STEP = {4: 652, 6: 936}
NOMINAL_GHZ = 2.4                            # the clock the 78.6 / 157.3 TFLOP/s peaks are quoted at
PEAK_TFLOPS = {"f64": 78.6,                  # 256 CU x 4 SIMD x 16 lanes x 2 flop x 2.4 GHz (vector FP64)
               "f32": 157.3}                 # the same with two packed float32 per lane (v_pk_fma_f32)
PEAK_HBM_GBS = 8000.0                        # MI355X_MICROARCH.md: HBM3E 8 TB/s spec

CONFIGS = {
    "c2": dict(baseline="configs[1]", n_waves=4, dtype="f64", pts_per_gpu=65_536, n_steps=100_000, p_in=P_C2,
               what="65536 dbeta sweep points x 4 fields x 100000 z-steps, float64, per GPU (C2 inputs of SURVEY 8d)"),
    "c3": dict(baseline="configs[2]", n_waves=4, dtype="f64", pts_per_gpu=1_048_576, n_steps=100_000, p_in=P_GRID, cols=1024,
               what="1048576 sweep points (lambda_p2 x lambda_signal grid, 1024 columns) x 4 fields x 100000 z-steps, "
                    "float64, per GPU; dbeta generated on the device"),
    "c4": dict(baseline="configs[3]", n_waves=4, dtype="f32", pts_per_gpu=131_072, n_steps=1_000_000, p_in=P_GRID, cols=1024,
               what="131072 sweep points per GPU (1/8 of the 1024 x 1024 grid) x 4 fields x 1000000 z-steps, float32; "
                    "dbeta generated on the device"),
    "c5": dict(baseline="configs[4]", n_waves=6, dtype="f64", pts_per_gpu=32_768, n_steps=100_000, p_in=P_SIX, cols=512,
               what="6-wave RHS, 32768 sweep points per GPU (1/8 of the 512 x 512 (Omega1, Omega2) grid) x 6 fields x "
                    "100000 z-steps, float64; dbeta_1, dbeta_2 generated on the device"),
}


def grid_dispersion():
    """Dispersion of the grid workloads (D = 0.1 ps/nm/km, S = 0.02 ps/nm^2/km at the G2 plan's centre)."""
    from psa_amd import dispersion, frequency_plan
    om = frequency_plan.plan_from_wavelengths(1550e-9, 1558e-9, 1540e-9)
    sp = frequency_plan.infer_symmetry_from_omegas(*om)
    return dispersion.dispersion_params_from_D_S(frequency_plan.lambda_from_omega(sp.omega_c), 0.1, 0.02, 0,
                                                 D_units="ps/nm/km", S_units="ps/nm^2/km", dSdlmbd_units="ps/nm^3/km",
                                                 omega_ref=sp.omega_c)


def build_shard(name: str, world: int, rank: int, dev, extra_flags: int = 0, exact_step: bool = False):
    """-> (DeviceSweep of this rank's block, n_global, host_dbeta(pick) -> (dbeta, dbeta2|None) for the parity guard)."""
    from psa_amd import dispersion, frequency_plan
    from psa_amd.phase_matching import PhaseMatchingConfig
    c = CONFIGS[name]
    n_global = c["pts_per_gpu"] * world
    lo, hi = shard_bounds(n_global, world, rank)
    np_dtype = np.float64 if c["dtype"] == "f64" else np.float32
    kw = dict(n_steps=c["n_steps"], z_max=Z_MAX, save_every=SAVE_EVERY, gamma=GAMMA, alpha=ALPHA,
              a0=np.sqrt(c["p_in"]).astype(complex), dtype=np_dtype, check_nan=True, exact_step=exact_step, device=dev,
              extra_flags=extra_flags, pad_to=c["pts_per_gpu"])
    if name == "c2":
        full = np.linspace(*DBETA_RANGE, n_global)
        sweep = DeviceSweep(full[lo:hi], **kw)
        return sweep, n_global, lambda pick: (full[lo:hi][pick], None)
    d = grid_dispersion()
    rows = n_global // c["cols"]
    if c["n_waves"] == 4:
        lam2, lam3 = np.linspace(1552e-9, 1562e-9, rows), np.linspace(1540e-9, 1565e-9, c["cols"])
        sweep = DeviceSweep(n_local=hi - lo, **kw)
        sweep.fill_dbeta_grid(nat.dbeta_model(d, PhaseMatchingConfig()), 1550e-9, lam2, lam3, first=lo)

        def host_dbeta(pick):
            from psa_amd.phase_matching import compute_phase_mismatch_batch
            i = lo + np.asarray(pick)
            om, ok = frequency_plan.plan_from_wavelengths_batch(1550e-9, lam2[i // c["cols"]], lam3[i % c["cols"]])
            db, ok2 = compute_phase_mismatch_batch(om, d, PhaseMatchingConfig())
            assert ok.all() and ok2.all()
            return db, None
        return sweep, n_global, host_dbeta
    w1, w2 = frequency_plan.omega_from_lambda(1550e-9), frequency_plan.omega_from_lambda(1558e-9)
    wd = 0.5 * (w1 - w2)
    O1, O2 = np.linspace(2e12, 2.4e13, rows), np.linspace(3e12, 2.0e13, c["cols"])
    sweep = DeviceSweep(n_local=hi - lo, **kw)
    sweep.fill_dbeta_pairs(nat.dbeta_model(d, None, even_orders=(2, 4)), wd, O1, O2, first=lo)

    def host_pairs(pick):
        i = lo + np.asarray(pick)
        return (dispersion.delta_beta_symmetric_array(wd, O1[i // c["cols"]], d),
                dispersion.delta_beta_symmetric_array(wd, O2[i % c["cols"]], d))
    return sweep, n_global, host_pairs


def cpu_baseline(name: str, target_seconds: float = 6.0) -> dict:
    """CPU legs, all on a bounded sample of the SAME workload (called BEFORE the GPU is initialised):
      * value: the oracle's C port (OpenMP over points, all host cores), P points at the configuration's full step count
        with P sized from a probe so it takes ~target_seconds of wall time (float64 arithmetic also for c4);
      * c2 only: the structurally faithful NumPy per-point restatement (the reference's own loop shape) on 1 core and on
        every core (one process per core, forked before any OpenMP thread exists), and a batched-NumPy form, for context
        (SURVEY 8(d)(ii))."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle as O
    c = CONFIGS[name]
    nw, n_z = c["n_waves"], c["n_steps"]
    a0 = np.sqrt(c["p_in"]).astype(complex)
    O.lib()
    # a one-GPU box's CPU share is 16 cores (more hardware threads may be visible); PSA_BENCH_CPU_CORES overrides
    cores = max(1, min(O.max_threads(), len(os.sched_getaffinity(0)), int(os.environ.get("PSA_BENCH_CPU_CORES", "16"))))
    extra = {}
    if name == "c2":
        np_steps = n_z // 50
        t = time.perf_counter()
        O.np_integrate(a0, z_max=Z_MAX * 0.02, dz=Z_MAX / n_z, save_every=SAVE_EVERY, check_nan=True, gamma=GAMMA,
                       alpha=ALPHA, dbeta=0.01)
        np_wall = time.perf_counter() - t
        t = time.perf_counter()
        O.np_integrate_all_cores(a0, z_max=Z_MAX * 0.02, dz=Z_MAX / n_z, dbetas=np.linspace(*DBETA_RANGE, cores), procs=cores)
        np_all_wall = time.perf_counter() - t
        nb_pts, nb_steps = 4096, 100
        t = time.perf_counter()
        O.np_sweep_batched(np.linspace(*DBETA_RANGE, nb_pts), z_max=Z_MAX * nb_steps / n_z, n=nb_steps, gamma=GAMMA,
                           alpha=ALPHA, a0=a0)
        nb_wall = time.perf_counter() - t
        extra = {"numpy_restatement_1core": {"value": nw * np_steps / np_wall,
                                             "sample": f"1 point x {np_steps} z-steps, oracle.np_integrate "
                                                       "(reference-shaped Python loop), 1 core"},
                 "numpy_restatement_all_cores": {"value": cores * nw * np_steps / np_all_wall, "cores": cores,
                                                 "sample": f"{cores} points x {np_steps} z-steps, one process per core "
                                                           "(multiprocessing.Pool, pool start-up included)"},
                 "numpy_batched_1core": {"value": nb_pts * nw * nb_steps / nb_wall,
                                         "sample": f"{nb_pts} points x {nb_steps} z-steps, oracle.np_sweep_batched "
                                                   "(arrays over sweep points)"}}

    def run(pts):
        db = np.linspace(*DBETA_RANGE, pts)
        d2 = {"dbeta2": 0.5 * db[::-1]} if nw == 6 else {}
        t0 = time.perf_counter()
        O.sweep(db, z_max=Z_MAX, n=n_z, save_every=SAVE_EVERY, gamma=GAMMA, alpha=ALPHA, a0=a0, threads=cores, **d2)
        return time.perf_counter() - t0

    # the C port last: OpenMP worker threads appear only now
    O.sweep(np.zeros(cores), z_max=1.0, n=100, save_every=SAVE_EVERY, gamma=GAMMA, alpha=ALPHA, a0=a0, threads=cores,
            **({"dbeta2": np.zeros(cores)} if nw == 6 else {}))  # warm
    probe_pts = cores
    probe = run(probe_pts)                                        # one point per core at full length
    pts = max(cores, min(1 << 16, int(probe_pts * max(1.0, target_seconds / max(probe, 1e-3)))))
    wall = run(pts) if pts > probe_pts else probe
    out = {"value": pts * nw * n_z / wall, "unit": "field-point updates/s", "cores": cores, "kind": "port",
           "sample": f"{pts} sweep points x {nw} fields x {n_z} z-steps of the bench workload, "
                     f"oracle/psa_oracle.c (scalar C99 float64, OpenMP over points), {wall:.1f} s wall"}
    out.update(extra)
    return out


def profile_facts() -> dict:
    """What the committed rocprofv3 runs measured for each configuration's dominant kernel (profiles/kernels.json, written
    by tools/profile_summary.py): kernel symbol, VALU instructions per wave-step, HBM bytes per launch."""
    try:
        with open(os.path.join(ROOT, "profiles", "kernels.json")) as f:
            return json.load(f)
    except Exception:
        return {}


def emit(out: dict, saved_stdout_fd: int) -> None:
    sys.stdout.flush()
    os.dup2(saved_stdout_fd, 1)
    print(json.dumps(out), flush=True)
    os.dup2(2, 1)


TRAJ_VARIANTS = {
    # name: (n_waves, dtype, points, z-steps, flags, what)
    "c2": (4, "f64", 262_144, 400, 0, "4 waves, float64, one lane per point"),
    "c4": (4, "f32", 524_288, 400, 0, "4 waves, float32, two points per lane (packed)"),
    "c5": (6, "f64", 262_144, 400, 0, "6 waves, float64, one lane per point"),
    "c2split": (4, "f64", 32_768, 3200, "split", "4 waves, float64, two lanes per point (a sweep smaller than the chip)"),
    "c5split": (6, "f64", 32_768, 2000, "split", "6 waves, float64, two lanes per point (BASELINE config 5's shard shape)"),
}


def trajectory_mode(args, dev, saved_stdout_fd) -> None:
    """integrate_fixed_step's strided save with save_every = 1 (integrators.py:137-140) for a whole sweep: the one
    regime where this path is HBM-bound (16 B per wave per point per step, ~10 flop/B in float64).  Single GPU.
    `--config c2` (default) | c4 (float32 packed) | c5 (6 waves), `--split` for the two-lane float64 layout."""
    key = args.config + ("split" if args.split else "")
    if key not in TRAJ_VARIANTS:
        raise SystemExit(f"--mode trajectory: no variant {key!r} (have {sorted(TRAJ_VARIANTS)})")
    nw, dt, pts, nz, fl, what = TRAJ_VARIANTS[key]
    np_dtype = np.float64 if dt == "f64" else np.float32
    p_in = P_C2 if nw == 4 else P_SIX
    a0 = np.sqrt(p_in).astype(complex)
    db = np.linspace(*DBETA_RANGE, pts).astype(np_dtype)
    db2 = (0.5 * db[::-1]).astype(np_dtype) if nw == 6 else None
    sweep = DeviceSweep(db, dbeta2_local=db2, n_steps=nz, z_max=nz * 0.01, save_every=1, gamma=GAMMA, alpha=ALPHA, a0=a0,
                        dtype=np_dtype, check_nan=True, device=dev,
                        extra_flags=(nat.OPT_SPLIT_POINT if fl == "split" else (nat.OPT_ONE_LANE if dt == "f64" else 0)))
    traj_bytes = sweep.enable_trajectory()
    for _ in range(args.warmup):
        sweep.launch()
    torch.cuda.synchronize()
    events = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    t0 = time.perf_counter()
    for e0, e1 in events:
        e0.record()
        sweep.launch()
        e1.record()
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    kern_ms = float(np.mean([e0.elapsed_time(e1) for e0, e1 in events]))
    # guard: rows of 3 points against the oracle
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle as O
    pick = [0, pts // 2 + 1, pts - 1]
    tr = sweep.traj[:, :, pick, :].cpu().numpy().astype(np.float64)      # [rows][nw][3][2]
    err = 0.0
    for j, p in enumerate(pick):
        kw6 = {"dbeta2": float(db2[p])} if nw == 6 else {}
        z, A, _ = O.integrate(a0, z_max=nz * 0.01, n=nz, save_every=1, gamma=GAMMA, alpha=ALPHA, dbeta=float(db[p]), **kw6)
        got = tr[:, :, j, 0] + 1j * tr[:, :, j, 1]
        err = max(err, float(np.max(np.abs(got - A) / np.abs(A))))
    if not err < (1e-9 if dt == "f64" else 1e-4):
        raise SystemExit(f"trajectory bench failed its parity guard: {err}")
    facts = profile_facts().get("trajectory" if key == "c2" else "traj_" + key, {})
    alg_bytes = traj_bytes + sweep.layout.bytes_per_point() * pts + sweep.layout.es * pts * (2 if nw == 6 else 1)
    gbs = alg_bytes / (kern_ms * 1e-3) / 1e9
    out = {"metric": "RK4 field-point updates/sec (sweep_pts x n_fields x n_zsteps / wall_s)",
           "value": pts * nw * nz * args.steps / wall, "unit": "field-point updates/s", "n_gpus": 1,
           "steps": args.steps, "warmup": args.warmup, "ms_per_step": wall / args.steps * 1e3, "higher_is_better": True,
           "scaling": "weak", "vs_baseline": None, "dtype": dt, "data": "synthetic",
           "config": {"workload": f"trajectory mode (NOT the headline config): {pts} sweep points x {nw} fields x {nz} z-steps, "
                                  f"every step saved to HBM (save_every = 1); {what}", "name": "traj_" + key,
                      "sweep_pts": pts, "n_fields": nw, "n_zsteps": nz, "save_every": 1},
           "roofline": {"kernel": facts.get("kernel"),
                        "bound": "hbm", "achieved": gbs, "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": gbs / PEAK_HBM_GBS,
                        "algorithmic_bytes_per_launch": alg_bytes, "kernel_ms_avg": kern_ms,
                        "traffic": facts.get("hbm_bytes_per_launch"), "traffic_source": facts.get("source"),
                        "store_only_ceiling_gbs": profile_facts().get("trajectory", {}).get("store_only_ceiling_gbs"),
                        "note": "one (re, im) pair per wave per point per saved row, coalesced 1-KiB wave stores; a kernel that "
                                "does nothing but these stores reaches store_only_ceiling_gbs on this chip "
                                "(tools/hbm_write_peak.hip)"},
           "verify": {"trajectories_checked_vs_oracle": 3, "max_rel_err": err}}
    emit(out, saved_stdout_fd)


def self_launch_command(n_gpus: int, argv, port: int | None = None) -> list:
    """The command that starts one rank per GPU for `python bench.py --gpus N` (N > 1) run WITHOUT a launcher: the
    contract's own `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ...` line
    with this invocation's arguments passed through unchanged."""
    if port is None:
        import socket
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={int(n_gpus)}",
            "--master-addr", "127.0.0.1", "--master-port", str(int(port)), os.path.abspath(__file__), *argv]


def self_launch(n_gpus: int, argv) -> int:
    """Start the ranks as fresh child processes and wait for them.  Called before this process has made any GPU call
    (`import torch` does not initialise HIP; torch.cuda.* would) and never replaces this process: rank 0's JSON line
    reaches our stdout through the launcher, the exit status is the launcher's (non-zero if any rank failed)."""
    import subprocess
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "1")
    return subprocess.call(self_launch_command(n_gpus, list(argv)), env=env)


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None, help="timed passes (default 10; 100 in trajectory mode)")
    ap.add_argument("--warmup", type=int, default=None, help="untimed passes first (default 2; 30 in trajectory mode, "
                                                             "where the first ~20 launches run at a clock still ramping)")
    ap.add_argument("--config", choices=sorted(CONFIGS), default="c2",
                    help="BASELINE configuration per GPU (default c2 = the headline workload)")
    ap.add_argument("--no-cpu-baseline", action="store_true", help="skip the ~15 s CPU leg (profiling runs)")
    ap.add_argument("--block64", action="store_true", help="64-thread workgroups")
    ap.add_argument("--one-lane", action="store_true", help="float64: never split a point over two lanes (A/B for c5)")
    ap.add_argument("--split", action="store_true", help="trajectory mode: the two-lane float64 layout (32 768 points)")
    ap.add_argument("--mode", choices=["summary", "trajectory"], default="summary",
                    help="summary (default): the BASELINE workload.  trajectory: the path's HBM-bound regime -- every "
                         "step saved (save_every = 1), 262 144 points x 400 z-steps, 6.7 GB of rows per launch; reports "
                         "an HBM roofline object.  Not the headline metric.")
    ap.add_argument("--exact-step", action="store_true",
                    help="the reference's per-step finite test (integrators.py:132-135).  Default for the float64 workloads, "
                         "where the exact first_bad_step costs nothing in the loop (block test + replay of a failing block); "
                         "for float32 (c4) this flag turns on the packed kernel's in-loop test")
    ap.add_argument("--d2h", choices=["overlap", "inline", "off"], default="overlap",
                    help="how a pass's outputs reach pinned host memory inside the timed region: on a second stream under the "
                         "next pass's kernel (default), on the launch stream (A/B), or not at all (outputs left in HBM: the "
                         "round-1/2 clock, NOT the metric's)")
    ap.add_argument("--block-check", action="store_true", help="float64: test once per saved row only (first_bad_step = the "
                                                                "last step of the first non-finite block)")
    args = ap.parse_args()
    if args.steps is None:
        args.steps = 100 if args.mode == "trajectory" else 10
    if args.warmup is None:
        args.warmup = 30 if args.mode == "trajectory" else 2
    cfg = CONFIGS[args.config]
    nw, n_z, pts = cfg["n_waves"], cfg["n_steps"], cfg["pts_per_gpu"]

    # `python bench.py --gpus N` without a launcher: start the N ranks ourselves (nothing has touched the GPU yet)
    if args.gpus > 1 and "RANK" not in os.environ and "WORLD_SIZE" not in os.environ:
        if args.mode == "trajectory":
            raise SystemExit("--mode trajectory is a single-GPU measurement")
        raise SystemExit(self_launch(args.gpus, sys.argv[1:]))

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")

    # Exactly ONE line may reach stdout (the JSON, from rank 0).  RCCL prints a version banner with printf at
    # communicator creation, so fd 1 is pointed at stderr for the run and restored only for the final print.
    sys.stdout.flush()
    saved_stdout_fd = os.dup(1)
    os.dup2(2, 1)
    # CPU legs first: one of them forks worker processes, which must happen before this process initialises the GPU
    # runtime -- so the GPU is detected from the device node, not through torch / HIP
    cpu_leg = None
    if world == 1 and args.mode == "summary" and not args.no_cpu_baseline and os.path.exists("/dev/kfd"):
        cpu_leg = cpu_baseline(args.config)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the sweep has no CPU fallback")
    # PSA_BENCH_BACKEND=gloo rehearses the N > 1 code path on a box with fewer GPUs than ranks (ranks then share devices and
    # the gather is staged through the host); the measured configuration is always nccl = RCCL, one GPU per rank.
    backend = os.environ.get("PSA_BENCH_BACKEND", "nccl")
    n_dev = torch.cuda.device_count()
    if backend == "nccl" and local_rank >= n_dev:
        raise SystemExit(f"rank {rank}: LOCAL_RANK {local_rank} but only {n_dev} GPU(s) visible (RCCL needs one GPU per rank)")
    local_dev = local_rank % n_dev
    torch.cuda.set_device(local_dev)
    dev = torch.device("cuda", local_dev)
    # Under torch.distributed.run (RANK set) the gather path is used even for one rank, so the RCCL plumbing can be
    # rehearsed on a one-GPU box; plain `python bench.py` stays collective-free.
    use_dist = world > 1 or ("RANK" in os.environ and os.environ.get("PSA_BENCH_DIST_ON_ONE", "0") == "1")
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)   # nccl == RCCL on ROCm
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    if args.mode == "trajectory":
        return trajectory_mode(args, dev, saved_stdout_fd)

    # -- synthetic inputs of the workload, resident in HBM (rank's contiguous block of the global sweep)
    flags = (nat.OPT_BLOCK64 if args.block64 else 0) | (nat.OPT_ONE_LANE if args.one_lane else 0)
    exact = args.exact_step or (cfg["dtype"] == "f64" and not args.block_check)
    sweep, n_global, host_dbeta = build_shard(args.config, world, rank, dev, extra_flags=flags, exact_step=exact)
    p0_sig = float(cfg["p_in"][2])

    def one_step(evs=None):
        """launch -> gain reduction -> (all_gather) -> device-to-host copy of the outputs into pinned memory.  The copy is
        enqueued on a second stream and the sweep switches to its other record, so pass k's copy overlaps pass k+1's kernel;
        the synchronize that ends the timed region waits for the last copy: the clock is SURVEY 8(d)'s "kernel launch to
        completed D2H of outputs (and completed gather)"."""
        if evs is not None:
            evs[0].record()                      # torch's current stream == the stream the kernel is launched on
        sweep.launch()
        if evs is not None:
            evs[1].record()
        sweep.summarize(p0_sig, mode="max", gain_db=True)   # the drivers' per-point gain + argmax, on device
        g = sweep.gather() if use_dist else None
        if evs is not None:
            evs[2].record()
        if args.d2h != "off":
            sweep.stage_to_host(g, overlap=(args.d2h == "overlap"))
        return g

    if args.d2h != "off":
        sweep.reserve_staging(world if use_dist else 1)      # page-locked images, second record: never inside the clock
    for _ in range(args.warmup):
        one_step()
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    events = [tuple(torch.cuda.Event(enable_timing=True) for _ in range(3)) for _ in range(args.steps)]
    t0 = time.perf_counter()
    gathered = None
    for evs in events:
        gathered = one_step(evs)
    torch.cuda.synchronize()                     # every stream of the device: kernels, gathers and host copies
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    if use_dist:
        tw = torch.tensor([wall], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(tw, op=dist.ReduceOp.MAX)
        wall = float(tw.item())
    kern_ms = float(np.mean([e[0].elapsed_time(e[1]) for e in events]))
    resident_ms = float(np.mean([e[0].elapsed_time(e[2]) for e in events]))   # launch .. gain (.. gather), outputs left in HBM
    if args.d2h == "off":
        sweep.stage_to_host(gathered)              # after the clock has stopped: the guard below reads the host image
        torch.cuda.synchronize()
    host_words, host_summ = sweep.host_result()

    # -- post-run guard (not timed): the numbers just produced are the right numbers
    # ... read from the PINNED HOST image the timed passes delivered (this rank's row of the gathered records for N > 1)
    host_words = host_words.reshape(world if use_dist else 1, -1)
    a_h, pe_h, pm_h, fb_h = sweep.layout.unpack(host_words[rank if use_dist else 0], pts)
    from psa_amd.sweep import SweepResult
    res = SweepResult(a_h, pe_h, pm_h, fb_h, n_z, SAVE_EVERY, kern_ms)
    if use_dist:
        assert gathered is not None and torch.equal(gathered[rank], sweep.record)
        assert np.array_equal(host_words[rank], sweep.record.cpu().numpy())
    verify = None
    if rank == 0:
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        import oracle as O
        f32 = cfg["dtype"] == "f32"
        pick = np.array([0, pts // 3, pts // 2, pts - 1])
        db_host, db2_host = host_dbeta(pick)
        db_dev = sweep.dbeta.cpu().numpy()[pick].astype(np.float64)
        err_db = float(np.max(np.abs(db_dev - db_host) / np.abs(db_host)))      # device-generated vs host producer
        ref = O.sweep(db_dev, z_max=Z_MAX, n=n_z, save_every=SAVE_EVERY, gamma=GAMMA, alpha=ALPHA,
                      a0=np.sqrt(cfg["p_in"]).astype(complex), threads=4,
                      **({"dbeta2": sweep.dbeta2.cpu().numpy()[pick].astype(np.float64)} if nw == 6 else {}))
        err = float(np.max(np.abs(res.a_end[pick].astype(complex) - ref["a_end"]) / np.abs(ref["a_end"])))
        gain_dev = host_summ[0].astype(np.float64)
        gain_ref = O.gain_from_summary(ref["p_max"], ref["first_bad_step"], p0_sig, "db")
        best_i, n_fin = (int(v) for v in host_summ[1])
        err_gain = float(np.max(np.abs(gain_dev[pick] - gain_ref)))
        verify = {"points_checked_vs_oracle": int(pick.size), "max_rel_err_a_end": err, "max_err_gain_db": err_gain,
                  "max_rel_err_dbeta_device_vs_host": err_db,
                  "all_finite": bool((res.first_bad_step == -1).all()), "best_gain_db": float(host_summ[2][0]),
                  "best_index": best_i}
        tol_a, tol_g, tol_db = (1e-4, 5e-4, 1e-7) if f32 else (1e-9, 5e-9, 4e-16)
        if not (err < tol_a and err_gain < tol_g and err_db <= tol_db and verify["all_finite"] and n_fin == pts
                and abs(gain_dev[best_i] - np.max(gain_dev)) <= (1e-6 if f32 else 0.0)):
            raise SystemExit(f"bench result failed its parity guard: {verify}")

    if rank == 0:
        facts = profile_facts().get("c5one" if (args.config == "c5" and args.one_lane) else args.config, {})
        updates_per_step = n_global * nw * n_z
        value = updates_per_step * args.steps / wall
        rk4_steps_per_launch = pts * n_z
        flops = FLOPS_PER_RK4_STEP[nw] * rk4_steps_per_launch
        tflops = flops / (kern_ms * 1e-3) / 1e12
        peak = PEAK_TFLOPS[cfg["dtype"]]
        alg_bytes = sweep.layout.bytes_per_point() * pts + sweep.layout.es * pts * (2 if nw == 6 else 1)   # record + dbeta
        gbs = alg_bytes / (kern_ms * 1e-3) / 1e9
        ipw = facts.get("valu_insts_per_wave_step")
        # lanes that carry one sweep point: 1 (one lane per point), 0.5 (float32 packed: two points per lane), 2 (split)
        lanes_per_point = facts.get("lanes_per_point", 1.0)
        # -- the same launch in HARDWARE units (what the vector ALU was asked to do, not what the algorithm is credited with):
        #    every VALU wave-instruction holds its SIMD for 4 cycles (wave64 on 16 lanes), FP64 and packed FP32 alike.
        kern_s = kern_ms * 1e-3
        simds = 4 * torch.cuda.get_device_properties(dev).multi_processor_count
        waves = -(-int(round(pts * lanes_per_point)) // 64)
        issue_nominal = None if ipw is None else ipw * waves * n_z * 4 / (simds * NOMINAL_GHZ * 1e9 * kern_s)
        ex_lane = facts.get("executed_flops_per_lane_step")      # (2 FMA + MUL + ADD [x2 packed]) per lane per z-step, PMC
        executed = None
        if ex_lane is not None:
            ex_tflops = ex_lane * 64 * waves * n_z / kern_s / 1e12
            executed = {"flops_per_lane_step": ex_lane, "achieved": ex_tflops, "unit": "TFLOP/s", "frac": ex_tflops / peak,
                        "counters": facts.get("executed_flops_counters"),
                        "note": "flops the VALU executed per the SQ_INSTS_VALU_{FMA,MUL,ADD} pass of profiles/ (an FMA = 2), "
                                "scaled to this run's kernel time: the fraction of the dense vector peak actually used"}
        out = {
            "metric": "RK4 field-point updates/sec (sweep_pts x n_fields x n_zsteps / wall_s)",
            "value": value, "unit": "field-point updates/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": wall / args.steps * 1e3, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": cfg["dtype"], "data": "synthetic",
            "config": {"workload": f"BASELINE {cfg['baseline']}: {cfg['what']}", "name": args.config,
                       "sweep_pts_per_gpu": pts, "sweep_pts_total": n_global, "n_fields": nw, "n_zsteps": n_z,
                       "save_every": SAVE_EVERY, "check_nan": True,
                       "first_bad_step": "exact (per-step test semantics of integrators.py:132-135)" if exact else
                                         "last step of the first non-finite save block",
                       "parallelism": (f"sweep sharded x{world}, one RCCL all_gather per pass" if backend == "nccl" else
                                       f"REHEARSAL: {world} ranks over {n_dev} GPU(s), {backend} gather staged through the host")
                       if world > 1 else "single GPU"},
            "rk4_steps_per_s": value / nw,
            # the clock of `value`: launch -> gain reduction -> (gather) -> outputs complete in pinned host memory,
            # the copy of pass k overlapped with the kernel of pass k+1 on a second stream.  The figure with the outputs
            # left in HBM (launch .. gather, HIP events on the launch stream) is kept beside it:
            "value_clock": ("kernel launch to completed D2H of the outputs (and completed gather), K passes back to back"
                            if args.d2h != "off" else "DEVICE-RESIDENT (--d2h off): outputs left in HBM, not the metric's clock"),
            "d2h": args.d2h,
            "value_device_resident": updates_per_step / (resident_ms * 1e-3),   # rank 0's events
            "device_resident_ms_per_step": resident_ms,
            "d2h_bytes_per_step": int(host_words.nbytes + sum(t.nbytes for t in host_summ)),
            "roofline": {
                "kernel": facts.get("kernel"),
                "bound": "mfma",   # the contract's label for the COMPUTE roofline (enum hbm | mfma); see bound_detail
                "bound_detail": "compute-bound on the VECTOR ALU: the kernel issues 0 MFMA instructions (elementwise complex "
                                "recurrence, nothing to contract); for float64 the MI355X vector and matrix dense peaks are "
                                "the same 78.6 TFLOP/s, for float32 the peak used is the packed-vector rate 157.3 TFLOP/s",
                "achieved": tflops, "peak": peak, "unit": "TFLOP/s", "frac": tflops / peak,
                "frac_is": "ALGORITHMIC flops (flops_per_rk4_step x points x z-steps, SURVEY 8d accounting) / kernel time / "
                           "peak -- the contract's definition; it credits 652 flops where the kernel executes 524 in 298 "
                           "instructions, so it can exceed the issue-slot utilisation: read issue_frac_nominal and executed",
                "issue_frac_nominal": issue_nominal, "issue_frac_nominal_is": "VALU wave-instructions x 4 cycles / (SIMDs x "
                                                                              f"{NOMINAL_GHZ} GHz x kernel time)",
                "executed": executed,
                "flops_per_launch": flops, "flops_per_rk4_step": FLOPS_PER_RK4_STEP[nw], "kernel_ms_avg": kern_ms,
                "waves_per_launch": waves, "simds": simds,
                # VALU wave-instructions per z-step (SQ_INSTS_VALU, profiles/kernels.json) x 4 issue cycles: the clock the chip
                # would need if the vector pipe never idled = a LOWER bound on the clock it held.  Boxes of the pool differ
                # by ~10 % here (DVFS / silicon), which moves `frac` with no change in the code.
                "valu_issue_ghz_equiv": None if ipw is None else ipw * 4 * n_z / (kern_ms * 1e-3) / 1e9,
                "valu_insts_per_wave_step": ipw, "lanes_per_point": lanes_per_point,
                "traffic": facts.get("hbm_bytes_per_launch"), "traffic_source": facts.get("source"),
                "note": "elementwise complex recurrence: no MFMA, ~2e-4 B per update -> vector issue is the binding roofline "
                        "(DESIGN.md section 5); the HBM view of the same launch follows",
                "hbm": {"bound": "hbm", "achieved": gbs, "peak": PEAK_HBM_GBS, "unit": "GB/s",
                        "frac": gbs / PEAK_HBM_GBS, "algorithmic_bytes_per_launch": alg_bytes},
            },
            "verify": verify,
        }
        if cpu_leg is not None:
            out["cpu_baseline"] = cpu_leg
            out["gpu_over_cpu"] = value / cpu_leg["value"]
        if use_dist:
            torch.cuda.synchronize()
        emit(out, saved_stdout_fd)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
