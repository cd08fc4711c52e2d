#!/usr/bin/env python3
"""bench.py -- RK4 field-point updates/s of the HIP sweep on MI355X (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One bench "step" = one pass of the hot path over one batch: BASELINE config 2 per GPU -- 65 536 dbeta sweep
points x 4 fields x 100 000 z-steps, float64, dbeta = linspace(-0.05, 0.05), gamma = 0.0115, alpha = 1.15e-4,
P = (0.5, 0.5, 1e-5, 1e-5) W, L = 1000 m, save_every = 10, check_nan on (SURVEY 8d "C2").  Inputs are resident in
HBM before the timed region; a step is the sweep kernel on the rank's shard, the on-device gain reduction (per-point gain
+ argmax, the sweep drivers' summary) and, for N > 1, the single RCCL all_gather of the 88 B/point output record.  Weak scaling: every rank owns 65 536 points (rank r gets the r-th
contiguous block of the global linspace).

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

# ---- workload: BASELINE.json configs[1] ------------------------------------------------------------------------
PTS_PER_GPU = 65_536
N_FIELDS = 4
N_ZSTEPS = 100_000
Z_MAX = 1000.0
SAVE_EVERY = 10
GAMMA, ALPHA = 0.0115, 1.15e-4
P_IN = np.array([0.5, 0.5, 1e-5, 1e-5])
DBETA_RANGE = (-0.05, 0.05)

# ---- accounting agreed in BASELINE.md section 2 / SURVEY 8(d) ---------------------------------------------------
FLOPS_PER_RK4_STEP = 652          # FP64 flops per sweep point per z-step (4-wave), + 2 sincos not counted
BYTES_PER_POINT = 96              # dbeta in (8) + A_end (64) + p_end (8) + p_max (8) + first_bad_step (8)
PEAK_FP64_VALU_TFLOPS = 78.6      # 256 CU x 4 SIMD x 16 lanes x 2 flop x 2.4 GHz (MI355X vector FP64)
PEAK_HBM_GBS = 8000.0             # MI355X_MICROARCH.md: HBM3E 8 TB/s spec


def cpu_baseline(target_seconds: float = 6.0) -> dict:
    """CPU legs, all on a bounded sample of the SAME workload (called BEFORE the GPU is initialised):
      * value: the oracle's C port (OpenMP over points, all host cores), P points x 100 000 steps with P sized from a
        probe so it takes ~target_seconds of wall time (= target_seconds x cores of CPU work);
      * the structurally faithful NumPy per-point restatement (the reference's own loop shape) on 1 core and on every
        core (one process per core, forked before any OpenMP thread exists), and a batched-NumPy form, for context
        (SURVEY 8(d)(ii))."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle as O
    a0 = np.sqrt(P_IN).astype(complex)
    O.lib()
    # a one-GPU box's CPU share is 16 cores (more hardware threads may be visible); PSA_BENCH_CPU_CORES overrides
    cores = max(1, min(O.max_threads(), len(os.sched_getaffinity(0)), int(os.environ.get("PSA_BENCH_CPU_CORES", "16"))))
    np_steps = N_ZSTEPS // 50
    t = time.perf_counter()
    O.np_integrate(a0, z_max=Z_MAX * 0.02, dz=Z_MAX / N_ZSTEPS, save_every=SAVE_EVERY, check_nan=True, gamma=GAMMA,
                   alpha=ALPHA, dbeta=0.01)
    np_wall = time.perf_counter() - t
    t = time.perf_counter()
    O.np_integrate_all_cores(a0, z_max=Z_MAX * 0.02, dz=Z_MAX / N_ZSTEPS, dbetas=np.linspace(*DBETA_RANGE, cores), procs=cores)
    np_all_wall = time.perf_counter() - t
    nb_pts, nb_steps = 4096, 100
    t = time.perf_counter()
    O.np_sweep_batched(np.linspace(*DBETA_RANGE, nb_pts), z_max=Z_MAX * nb_steps / N_ZSTEPS, n=nb_steps, gamma=GAMMA,
                       alpha=ALPHA, a0=a0)
    nb_wall = time.perf_counter() - t
    # the C port last: OpenMP worker threads appear only now
    O.sweep(np.zeros(cores), z_max=1.0, n=100, save_every=SAVE_EVERY, gamma=GAMMA, alpha=ALPHA, a0=a0, threads=cores)  # warm
    probe_pts = 4 * cores
    t = time.perf_counter()
    O.sweep(np.linspace(*DBETA_RANGE, probe_pts), z_max=Z_MAX, n=N_ZSTEPS, save_every=SAVE_EVERY, gamma=GAMMA,
            alpha=ALPHA, a0=a0, threads=cores)
    probe = time.perf_counter() - t                               # four points per core at full length
    pts = max(cores, min(1 << 16, int(probe_pts * max(1.0, target_seconds / max(probe, 1e-3)))))
    db = np.linspace(*DBETA_RANGE, pts)
    t = time.perf_counter()
    O.sweep(db, z_max=Z_MAX, n=N_ZSTEPS, save_every=SAVE_EVERY, gamma=GAMMA, alpha=ALPHA, a0=a0, threads=cores)
    wall = time.perf_counter() - t
    return {"value": pts * N_FIELDS * N_ZSTEPS / wall, "unit": "field-point updates/s", "cores": cores, "kind": "port",
            "sample": f"{pts} sweep points x {N_FIELDS} fields x {N_ZSTEPS} z-steps of the bench workload, "
                      f"oracle/psa_oracle.c (scalar C99, OpenMP over points), {wall:.1f} s wall",
            "numpy_restatement_1core": {"value": N_FIELDS * np_steps / np_wall,
                                        "sample": f"1 point x {np_steps} z-steps, oracle.np_integrate "
                                                  "(reference-shaped Python loop), 1 core"},
            "numpy_restatement_all_cores": {"value": cores * N_FIELDS * np_steps / np_all_wall, "cores": cores,
                                            "sample": f"{cores} points x {np_steps} z-steps, one process per core "
                                                      "(multiprocessing.Pool, pool start-up included)"},
            "numpy_batched_1core": {"value": nb_pts * N_FIELDS * nb_steps / nb_wall,
                                    "sample": f"{nb_pts} points x {nb_steps} z-steps, oracle.np_sweep_batched "
                                              "(arrays over sweep points)"}}


def measured_traffic() -> dict | None:
    """HBM bytes per launch from rocprofv3 PMC passes, if a summary has been committed under profiles/."""
    path = os.path.join(ROOT, "profiles", "hbm_traffic.json")
    try:
        with open(path) as f:
            return json.load(f)
    except Exception:
        return None


def trajectory_mode(args, dev, saved_stdout_fd) -> None:
    """integrate_fixed_step's strided save with save_every = 1 (integrators.py:137-140) for a whole sweep: the one
    regime where this path is HBM-bound (64 B per point per step, ~10 flop/B).  Single GPU."""
    pts, nz = 262_144, 400
    sweep = DeviceSweep(np.linspace(*DBETA_RANGE, pts), n_steps=nz, z_max=nz * 0.01, save_every=1, gamma=GAMMA,
                        alpha=ALPHA, a0=np.sqrt(P_IN).astype(complex), check_nan=True, device=dev)
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
    tr = sweep.traj[:, :, [0, pts // 2, pts - 1], :].cpu().numpy()       # [rows][4][3][2]
    err = 0.0
    for j, p in enumerate((0, pts // 2, pts - 1)):
        z, A, _ = O.integrate(np.sqrt(P_IN).astype(complex), z_max=nz * 0.01, n=nz, save_every=1, gamma=GAMMA, alpha=ALPHA,
                              dbeta=float(np.linspace(*DBETA_RANGE, pts)[p]))
        got = tr[:, :, j, 0] + 1j * tr[:, :, j, 1]
        err = max(err, float(np.max(np.abs(got - A) / np.abs(A))))
    if not err < 1e-9:
        raise SystemExit(f"trajectory bench failed its parity guard: {err}")
    alg_bytes = traj_bytes + BYTES_PER_POINT * pts
    gbs = alg_bytes / (kern_ms * 1e-3) / 1e9
    out = {"metric": "RK4 field-point updates/sec (sweep_pts x n_fields x n_zsteps / wall_s)",
           "value": pts * N_FIELDS * nz * args.steps / wall, "unit": "field-point updates/s", "n_gpus": 1,
           "steps": args.steps, "warmup": args.warmup, "ms_per_step": wall / args.steps * 1e3, "higher_is_better": True,
           "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
           "config": {"workload": "trajectory mode (NOT the headline config): 262144 sweep points x 4 fields x 400 z-steps, "
                                  "float64, every step saved to HBM (save_every = 1)", "sweep_pts": pts, "n_zsteps": nz,
                      "save_every": 1},
           "roofline": {"kernel": "psa::rk4_sweep_kernel<double, 4, CHECK_BLOCK, true, 256>", "bound": "hbm",
                        "achieved": gbs, "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": gbs / PEAK_HBM_GBS,
                        "algorithmic_bytes_per_launch": alg_bytes, "kernel_ms_avg": kern_ms, "traffic": None,
                        "note": "64 B per point per saved row, coalesced 512-B wave stores; the guide's measured "
                                "achievable HBM rate is 6.3 TB/s"},
           "verify": {"trajectories_checked_vs_oracle": 3, "max_rel_err": err}}
    sys.stdout.flush()
    os.dup2(saved_stdout_fd, 1)
    print(json.dumps(out), flush=True)
    os.dup2(2, 1)


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--no-cpu-baseline", action="store_true", help="skip the ~15 s CPU leg (profiling runs)")
    ap.add_argument("--block64", action="store_true", help="64-thread workgroups")
    ap.add_argument("--mode", choices=["summary", "trajectory"], default="summary",
                    help="summary (default): the BASELINE workload.  trajectory: the path's HBM-bound regime -- every "
                         "step saved (save_every = 1), 262 144 points x 400 z-steps, 6.7 GB of rows per launch; reports "
                         "an HBM roofline object.  Not the headline metric.")
    ap.add_argument("--exact-step", action="store_true", help="per-step finite test instead of per save block")
    args = ap.parse_args()

    # Exactly ONE line may reach stdout (the JSON, from rank 0).  RCCL prints a version banner with printf at
    # communicator creation, so fd 1 is pointed at stderr for the run and restored only for the final print.
    sys.stdout.flush()
    saved_stdout_fd = os.dup(1)
    os.dup2(2, 1)

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus N > 1 must be launched with `python -m torch.distributed.run --nproc-per-node N`")
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    # CPU legs first: they fork worker processes, which must happen before this process touches the GPU
    cpu_leg = None
    if world == 1 and args.mode == "summary" and not args.no_cpu_baseline and torch.cuda.device_count() > 0:
        cpu_leg = cpu_baseline()
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the sweep has no CPU fallback")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    # Under torch.distributed.run (RANK set) the gather path is used even for one rank, so the RCCL plumbing can be
    # rehearsed on a one-GPU box; plain `python bench.py` stays collective-free.
    use_dist = world > 1 or ("RANK" in os.environ and os.environ.get("PSA_BENCH_DIST_ON_ONE", "0") == "1")
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)   # nccl == RCCL on ROCm

    if args.mode == "trajectory":
        return trajectory_mode(args, dev, saved_stdout_fd)

    # -- synthetic inputs of the workload, resident in HBM (rank's contiguous block of the global sweep)
    n_global = PTS_PER_GPU * world
    lo, hi = shard_bounds(n_global, world, rank)
    dbeta = np.linspace(*DBETA_RANGE, n_global)[lo:hi]
    sweep = DeviceSweep(dbeta, n_steps=N_ZSTEPS, z_max=Z_MAX, save_every=SAVE_EVERY, gamma=GAMMA, alpha=ALPHA,
                        a0=np.sqrt(P_IN).astype(complex), check_nan=True, exact_step=args.exact_step, device=dev,
                        extra_flags=(nat.OPT_BLOCK64 if args.block64 else 0))

    def one_step(ev0=None, ev1=None):
        if ev0 is not None:
            ev0.record()                         # torch's current stream == the stream the kernel is launched on
        sweep.launch()
        if ev1 is not None:
            ev1.record()
        sweep.summarize(float(P_IN[2]), mode="max", gain_db=True)   # the drivers' per-point gain + argmax, on device
        return sweep.gather() if use_dist else None

    for _ in range(args.warmup):
        one_step()
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    events = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    t0 = time.perf_counter()
    gathered = None
    for e0, e1 in events:
        gathered = one_step(e0, e1)
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    if use_dist:
        tw = torch.tensor([wall], dtype=torch.float64, device=dev)
        dist.all_reduce(tw, op=dist.ReduceOp.MAX)
        wall = float(tw.item())
    kern_ms = float(np.mean([e0.elapsed_time(e1) for e0, e1 in events]))

    # -- post-run guard (not timed): the numbers just produced are the right numbers
    res = sweep.result()
    if use_dist:
        assert gathered is not None and torch.equal(gathered[rank].view(torch.int64), sweep.record.view(torch.int64))
    verify = None
    if rank == 0:
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        import oracle as O
        pick = np.array([0, PTS_PER_GPU // 3, PTS_PER_GPU // 2, PTS_PER_GPU - 1])
        ref = O.sweep(dbeta[pick], z_max=Z_MAX, n=N_ZSTEPS, save_every=SAVE_EVERY, gamma=GAMMA, alpha=ALPHA,
                      a0=np.sqrt(P_IN).astype(complex), threads=4)
        err = float(np.max(np.abs(res.a_end[pick] - ref["a_end"]) / np.abs(ref["a_end"])))
        gain_dev = sweep.gain.cpu().numpy()
        gain_ref = O.gain_from_summary(ref["p_max"], ref["first_bad_step"], P_IN[2], "db")
        best_i, n_fin = (int(v) for v in sweep.best.cpu().numpy())
        err_gain = float(np.max(np.abs(gain_dev[pick] - gain_ref)))
        verify = {"points_checked_vs_oracle": int(pick.size), "max_rel_err_a_end": err, "max_err_gain_db": err_gain,
                  "all_finite": bool((res.first_bad_step == -1).all()), "best_gain_db": float(sweep.best_gain.item()),
                  "best_index": best_i}
        if not (err < 1e-9 and err_gain < 5e-9 and verify["all_finite"] and n_fin == PTS_PER_GPU
                and best_i == int(np.argmax(gain_dev))):
            raise SystemExit(f"bench result failed its parity guard: {verify}")

    if rank == 0:
        updates_per_step = n_global * N_FIELDS * N_ZSTEPS
        value = updates_per_step * args.steps / wall
        rk4_steps_per_launch = PTS_PER_GPU * N_ZSTEPS
        tflops = FLOPS_PER_RK4_STEP * rk4_steps_per_launch / (kern_ms * 1e-3) / 1e12
        alg_bytes = BYTES_PER_POINT * PTS_PER_GPU
        gbs = alg_bytes / (kern_ms * 1e-3) / 1e9
        traffic = measured_traffic()
        out = {
            "metric": "RK4 field-point updates/sec (sweep_pts x n_fields x n_zsteps / wall_s)",
            "value": value, "unit": "field-point updates/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": wall / args.steps * 1e3, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "BASELINE configs[1]: 65536 dbeta sweep points x 4 fields x 100000 z-steps, float64, "
                                   "per GPU (C2 inputs of SURVEY 8d)", "sweep_pts_per_gpu": PTS_PER_GPU,
                       "sweep_pts_total": n_global, "n_fields": N_FIELDS, "n_zsteps": N_ZSTEPS, "save_every": SAVE_EVERY,
                       "check_nan": True, "parallelism": f"sweep sharded x{world}, one RCCL all_gather per pass"
                       if world > 1 else "single GPU"},
            "rk4_steps_per_s": value / N_FIELDS,
            "roofline": {
                "kernel": "psa::rk4_sweep_kernel<double, 4, CHECK_BLOCK, false, 256>",
                "bound": "mfma",   # the contract's label for the COMPUTE roofline (enum hbm | mfma); see bound_detail
                "bound_detail": "compute-bound on the FP64 VECTOR ALU: the kernel issues 0 MFMA instructions (elementwise "
                                "complex recurrence, nothing to contract); MI355X FP64 vector and FP64 matrix dense peaks "
                                "are both 78.6 TFLOP/s, so the peak is the same number either way",
                "achieved": tflops, "peak": PEAK_FP64_VALU_TFLOPS, "unit": "TFLOP/s", "frac": tflops / PEAK_FP64_VALU_TFLOPS,
                "flops_per_launch": FLOPS_PER_RK4_STEP * rk4_steps_per_launch,
                "kernel_ms_avg": kern_ms,
                # 301.8 FP64 wave-instructions per z-step per wave (SQ_INSTS_VALU, profiles/README.md) x 4 issue cycles:
                # the clock the chip would need if the FP64 pipe never idled = a LOWER bound on the clock it held.  Boxes of
                # the pool differ by ~10 % here (DVFS / silicon), which moves `frac` with no change in the code.
                "fp64_issue_ghz_equiv": 301.8 * 4 * N_ZSTEPS / (kern_ms * 1e-3) / 1e9,
                "traffic": None if traffic is None else traffic.get("bytes_per_launch"),
                "traffic_source": None if traffic is None else traffic.get("source"),
                "note": "elementwise complex recurrence: no MFMA, ~2e-4 B per update -> FP64 vector issue is the binding "
                        "roofline (DESIGN.md section 5); the HBM view of the same launch follows",
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
        sys.stdout.flush()
        os.dup2(saved_stdout_fd, 1)
        print(json.dumps(out), flush=True)
        os.dup2(2, 1)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
