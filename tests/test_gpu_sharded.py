"""The HBM-resident shard object the multi-GPU path and bench.py are built on (``distributed.DeviceSweep``) for every
configuration BASELINE.json shards: float64 / float32, four / six waves, dbeta from the host or generated on the GPU,
device-side gain summary, and the RCCL leg (world_size 1 on this one-GPU box; world_size 2 and 3 run under gloo on
CPU in tests/test_distributed_gloo.py).  Checker: the oracle; tolerances as tests/test_gpu_parity.py.
"""
import socket

import numpy as np
import pytest

import psa_amd._native as nat
from conftest import ATOL_DB, RTOL_F32, RTOL_F64, rel_err

pytestmark = pytest.mark.gpu

P4 = np.array([0.5, 0.5, 1e-5, 1e-5])
P6 = np.array([0.3, 0.25, 1e-6, 1e-6, 2e-6, 5e-7])
A4, A6 = np.sqrt(P4).astype(complex), np.sqrt(P6).astype(complex)


@pytest.fixture(scope="module")
def rccl_one_rank():
    import torch
    import torch.distributed as dist
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1,
                            device_id=torch.device("cuda", 0))
    yield dist
    dist.destroy_process_group()


def test_device_gain_summary_and_gather_on_a_sweep_with_failing_points(oracle, rccl_one_rank):
    """scan_mismtach.py:376-392 on the device: per-point gain in dB, NaN for failed points, best index, n_finite -- for
    1 500 points of which every 97th blows up (gamma past the RK4 stability edge) -- then the record through RCCL."""
    import torch
    from psa_amd.distributed import DeviceSweep, unpack_gathered
    N = 1500
    rng = np.random.default_rng(8)
    db = rng.uniform(-0.05, 0.05, N)
    gam = np.full(N, 0.0115)
    gam[::97] = 60.0
    ds = DeviceSweep(db, n_steps=1000, z_max=100.0, save_every=10, gamma=gam, alpha=1.15e-4, a0=A4, pad_to=N + 3)
    assert not ds.flags & nat.BCAST_GAMMA and ds.flags & nat.BCAST_ALPHA and ds.flags & nat.BCAST_A0
    ref = oracle.sweep(db, z_max=100.0, n=1000, save_every=10, gamma=gam, alpha=1.15e-4, a0=A4)
    for mode, key in (("max", "p_max"), ("end", "p_end")):
        ds.launch()
        ds.summarize(float(P4[2]), mode=mode, gain_db=True)
        torch.cuda.synchronize()
        want = oracle.gain_from_summary(ref[key], ref["first_bad_step"], P4[2], "db")
        got = ds.gain.cpu().numpy()
        failed = ref["first_bad_step"] >= 0
        assert failed.sum() == len(range(0, N, 97)) and np.array_equal(np.isnan(got), failed)
        assert np.max(np.abs(got[~failed] - want[~failed])) < ATOL_DB
        best_i, n_fin = (int(v) for v in ds.best.cpu().numpy())
        assert n_fin == int((~failed).sum())
        assert best_i == int(np.nanargmax(want)) and abs(float(ds.best_gain.item()) - np.nanmax(want)) < ATOL_DB
    ds.summarize(float(P4[2]), mode="max", gain_db=False)            # linear gain
    torch.cuda.synchronize()
    lin = ds.gain.cpu().numpy()
    assert rel_err(lin[~failed], (ref["p_max"] / P4[2])[~failed]) < RTOL_F64
    g = ds.gather()
    torch.cuda.synchronize()
    assert g.shape == (1, ds.layout.words(N + 3))                     # padded to the widest block of the (virtual) split
    a, pe, pm, fb = unpack_gathered(ds.layout, g.cpu().numpy(), N, 1)
    res = ds.result()
    assert np.array_equal(a, res.a_end, equal_nan=True) and np.array_equal(fb, res.first_bad_step)
    assert np.array_equal(fb >= 0, failed)
    assert rel_err(a[~failed], ref["a_end"][~failed]) < RTOL_F64


@pytest.mark.parametrize("N", [4096, 4097])
def test_float32_shard_object(oracle, rccl_one_rank, N):
    """BASELINE config 4's arithmetic: the packed float32 kernel writes straight into the float32 record (48 B/point);
    gain summary through psa_gain_summary_f32_dev; odd N exercises the half-filled last lane."""
    import torch
    from psa_amd.distributed import DeviceSweep, unpack_gathered
    db = np.linspace(-0.02, 0.02, N)
    ds = DeviceSweep(db, n_steps=4000, z_max=400.0, save_every=10, gamma=0.0115, alpha=1.15e-4, a0=A4, dtype=np.float32)
    assert ds.record.numel() * 8 == 48 * N
    ds.launch()
    ds.summarize(float(P4[2]), mode="max", gain_db=True)
    g = ds.gather()
    torch.cuda.synchronize()
    ref = oracle.sweep(db.astype(np.float32).astype(np.float64), z_max=400.0, n=4000, save_every=10, gamma=0.0115,
                       alpha=1.15e-4, a0=A4)
    a, pe, pm, fb = unpack_gathered(ds.layout, g.cpu().numpy(), N, 1)
    assert a.dtype == np.complex64 and pm.dtype == np.float32 and (fb == -1).all()
    assert rel_err(a.astype(complex), ref["a_end"]) < RTOL_F32 and rel_err(pm.astype(float), ref["p_max"]) < RTOL_F32
    want = oracle.gain_from_summary(ref["p_max"], ref["first_bad_step"], P4[2], "db")
    got = ds.gain.cpu().numpy()
    assert got.dtype == np.float32 and np.max(np.abs(got - want)) < 5e-4
    best_i, n_fin = (int(v) for v in ds.best.cpu().numpy())
    assert n_fin == N and abs(want[best_i] - want.max()) < 5e-4


def test_six_wave_shard_object_with_device_generated_mismatch(oracle, rccl_one_rank):
    """BASELINE config 5's shape in small: a block of the (Omega1 x Omega2) grid, (dbeta_1, dbeta_2) produced on the GPU
    by psa_dbeta_pairs_f64_dev, six-wave float64 kernel, 120 B/point record."""
    import torch
    from psa_amd import dispersion, frequency_plan
    from psa_amd.distributed import DeviceSweep, shard_bounds
    d = dispersion.dispersion_params_from_D_S(1554e-9, 0.1, 0.02, 0.0, D_units="ps/nm/km", S_units="ps/nm^2/km",
                                              dSdlmbd_units="ps/nm^3/km")
    w1, w2 = frequency_plan.omega_from_lambda(1550e-9), frequency_plan.omega_from_lambda(1558e-9)
    wd = 0.5 * (w1 - w2)
    O1, O2 = np.linspace(2e12, 2.4e13, 24), np.linspace(3e12, 2.0e13, 31)
    r1 = dispersion.delta_beta_symmetric_array(wd, O1, d)
    r2 = dispersion.delta_beta_symmetric_array(wd, O2, d)
    R1, R2 = (m.ravel() for m in np.meshgrid(r1, r2, indexing="ij"))
    lo, hi = shard_bounds(O1.size * O2.size, 3, 1)                    # the middle one of three ragged blocks
    ds = DeviceSweep(n_local=hi - lo, n_steps=3000, z_max=300.0, save_every=10, gamma=0.0115, alpha=1.15e-4, a0=A6)
    ds.fill_dbeta_pairs(nat.dbeta_model(d, None, even_orders=(2, 4)), wd, O1, O2, first=lo)
    ds.launch()
    ds.summarize(float(P6[2]), mode="max", gain_db=True)
    torch.cuda.synchronize()
    assert ds.record.numel() * 8 == 120 * (hi - lo)
    np.testing.assert_allclose(ds.dbeta.cpu().numpy(), R1[lo:hi], rtol=3e-16)
    np.testing.assert_allclose(ds.dbeta2.cpu().numpy(), R2[lo:hi], rtol=3e-16)
    ref = oracle.sweep(R1[lo:hi], dbeta2=R2[lo:hi], z_max=300.0, n=3000, save_every=10, gamma=0.0115, alpha=1.15e-4, a0=A6)
    res = ds.result()
    assert res.a_end.shape == (hi - lo, 6)
    assert rel_err(res.a_end, ref["a_end"]) < RTOL_F64 and rel_err(res.p_max, ref["p_max"]) < RTOL_F64
    want = oracle.gain_from_summary(ref["p_max"], ref["first_bad_step"], P6[2], "db")
    assert np.max(np.abs(ds.gain.cpu().numpy() - want)) < ATOL_DB


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_wavelength_grid_generated_per_block_equals_the_grid_driver(oracle, dtype):
    """Config 3 / config 4 shape in small: every rank's block of the lambda_p2 x lambda_3 grid gets its dbeta from
    psa_dbeta_grid_*_dev; the three blocks together must equal scan_gain_grid's host-generated run (gain and NaN mask:
    the 0.7 um column has no idler, scan_mismtach.py:391-392)."""
    import torch
    from psa_amd import config, dispersion, scan_mismtach
    from psa_amd.distributed import DeviceSweep, shard_bounds
    from psa_amd.phase_matching import PhaseMatchingConfig
    lam2 = np.linspace(1552e-9, 1562e-9, 9)
    lam3 = np.concatenate([np.linspace(1540e-9, 1565e-9, 20), [0.7e-6]])
    d = dispersion.dispersion_params_from_D_S(1554e-9, 0.1, 0.02, 0.0, D_units="ps/nm/km", S_units="ps/nm^2/km",
                                              dSdlmbd_units="ps/nm^3/km")
    p_in = np.array([0.1, 0.1, 1e-7, 1e-7])
    cfg = config.custom_simulation_config(z_max=300.0, dz=0.1)
    host = scan_mismtach.scan_gain_grid(cfg=cfg, lambda_p1_m=1550e-9, lambda_p2_m=lam2, lambda_signal_m=lam3, gamma=0.0115,
                                        alpha=1.15e-4, p_in=p_in, dispersion=d)
    model = nat.dbeta_model(d, PhaseMatchingConfig())
    N = lam2.size * lam3.size
    gains = []
    for r in range(3):
        lo, hi = shard_bounds(N, 3, r)
        ds = DeviceSweep(n_local=hi - lo, n_steps=3000, z_max=300.0, save_every=cfg.save_every, gamma=0.0115, alpha=1.15e-4,
                         a0=np.sqrt(p_in).astype(complex), dtype=dtype)
        ds.fill_dbeta_grid(model, 1550e-9, lam2, lam3, first=lo)
        ds.launch()
        ds.summarize(float(p_in[2]), mode="max", gain_db=True)
        torch.cuda.synchronize()
        gains.append(ds.gain.cpu().numpy().astype(np.float64))
    gain = np.concatenate(gains).reshape(lam2.size, lam3.size)
    assert np.array_equal(np.isnan(gain), np.isnan(host["gain"])) and np.isnan(gain[:, -1]).all()
    live = ~np.isnan(gain)
    tol = ATOL_DB if dtype == np.float64 else 2e-3
    assert np.max(np.abs(gain[live] - host["gain"][live])) < tol


def test_sweep_sharded_float32_and_six_waves_through_rccl(oracle, rccl_one_rank):
    """distributed.sweep_sharded(dtype=float32) and the six-wave record over the nccl backend (one rank here)."""
    from psa_amd.distributed import sweep_sharded
    db = np.linspace(-0.03, 0.03, 501)
    r32 = sweep_sharded(db, n_steps=2000, z_max=200.0, save_every=10, gamma=0.0115, alpha=1.15e-4, a0=A4, dtype=np.float32)
    ref = oracle.sweep(db.astype(np.float32).astype(np.float64), z_max=200.0, n=2000, save_every=10, gamma=0.0115,
                       alpha=1.15e-4, a0=A4)
    assert r32.a_end.dtype == np.complex64 and rel_err(r32.a_end.astype(complex), ref["a_end"]) < RTOL_F32
    g32 = r32.gain(P4[2], mode="max", unit="dB")
    assert g32.dtype == np.float32
    r6 = sweep_sharded(db, dbeta2=-0.4 * db, n_steps=2000, z_max=200.0, save_every=10, gamma=0.0115, alpha=1.15e-4, a0=A6)
    ref6 = oracle.sweep(db, dbeta2=-0.4 * db, z_max=200.0, n=2000, save_every=10, gamma=0.0115, alpha=1.15e-4, a0=A6)
    assert r6.a_end.shape == (501, 6) and rel_err(r6.a_end, ref6["a_end"]) < RTOL_F64
    assert np.array_equal(r6.first_bad_step, ref6["first_bad_step"])


def _shard_object_worker(rank, world, port, out_dir):
    import os
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for q in (root, os.path.join(root, "oracle")):
        sys.path.insert(0, q)
    import numpy as np
    import torch
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import psa_amd._native as nat
        from psa_amd import dispersion
        from psa_amd.distributed import DeviceSweep, shard_bounds, unpack_gathered
        from psa_amd.phase_matching import PhaseMatchingConfig
        lam2 = np.linspace(1552e-9, 1562e-9, 11)
        lam3 = np.linspace(1540e-9, 1565e-9, 23)
        N = lam2.size * lam3.size                                       # 253 = 85 + 84 + 84
        d = dispersion.dispersion_params_from_D_S(1554e-9, 0.1, 0.02, 0.0, D_units="ps/nm/km", S_units="ps/nm^2/km",
                                                  dSdlmbd_units="ps/nm^3/km")
        lo, hi = shard_bounds(N, world, rank)
        out = {}
        for name, dtype in (("f64", np.float64), ("f32", np.float32)):
            ds = DeviceSweep(n_local=hi - lo, n_steps=2000, z_max=200.0, save_every=10, gamma=0.0115, alpha=1.15e-4,
                             a0=np.sqrt([0.1, 0.1, 1e-7, 1e-7]).astype(complex), dtype=dtype, pad_to=(N + world - 1) // world,
                             device=torch.device("cuda", 0))
            ds.fill_dbeta_grid(nat.dbeta_model(d, PhaseMatchingConfig()), 1550e-9, lam2, lam3, first=lo)
            ds.launch()
            g = ds.gather()                                             # gloo: staged through the host, same words
            torch.cuda.synchronize()
            a, pe, pm, fb = unpack_gathered(ds.layout, g.cpu().numpy(), N, world)
            db_all = [torch.zeros((N + world - 1) // world, dtype=ds.dbeta.dtype) for _ in range(world)]
            mine = torch.zeros((N + world - 1) // world, dtype=ds.dbeta.dtype)
            mine[:hi - lo] = ds.dbeta.cpu()
            dist.all_gather(db_all, mine)
            out.update({f"a_{name}": a, f"pm_{name}": pm, f"fb_{name}": fb,
                        f"db_{name}": np.concatenate([t.numpy()[:shard_bounds(N, world, r)[1] - shard_bounds(N, world, r)[0]]
                                                      for r, t in enumerate(db_all)])})
        np.savez(os.path.join(out_dir, f"r{rank}.npz"), **out)
    finally:
        dist.destroy_process_group()


def test_three_ranks_generate_their_blocks_and_gather_ragged_records(tmp_path, oracle):
    """The 8-GPU code path with three processes on this box's one GPU: every rank builds the shard object for its RAGGED
    block (85 / 84 / 84 points), generates the block's dbeta on the device from the grid definition, runs the kernel and
    gathers the zero-padded records (gloo carries the words; RCCL needs a GPU per rank).  Every rank must end up with the
    whole sweep, equal to the oracle on the host-generated grid -- float64 and float32 records."""
    import torch.multiprocessing as mp
    from psa_amd import dispersion, frequency_plan, phase_matching
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    mp.spawn(_shard_object_worker, args=(3, port, str(tmp_path)), nprocs=3, join=True)
    r = [np.load(tmp_path / f"r{k}.npz") for k in range(3)]
    for k in (1, 2):
        for key in r[0].files:
            assert np.array_equal(r[0][key], r[k][key], equal_nan=True), (k, key)
    lam2, lam3 = np.linspace(1552e-9, 1562e-9, 11), np.linspace(1540e-9, 1565e-9, 23)
    d = dispersion.dispersion_params_from_D_S(1554e-9, 0.1, 0.02, 0.0, D_units="ps/nm/km", S_units="ps/nm^2/km",
                                              dSdlmbd_units="ps/nm^3/km")
    L2, L3 = np.meshgrid(lam2, lam3, indexing="ij")
    om, ok = frequency_plan.plan_from_wavelengths_batch(1550e-9, L2.ravel(), L3.ravel())
    db, ok2 = phase_matching.compute_phase_mismatch_batch(om, d, phase_matching.PhaseMatchingConfig())
    assert ok.all() and ok2.all()
    np.testing.assert_allclose(r[0]["db_f64"], db, rtol=3e-16)
    assert np.array_equal(r[0]["db_f32"], r[0]["db_f64"].astype(np.float32))       # the float32 producer rounds the float64 value
    a0 = np.sqrt([0.1, 0.1, 1e-7, 1e-7]).astype(complex)
    ref = oracle.sweep(r[0]["db_f64"], z_max=200.0, n=2000, save_every=10, gamma=0.0115, alpha=1.15e-4, a0=a0)
    assert r[0]["a_f64"].shape == (253, 4) and rel_err(r[0]["a_f64"], ref["a_end"]) < RTOL_F64
    assert rel_err(r[0]["pm_f64"], ref["p_max"]) < RTOL_F64 and (r[0]["fb_f64"] == -1).all()
    ref32 = oracle.sweep(r[0]["db_f32"].astype(np.float64), z_max=200.0, n=2000, save_every=10, gamma=0.0115, alpha=1.15e-4, a0=a0)
    assert r[0]["a_f32"].dtype == np.complex64 and rel_err(r[0]["a_f32"].astype(complex), ref32["a_end"]) < RTOL_F32


def test_six_wave_float32_shard_with_device_generated_pairs(oracle):
    """The one combination the BASELINE configurations do not use -- six waves in float32 (packed kernel, 64 B/point record,
    psa_dbeta_pairs_f32_dev) -- through the same shard object, against the float64 oracle."""
    import torch
    from psa_amd import dispersion, frequency_plan
    from psa_amd.distributed import DeviceSweep
    d = dispersion.dispersion_params_from_D_S(1554e-9, 0.1, 0.02, 0.0, D_units="ps/nm/km", S_units="ps/nm^2/km",
                                              dSdlmbd_units="ps/nm^3/km")
    w1, w2 = frequency_plan.omega_from_lambda(1550e-9), frequency_plan.omega_from_lambda(1558e-9)
    wd = 0.5 * (w1 - w2)
    O1, O2 = np.linspace(2e12, 2.4e13, 13), np.linspace(3e12, 2.0e13, 17)
    n = O1.size * O2.size                                             # 221: odd, so the last packed lane is half filled
    ds = DeviceSweep(n_local=n, n_steps=3000, z_max=300.0, save_every=10, gamma=0.0115, alpha=1.15e-4, a0=A6, dtype=np.float32)
    ds.fill_dbeta_pairs(nat.dbeta_model(d, None, even_orders=(2, 4)), wd, O1, O2, first=0)
    ds.launch()
    ds.summarize(float(P6[2]), mode="max", gain_db=True)
    torch.cuda.synchronize()
    assert ds.record.numel() * 8 == 64 * n
    db1, db2 = ds.dbeta.cpu().numpy(), ds.dbeta2.cpu().numpy()
    r1 = dispersion.delta_beta_symmetric_array(wd, O1, d)
    r2 = dispersion.delta_beta_symmetric_array(wd, O2, d)
    R1, R2 = (m.ravel() for m in np.meshgrid(r1, r2, indexing="ij"))
    assert db1.dtype == np.float32 and np.array_equal(db1, R1.astype(np.float32)) and np.array_equal(db2, R2.astype(np.float32))
    ref = oracle.sweep(db1.astype(np.float64), dbeta2=db2.astype(np.float64), z_max=300.0, n=3000, save_every=10, gamma=0.0115,
                       alpha=1.15e-4, a0=A6)
    res = ds.result()
    assert res.a_end.dtype == np.complex64 and res.a_end.shape == (n, 6) and (res.first_bad_step == -1).all()
    scale = np.abs(ref["a_end"]).max(axis=1, keepdims=True)          # float32: error relative to the point's largest wave
    assert np.max(np.abs(res.a_end.astype(complex) - ref["a_end"]) / scale) < RTOL_F32
    want = oracle.gain_from_summary(ref["p_max"], ref["first_bad_step"], P6[2], "db")
    assert np.max(np.abs(ds.gain.cpu().numpy().astype(float) - want)) < 2e-3


def test_shard_object_with_per_point_gamma_alpha_and_amplitudes(oracle):
    """DeviceSweep takes per-point gamma / alpha arrays and a (n_local, n_waves) amplitude matrix as well as the broadcast
    forms; alpha == 0 as a scalar selects the lossless instantiation."""
    import torch
    from psa_amd.distributed import DeviceSweep
    rng = np.random.default_rng(17)
    n = 301
    db = rng.uniform(-0.05, 0.05, n)
    gam, al = rng.uniform(5e-3, 2e-2, n), rng.uniform(0, 3e-4, n)
    a0 = np.sqrt(rng.uniform(1e-6, 0.8, (n, 4))) * np.exp(1j * rng.uniform(-3, 3, (n, 4)))
    for kw in (dict(gamma=gam, alpha=al, a0=a0), dict(gamma=0.0115, alpha=al, a0=a0[5]), dict(gamma=gam, alpha=0.0, a0=a0)):
        ds = DeviceSweep(db, n_steps=1500, z_max=150.0, save_every=10, **kw)
        assert bool(ds.flags & nat.OPT_LOSSLESS) == (np.ndim(kw["alpha"]) == 0 and kw["alpha"] == 0.0)
        ds.launch()
        torch.cuda.synchronize()
        ref = oracle.sweep(db, z_max=150.0, n=1500, save_every=10, **kw)
        res = ds.result()
        assert rel_err(res.a_end, ref["a_end"]) < RTOL_F64 and rel_err(res.p_max, ref["p_max"]) < RTOL_F64
    with pytest.raises(ValueError):
        DeviceSweep(db, n_steps=10, z_max=1.0, save_every=1, gamma=gam[:5], alpha=0.0, a0=a0)
    with pytest.raises(ValueError):
        DeviceSweep(db, n_steps=10, z_max=1.0, save_every=1, gamma=0.01, alpha=0.0, a0=a0[:7])
