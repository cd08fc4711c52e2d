"""Saved rows (integrate_fixed_step's strided save, integrators.py:137-140) from every lane layout: the packed float32
kernel (two adjacent points per lane: one 16-B store per wave and lane, 8-B loads / stores of the summary rows) and the
two-lane float64 kernel (SGPR-base streaming stores, a dedicated save_every = 1 loop), against the one-point-per-lane
kernels and the oracle -- full waves, the odd tail, a partly filled wave, 4 and 6 waves, failures inside the saved range."""
import numpy as np
import pytest

import psa_amd._native as nat
from conftest import RTOL_F32, RTOL_F64, rel_err

pytestmark = pytest.mark.gpu


def _inputs(N, nw, seed):
    rng = np.random.default_rng(seed)
    db, db2 = rng.uniform(-0.05, 0.05, N), rng.uniform(-0.05, 0.05, N)
    gam, al = rng.uniform(0.008, 0.014, N), rng.uniform(0.0, 2e-4, N)
    lo = [0.3, 0.3] + [1e-6] * (nw - 2)
    hi = [0.6, 0.6] + [1e-4] * (nw - 2)
    a0 = np.sqrt(rng.uniform(lo, hi, (N, nw))) * np.exp(1j * rng.uniform(-3, 3, (N, nw)))
    return db, (db2 if nw == 6 else None), gam, al, a0


@pytest.mark.parametrize("nw", [4, 6])
@pytest.mark.parametrize("N", [1, 2, 7, 127, 128, 129, 1001])
@pytest.mark.parametrize("n,se", [(96, 1), (101, 1), (200, 7)])
def test_float32_packed_trajectories(oracle, nw, N, n, se):
    """N = 128 is one full packed wave (16-B stores), 127 / 129 / 1001 add the odd tail and a partly filled wave, 1 and 2
    are a single lane; save_every = 1 takes the dedicated loop (96 = one re-seed window + ..., 101 = odd step count)."""
    db, db2, gam, al, a0 = _inputs(N, nw, 1000 * nw + N)
    kw = dict(n_steps=n, z_max=0.1 * n, save_every=se, gamma=gam, alpha=al, a0=a0, dbeta2=db2, dtype=np.float32, want_traj=True)
    ref = oracle.sweep(db.astype(np.float32).astype(float), z_max=0.1 * n, n=n, save_every=se, gamma=gam.astype(np.float32).astype(float),
                       alpha=al.astype(np.float32).astype(float), a0=a0.astype(np.complex64).astype(complex),
                       dbeta2=None if db2 is None else db2.astype(np.float32).astype(float))
    sc = nat.sweep_host(db, extra_flags=nat.OPT_F32_SCALAR, **kw)
    for check in (dict(check_nan=True, exact_step=False), dict(check_nan=True, exact_step=True), dict(check_nan=False)):
        pk = nat.sweep_host(db, extra_flags=nat.OPT_F32_PACKED, **kw, **check)
        assert pk["traj"].shape == (N, n // se + 1, nw) and pk["traj"].dtype == np.complex64
        scale = np.abs(ref["a_end"]).max(axis=1, keepdims=True)
        assert np.max(np.abs(pk["a_end"].astype(complex) - ref["a_end"]) / scale) < RTOL_F32
        assert np.array_equal(pk["traj"][:, 0, :], a0.astype(np.complex64))
        assert np.array_equal(pk["traj"][:, -1, :], pk["a_end"])                      # A[-1] is the last saved row
        p_rows = np.abs(pk["traj"][:, :, 2].astype(complex)) ** 2
        assert rel_err(pk["p_max"].astype(float), p_rows.max(axis=1)) < 1e-6
        assert rel_err(pk["p_end"].astype(float), p_rows[:, -1]) < 1e-6
        assert (pk["first_bad_step"] == -1).all()
        tscale = np.abs(sc["traj"].astype(complex)).max(axis=2, keepdims=True)
        assert np.max(np.abs(pk["traj"].astype(complex) - sc["traj"].astype(complex)) / tscale) < 2e-5


@pytest.mark.parametrize("nw", [4, 6])
@pytest.mark.parametrize("N", [1, 5, 32, 33, 1000])
@pytest.mark.parametrize("n,se", [(64, 1), (131, 1), (200, 7)])
def test_two_lane_float64_trajectories(oracle, nw, N, n, se):
    """Two lanes per point: rows of both lanes' waves land where the one-lane kernel puts them, for every check mode and for
    64- and 256-thread workgroups; N = 32 is one full wave of pairs, 33 starts a second one."""
    db, db2, gam, al, a0 = _inputs(N, nw, 77 * nw + N)
    kw = dict(n_steps=n, z_max=0.1 * n, save_every=se, gamma=gam, alpha=al, a0=a0, dbeta2=db2, want_traj=True)
    ref = oracle.sweep(db, z_max=0.1 * n, n=n, save_every=se, gamma=gam, alpha=al, a0=a0, dbeta2=db2)
    one = nat.sweep_host(db, extra_flags=nat.OPT_ONE_LANE, exact_step=True, **kw)
    for flags in (nat.OPT_SPLIT_POINT, nat.OPT_SPLIT_POINT | nat.OPT_BLOCK64):
        for check in (dict(check_nan=True, exact_step=False), dict(check_nan=True, exact_step=True), dict(check_nan=False)):
            two = nat.sweep_host(db, extra_flags=flags, **kw, **check)
            assert rel_err(two["a_end"], ref["a_end"]) < RTOL_F64 and rel_err(two["p_max"], ref["p_max"]) < RTOL_F64
            assert rel_err(two["traj"], one["traj"]) < 1e-10
            assert np.array_equal(two["traj"][:, -1, :], two["a_end"]) and np.array_equal(two["traj"][:, 0, :], a0)
            assert (two["first_bad_step"] == -1).all()
    for i in (0, N // 2, N - 1):
        z, A, _ = oracle.integrate(a0[i], z_max=0.1 * n, n=n, save_every=se, gamma=gam[i], alpha=al[i], dbeta=db[i],
                                   **({"dbeta2": db2[i]} if nw == 6 else {}))
        assert rel_err(two["traj"][i], A) < RTOL_F64


@pytest.mark.parametrize("layout", ["f64_one", "f64_two", "f64_four", "f32_packed", "f32_scalar"])
def test_failure_inside_a_save_every_1_trajectory(layout, oracle):
    """A point that blows up while every step is saved: first_bad_step is the reference's index in exact mode (and the same
    in block mode, where a block is one step), rows after it are non-finite, p_max is NaN -- neighbours (the other half of a
    packed lane, the other lanes) are untouched."""
    N, n = 9, 40
    db = np.linspace(-0.04, 0.04, N)
    gam = np.full(N, 0.0115)
    gam[4] = 3.0e4                                  # overflows within a few steps in either precision
    a0 = np.sqrt([0.5, 0.5, 1e-5, 1e-5]).astype(complex)
    f32 = layout.startswith("f32")
    flags = {"f64_one": nat.OPT_ONE_LANE, "f64_two": nat.OPT_SPLIT_POINT, "f64_four": nat.OPT_QUAD_POINT,
             "f32_packed": nat.OPT_F32_PACKED, "f32_scalar": nat.OPT_F32_SCALAR}[layout]
    ref = oracle.sweep(db, z_max=4.0, n=n, save_every=1, gamma=gam, alpha=1e-4, a0=a0)
    for exact in (True, False):
        got = nat.sweep_host(db, n_steps=n, z_max=4.0, save_every=1, gamma=gam, alpha=1e-4, a0=a0, want_traj=True,
                             dtype=np.float32 if f32 else np.float64, exact_step=exact, extra_flags=flags)
        bad = got["first_bad_step"]
        assert (np.delete(bad, 4) == -1).all() and bad[4] >= 0
        if not f32:
            assert bad[4] == ref["first_bad_step"][4]
        assert np.isnan(got["p_max"][4]) and np.isfinite(np.delete(got["p_max"], 4)).all()
        assert not np.isfinite(got["traj"][4, bad[4] + 1]).all() and np.isfinite(got["traj"][4, :bad[4] + 1]).all()
        ok = np.delete(np.arange(N), 4)
        assert rel_err(got["a_end"][ok].astype(complex), ref["a_end"][ok]) < (RTOL_F32 if f32 else RTOL_F64)


def test_two_lane_trajectory_through_the_device_api_with_an_odd_point_count(oracle):
    """psa_rk4_sweep_f64_dev / _f32_dev with N odd: row bases of the SoA buffers are then only element-aligned (the packed
    kernel's 8-B and 16-B accesses must not assume more)."""
    import torch
    from psa_amd.distributed import DeviceSweep
    a0 = np.sqrt([0.5, 0.5, 1e-5, 1e-5]).astype(complex)
    for N in (129, 4097):
        db = np.linspace(-0.05, 0.05, N)
        for dtype, flags, tol in ((np.float32, nat.OPT_F32_PACKED, RTOL_F32), (np.float64, nat.OPT_SPLIT_POINT, RTOL_F64)):
            ds = DeviceSweep(db, n_steps=50, z_max=5.0, save_every=1, gamma=0.0115, alpha=1.15e-4, a0=a0, dtype=dtype,
                             extra_flags=flags)
            ds.enable_trajectory()
            ds.launch()
            torch.cuda.synchronize()
            tr = ds.traj.cpu().numpy().astype(float)                       # [rows][4][N][2]
            res = ds.result()
            for i in (0, 1, N // 2, N - 2, N - 1):
                z, A, _ = oracle.integrate(a0, z_max=5.0, n=50, save_every=1, gamma=0.0115, alpha=1.15e-4,
                                           dbeta=float(np.asarray(db[i], dtype=dtype)))
                assert rel_err(tr[:, :, i, 0] + 1j * tr[:, :, i, 1], A) < tol, (N, dtype, i)
            last = tr[-1, :, :, 0] + 1j * tr[-1, :, :, 1]
            assert np.array_equal(last.T.astype(res.a_end.dtype), res.a_end)


def test_padded_trajectory_leading_dimension(oracle):
    """Sizes whose (row, wave) regions would lie a multiple of 2 MiB apart get a device buffer with ld = N + 4 352 B worth of
    points (psa_traj_ld) -- inside the host-buffer API (the caller's array stays dense) and, with
    PSA_OPT_TRAJ_LD, in the device API.  131 072 float64 points and 262 144 float32 points are such sizes: rows of points at
    the ends, around a chunk boundary of the host transpose and in the middle against the oracle, in every lane layout."""
    import torch
    from psa_amd.distributed import DeviceSweep
    assert nat.traj_ld(131_072) == 131_072 + 272 and nat.traj_ld(262_144, np.float32) == 262_144 + 544
    assert nat.traj_ld(65_536) == 65_536 and nat.traj_ld(100_000) == 100_000 and nat.traj_ld(1) == 1 and nat.traj_ld(0) == 0
    assert nat.traj_ld(262_144) == 262_144 + 272 and nat.traj_ld(131_072, np.float32) == 131_072
    a0 = np.sqrt([0.5, 0.5, 1e-5, 1e-5]).astype(complex)
    n = 5
    for N, dtype, layouts, tol in ((131_072, np.float64, (nat.OPT_ONE_LANE, nat.OPT_SPLIT_POINT, nat.OPT_QUAD_POINT), RTOL_F64),
                                  (262_144, np.float32, (nat.OPT_F32_PACKED, nat.OPT_F32_SCALAR), RTOL_F32)):
        db = np.linspace(-0.05, 0.05, N).astype(dtype)
        pick = [0, 1, 63, 64, N // 2 - 1, N // 2, N - 2, N - 1]
        rows = {i: oracle.integrate(a0, z_max=0.5, n=n, save_every=1, gamma=0.0115, alpha=1.15e-4, dbeta=float(db[i]))[1] for i in pick}
        for flags in layouts:
            got = nat.sweep_host(db, n_steps=n, z_max=0.5, save_every=1, gamma=0.0115, alpha=1.15e-4, a0=a0, dtype=dtype,
                                 want_traj=True, extra_flags=flags)
            assert got["traj"].shape == (N, n + 1, 4) and np.array_equal(got["traj"][:, -1, :], got["a_end"])
            for i in pick:
                assert rel_err(got["traj"][i].astype(complex), rows[i]) < tol, (N, flags, i)
            ds = DeviceSweep(db, n_steps=n, z_max=0.5, save_every=1, gamma=0.0115, alpha=1.15e-4, a0=a0, dtype=dtype,
                             extra_flags=flags)
            ds.enable_trajectory()
            assert ds._traj_full.shape[2] == nat.traj_ld(N, dtype) > N and ds.traj.shape[2] == N and ds.flags & nat.OPT_TRAJ_LD
            ds.launch()
            torch.cuda.synchronize()
            tr = ds.traj[:, :, pick, :].cpu().numpy().astype(float)
            for k, i in enumerate(pick):
                assert rel_err(tr[:, :, k, 0] + 1j * tr[:, :, k, 1], rows[i]) < tol, (N, flags, i)
            assert np.array_equal(got["a_end"], ds.result().a_end)


@pytest.mark.parametrize("layout,nw", [(lay, nw) for lay in ("f64_one", "f64_two", "f32_packed", "f32_scalar") for nw in (4, 6)]
                         + [("f64_four", 4)])
def test_the_computed_trajectory_does_not_depend_on_save_every(layout, nw):
    """Upstream the stride only SELECTS rows (integrators.py:137-140): the arithmetic of step i is the same whatever is saved.
    Here the phase recurrence is re-seeded (and the float32 state folded) on the absolute grid i = 0, 64, 128, ... (16 in
    float32), not relative to saved rows, so the same holds bit for bit: row k of a run with stride s equals row k*s of the
    every-step run, A[-1] is that run's row (n // s) * s, for strides around the seed grid and beyond the run -- in every lane
    layout, and also for the summary-only kernels (no trajectory requested)."""
    f32 = layout.startswith("f32")
    flags = {"f64_one": nat.OPT_ONE_LANE, "f64_two": nat.OPT_SPLIT_POINT, "f64_four": nat.OPT_QUAD_POINT,
             "f32_packed": nat.OPT_F32_PACKED, "f32_scalar": nat.OPT_F32_SCALAR}[layout]
    N, n = 131, 333
    db, db2, gam, al, a0 = _inputs(N, nw, 5 * nw + len(layout))
    kw = dict(n_steps=n, z_max=33.3, gamma=gam, alpha=al, a0=a0, dbeta2=db2, dtype=np.float32 if f32 else np.float64,
              extra_flags=flags)
    every = nat.sweep_host(db, save_every=1, want_traj=True, **kw)
    assert every["traj"].shape == (N, n + 1, nw) and (every["first_bad_step"] == -1).all()
    for se in (2, 3, 7, 10, 16, 31, 64, 65, 100, 333, 1000):
        got = nat.sweep_host(db, save_every=se, want_traj=True, **kw)
        assert np.array_equal(got["traj"], every["traj"][:, ::se, :][:, :n // se + 1, :]), (layout, nw, se)
        assert np.array_equal(got["a_end"], every["traj"][:, n // se * se, :]), (layout, nw, se)
        summ = nat.sweep_host(db, save_every=se, want_traj=False, **kw)             # the non-trajectory instantiation
        assert np.array_equal(summ["a_end"], got["a_end"]) and np.array_equal(summ["p_max"], got["p_max"]), (layout, nw, se)
        assert np.array_equal(summ["p_end"], got["p_end"])
