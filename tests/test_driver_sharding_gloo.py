"""The sweep DRIVERS under a process group (CPU, ``gloo``, world 2 and 3): ``plot_max_gain_and_dbeta_vs_lambda_signal``,
``plot_max_signal_gain_vs_lambda_signal``, ``scan_gain_grid``, ``scan_dbeta_seeded_signal`` and ``scan_six_wave_grid`` split
their points over the ranks, exchange ONE gathered image and return the full arrays on every rank -- equal to the unsharded
call bit for bit, and to the reference's own numbers (goldens G2, G3, G13) within the tolerance of record.

What is under test is the drivers' shard / pack / gather / unpack logic and their NaN rules; the product executor is the HIP
kernel (no CPU fallback), so in these CPU-only workers the two native entry points the drivers reach are replaced by the
test-suite's oracle -- inside the worker process only (``_install_cpu_executor``)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")
RTOL_F64, ATOL_DB = 1e-9, 5e-9


def _free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _install_cpu_executor():
    """Test double for libpsa_hip's two host entry points, built on the oracle (tests only)."""
    import oracle as O
    import psa_amd._native as nat

    def sweep_host(dbeta, *, n_steps, z_max, save_every, gamma, alpha, a0, dbeta2=None, check_nan=True, exact_step=False,
                   want_traj=False, dtype=np.float64, device=0, extra_flags=0):
        r = O.sweep(np.asarray(dbeta, dtype=np.float64), z_max=z_max, n=n_steps, save_every=save_every, check_nan=check_nan,
                    gamma=gamma, alpha=alpha, a0=a0, dbeta2=dbeta2, threads=1)
        r.update(traj=None, elapsed_ms=1.0)
        return r

    def gain_summary_host(p_metric, first_bad_step, p0_sig, *, gain_db=True, device=0):
        g = O.gain_from_summary(p_metric, first_bad_step, p0_sig, "db" if gain_db else "linear")
        fin = np.isfinite(g)
        bi = int(np.nanargmax(g)) if fin.any() else -1
        return g, bi, (float(g[bi]) if bi >= 0 else float("nan")), int(fin.sum())

    nat.sweep_host, nat.gain_summary_host = sweep_host, gain_summary_host
    return sweep_host


def _disp(g):
    from psa_amd import dispersion
    return dispersion.dispersion_params_from_D_S(float(g["lambda_c"]), float(g["D"]), float(g["S"]), 0.0, D_units="ps/nm/km",
                                                 S_units="ps/nm^2/km", dSdlmbd_units="ps/nm^3/km",
                                                 omega_ref=float(g["omega_ref"]))


def _driver_calls():
    """name -> thunk returning a dict of arrays; the same calls are made without and with a process group."""
    from psa_amd import config, dispersion, scan_mismtach
    from psa_amd.phase_matching import PhaseMatchingConfig, PhaseMatchingMethod
    g2, g3, g13 = (np.load(os.path.join(GOLDEN, n + ".npz")) for n in ("G2", "G3", "G13"))
    cfg = config.custom_simulation_config(z_max=500.0, dz=0.2)
    calls = {}

    def c_g2():
        x, gain, db = scan_mismtach.plot_max_gain_and_dbeta_vs_lambda_signal(
            cfg=cfg, lambda_p1_m=float(g2["lambda_p1"]), lambda_p2_m=float(g2["lambda_p2"]), lambda_signal_m=g2["lambda3"],
            gamma=float(g2["gamma"]), alpha=float(g2["alpha"]), p_in=g2["p_in"], dispersion=_disp(g2), length_unit="m",
            gain_unit="dB", phase_in=np.zeros(4), show=False, show_progress=False)
        return dict(x=x, gain=gain, dbeta=db)

    def c_g3():
        lam3 = g3["lambda3"][::4]          # every 4th of the 100 points: 25 points, an odd count for 2 and 3 ranks
        x, gain = scan_mismtach.plot_max_signal_gain_vs_lambda_signal(
            cfg=cfg, lambda_p1_m=float(g3["lambda_p1"]), lambda_p2_m=float(g3["lambda_p2"]), lambda_signal_m=lam3,
            gamma=float(g3["gamma"]), alpha=float(g3["alpha"]), p_in=g3["p_in"], phase_in=np.zeros(4), dispersion=_disp(g3),
            phase_matching_cfg=PhaseMatchingConfig(), gain_unit="db", show=False, show_progress=False)
        return dict(x=x, gain=gain)

    dv = g13["disp"]
    d13 = dispersion.DispersionParams(omega_ref=dv[0], beta2=dv[1], beta3=dv[2], beta4=dv[3])
    cfg_g = config.custom_simulation_config(z_max=250.0, dz=0.25, save_every=5)

    def c_g13(pm):
        def run():
            out = scan_mismtach.scan_gain_grid(cfg=cfg_g, lambda_p1_m=1550e-9, lambda_p2_m=g13["lambda2"],
                                               lambda_signal_m=g13["lambda3"], gamma=0.0115, alpha=1.0e-4,
                                               p_in=g13["grid_p_in"], dispersion=d13, phase_matching_cfg=pm,
                                               dbeta_producer="host")
            return dict(gain=out["gain"], dbeta=out["dbeta"], a_end=out["result"].a_end, n_finite=np.array(out["n_finite"]),
                        best=np.array(out["best_index"]))
        return run

    def c_holes():
        """A lambda3 axis with impossible plans in it (idler frequency <= 0): those points never run, on whichever rank."""
        lam3 = np.concatenate([g2["lambda3"][:7], [0.4e-6, 0.39e-6], g2["lambda3"][7:12]])
        x, gain, db = scan_mismtach.plot_max_gain_and_dbeta_vs_lambda_signal(
            cfg=config.custom_simulation_config(z_max=100.0, dz=0.2), lambda_p1_m=float(g2["lambda_p1"]),
            lambda_p2_m=float(g2["lambda_p2"]), lambda_signal_m=lam3, gamma=float(g2["gamma"]), alpha=float(g2["alpha"]),
            p_in=g2["p_in"], dispersion=_disp(g2), show=False, show_progress=False)
        return dict(gain=gain, dbeta=db)

    def c_scan():
        rng = np.random.default_rng(5)
        db = np.linspace(-0.05, 0.05, 41)
        gam = rng.uniform(5e-3, 2e-2, 41)
        gam[13] = 60.0                                     # overflows: first_bad_step >= 0 -> NaN gain
        out = scan_mismtach.scan_dbeta_seeded_signal(cfg=config.custom_simulation_config(z_max=60.0, dz=0.1, save_every=7),
                                                     delta_beta=db, gamma=gam, alpha=1.15e-4, p_in=[0.5, 0.5, 1e-5, 1e-5],
                                                     gain_mode="end")
        r = out["result"]
        return dict(gain=out["gain"], a_end=r.a_end, p_end=r.p_end, bad=r.first_bad_step, best=np.array(out["best_index"]),
                    n_finite=np.array(out["n_finite"]))

    def c_six():
        out = scan_mismtach.scan_six_wave_grid(cfg=config.custom_simulation_config(z_max=40.0, dz=0.1),
                                               lambda_p1_m=1550e-9, lambda_p2_m=1558e-9, Omega1=np.linspace(2e12, 2.4e13, 5),
                                               Omega2=np.linspace(3e12, 2.0e13, 7), gamma=0.0115, alpha=1.15e-4,
                                               p_in=[0.3, 0.25, 1e-6, 1e-6, 2e-6, 5e-7], dispersion=d13, dbeta_producer="host")
        return dict(gain=out["gain"], a_end=out["a_end"], bad=out["first_bad_step"])

    def c_tiny():
        """Fewer points than ranks (world 3): one rank's block is empty -- it still takes part in the exchange."""
        x, gain, db = scan_mismtach.plot_max_gain_and_dbeta_vs_lambda_signal(
            cfg=config.custom_simulation_config(z_max=50.0, dz=0.2), lambda_p1_m=float(g2["lambda_p1"]),
            lambda_p2_m=float(g2["lambda_p2"]), lambda_signal_m=g2["lambda3"][13:15], gamma=float(g2["gamma"]),
            alpha=float(g2["alpha"]), p_in=g2["p_in"], dispersion=_disp(g2), show=False, show_progress=False)
        return dict(gain=gain, dbeta=db)

    calls.update(g2=c_g2, g3=c_g3, tiny=c_tiny, g13_sym=c_g13(None),
                 g13_gen=c_g13(PhaseMatchingConfig(method=PhaseMatchingMethod.GENERAL_TAYLOR, max_order=4)),
                 holes=c_holes, scan=c_scan, six=c_six)
    return calls


def _worker(rank, world, port, out_dir):
    for p in (ROOT, os.path.join(ROOT, "oracle")):
        sys.path.insert(0, p)
    import torch.distributed as dist
    import psa_amd._native as nat
    inner, sizes = _install_cpu_executor(), []

    def counting(dbeta, **kw):
        sizes.append(len(dbeta))
        return inner(dbeta, **kw)
    nat.sweep_host = counting
    calls = _driver_calls()
    whole = {k: f() for k, f in calls.items()} if rank == 0 else None       # no process group yet: the unsharded call
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        shard, per_call = {}, []
        for k, f in calls.items():
            sizes.clear()
            shard[k] = f()
            per_call.append(sum(sizes))          # points this rank integrated for this driver call (0: an empty block)
        flat = {f"{k}.{name}": v for k, d in shard.items() for name, v in d.items()}
        if whole is not None:
            flat.update({f"whole.{k}.{name}": v for k, d in whole.items() for name, v in d.items()})
        np.savez(os.path.join(out_dir, f"rank{rank}.npz"), **flat)
        np.save(os.path.join(out_dir, f"sizes{rank}.npy"), np.array(per_call))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_drivers_shard_over_the_process_group_and_every_rank_gets_the_whole_sweep(world, tmp_path):
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    ranks = [np.load(tmp_path / f"rank{r}.npz") for r in range(world)]
    r0 = ranks[0]
    # each rank integrated only its block: g2 30 points, g3 25, tiny 2, g13 36 (twice), holes 12 of 14, scan 41, six 35
    totals = [30, 25, 2, 36, 36, 12, 41, 35]
    per_rank = np.stack([np.load(tmp_path / f"sizes{r}.npy") for r in range(world)])
    assert per_rank.shape == (world, len(totals)) and list(per_rank.sum(axis=0)) == totals
    assert (per_rank.max(axis=0) - per_rank.min(axis=0) <= 2).all()
    if world == 3:
        assert sorted(per_rank[:, 2]) == [0, 1, 1]                  # "tiny": two points, three ranks
    keys = [k for k in r0.files if not k.startswith("whole.")]
    assert len(keys) >= 20
    for k in keys:
        for r in ranks[1:]:
            assert np.array_equal(r0[k], r[k], equal_nan=True), f"rank results differ in {k}"
        assert np.array_equal(r0[k], r0["whole." + k], equal_nan=True), f"sharded != unsharded in {k}"
    # ... and the numbers are the reference's (goldens generated from it)
    g2, g3, g13 = (np.load(os.path.join(GOLDEN, n + ".npz")) for n in ("G2", "G3", "G13"))
    assert np.array_equal(r0["g2.x"], g2["x"]) and np.array_equal(r0["g2.dbeta"], g2["dbeta"])
    np.testing.assert_allclose(r0["g2.gain"], g2["gain_db"], rtol=RTOL_F64, atol=ATOL_DB)
    np.testing.assert_allclose(r0["g3.gain"], g3["gain_db"][::4], rtol=RTOL_F64, atol=ATOL_DB)
    assert np.max(np.abs(r0["g13_sym.gain"] - g13["grid_gain_sym"])) < ATOL_DB
    assert np.max(np.abs(r0["g13_gen.gain"] - g13["grid_gain_gen"])) < ATOL_DB
    assert np.all(np.abs(r0["g13_sym.dbeta"] - g13["grid_dbeta_sym"]) <= np.spacing(np.abs(g13["grid_dbeta_sym"])))
    np.testing.assert_allclose(r0["g13_gen.dbeta"], g13["grid_dbeta_gen"], rtol=1e-14, atol=0)
    # NaN rules survive the exchange: impossible plans (never ran) and a blown-up point (ran, failed)
    assert np.isnan(r0["holes.gain"][7:9]).all() and np.isnan(r0["holes.dbeta"][7:9]).all()
    assert np.isfinite(np.delete(r0["holes.gain"], [7, 8])).all()
    assert np.isnan(r0["scan.gain"][13]) and r0["scan.bad"][13] >= 0 and int(r0["scan.n_finite"]) == 40
    assert r0["six.a_end"].shape == (5, 7, 6) and (r0["six.bad"] == -1).all()


def test_devices_list_splits_the_points_over_threads():
    """devices=[...] (a plain Python caller, no process group): contiguous blocks, one thread per device, results in order."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import psa_amd._native as nat
    from psa_amd import sweep
    seen = []
    saved = (nat.sweep_host, nat.gain_summary_host)
    _install_cpu_executor()
    inner = nat.sweep_host

    def spy(dbeta, *, device=0, **kw):
        seen.append((int(device), len(dbeta)))
        return inner(dbeta, device=device, **kw)
    nat.sweep_host = spy
    try:
        rng = np.random.default_rng(2)
        db, gam = np.linspace(-0.04, 0.04, 11), rng.uniform(5e-3, 2e-2, 11)
        a0 = np.sqrt(rng.uniform(1e-5, 0.5, (11, 4))).astype(complex)
        kw = dict(z_max=30.0, n_steps=300, save_every=7, gamma=gam, alpha=1e-4, a0=a0)
        one = sweep.rk4_sweep(db, device=0, **kw)
        seen.clear()
        many = sweep.rk4_sweep(db, devices=[0, 1, 2], **kw)
        assert sorted(seen) == [(0, 4), (1, 4), (2, 3)]
        for f in ("a_end", "p_end", "p_max", "first_bad_step"):
            assert np.array_equal(getattr(one, f), getattr(many, f))
        with pytest.raises(ValueError):
            sweep.rk4_sweep(db, devices=[], **kw)
    finally:
        nat.sweep_host, nat.gain_summary_host = saved
