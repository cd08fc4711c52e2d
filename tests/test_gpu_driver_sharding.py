"""The sweep drivers sharded on hardware: three processes share the one GPU of the box (well inside its 6-process limit),
each integrates ITS block through libpsa_hip.so -- with its block's phase mismatch produced on the device where the call asks
for that -- `gloo` carries the one gathered image (RCCL needs a GPU per rank: the 8-GPU run is the driver's), and every rank
must return the reference's numbers (goldens G2, G3, G13).  Plus `devices=[...]` (threads of one process) and the
double-buffered host staging of `DeviceSweep`."""
import os
import socket
import sys

import numpy as np
import pytest

from conftest import ATOL_DB, RTOL_F64, rel_err

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")


def _disp(g):
    from psa_amd import dispersion
    return dispersion.dispersion_params_from_D_S(float(g["lambda_c"]), float(g["D"]), float(g["S"]), 0.0, D_units="ps/nm/km",
                                                 S_units="ps/nm^2/km", dSdlmbd_units="ps/nm^3/km",
                                                 omega_ref=float(g["omega_ref"]))


def _calls(devices=None):
    from psa_amd import config, dispersion, scan_mismtach
    from psa_amd.phase_matching import PhaseMatchingConfig
    g2, g3, g13 = (np.load(os.path.join(GOLDEN, n + ".npz")) for n in ("G2", "G3", "G13"))
    cfg = config.custom_simulation_config(z_max=500.0, dz=0.2)
    dv = g13["disp"]
    d13 = dispersion.DispersionParams(omega_ref=dv[0], beta2=dv[1], beta3=dv[2], beta4=dv[3])
    out = {}
    x, gain, db = scan_mismtach.plot_max_gain_and_dbeta_vs_lambda_signal(
        cfg=cfg, lambda_p1_m=float(g2["lambda_p1"]), lambda_p2_m=float(g2["lambda_p2"]), lambda_signal_m=g2["lambda3"],
        gamma=float(g2["gamma"]), alpha=float(g2["alpha"]), p_in=g2["p_in"], dispersion=_disp(g2), length_unit="m",
        gain_unit="dB", phase_in=np.zeros(4), show=False, show_progress=False, devices=devices)
    out.update(g2_x=x, g2_gain=gain, g2_dbeta=db)
    x, gain = scan_mismtach.plot_max_signal_gain_vs_lambda_signal(
        cfg=cfg, lambda_p1_m=float(g3["lambda_p1"]), lambda_p2_m=float(g3["lambda_p2"]), lambda_signal_m=g3["lambda3"],
        gamma=float(g3["gamma"]), alpha=float(g3["alpha"]), p_in=g3["p_in"], phase_in=np.zeros(4), dispersion=_disp(g3),
        phase_matching_cfg=PhaseMatchingConfig(), gain_unit="db", show=False, show_progress=False, devices=devices)
    out.update(g3_gain=gain)
    cfg_g = config.custom_simulation_config(z_max=250.0, dz=0.25, save_every=5)
    for producer in ("host", "device"):
        r = scan_mismtach.scan_gain_grid(cfg=cfg_g, lambda_p1_m=1550e-9, lambda_p2_m=g13["lambda2"],
                                         lambda_signal_m=g13["lambda3"], gamma=0.0115, alpha=1.0e-4, p_in=g13["grid_p_in"],
                                         dispersion=d13, dbeta_producer=producer, devices=devices)
        out.update({f"g13_{producer}_gain": r["gain"], f"g13_{producer}_dbeta": r["dbeta"]})
    six = scan_mismtach.scan_six_wave_grid(cfg=config.custom_simulation_config(z_max=40.0, dz=0.1), lambda_p1_m=1550e-9,
                                           lambda_p2_m=1558e-9, Omega1=np.linspace(2e12, 2.4e13, 5),
                                           Omega2=np.linspace(3e12, 2.0e13, 7), gamma=0.0115, alpha=1.15e-4,
                                           p_in=[0.3, 0.25, 1e-6, 1e-6, 2e-6, 5e-7], dispersion=d13, dbeta_producer="device",
                                           devices=devices)
    out.update(six_gain=six["gain"], six_a_end=six["a_end"])
    return out


def _check_against_the_reference(r):
    g2, g3, g13 = (np.load(os.path.join(GOLDEN, n + ".npz")) for n in ("G2", "G3", "G13"))
    assert np.array_equal(r["g2_x"], g2["x"]) and np.array_equal(r["g2_dbeta"], g2["dbeta"])
    np.testing.assert_allclose(r["g2_gain"], g2["gain_db"], rtol=RTOL_F64, atol=ATOL_DB)
    np.testing.assert_allclose(r["g3_gain"], g3["gain_db"], rtol=RTOL_F64, atol=ATOL_DB)
    for producer in ("host", "device"):
        assert np.max(np.abs(r[f"g13_{producer}_gain"] - g13["grid_gain_sym"])) < ATOL_DB
        assert np.all(np.abs(r[f"g13_{producer}_dbeta"] - g13["grid_dbeta_sym"]) <= np.spacing(np.abs(g13["grid_dbeta_sym"])))


def _worker(rank, world, port, out_dir):
    for p in (ROOT, os.path.join(ROOT, "oracle")):
        sys.path.insert(0, p)
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        np.savez(os.path.join(out_dir, f"r{rank}.npz"), **_calls())
    finally:
        dist.destroy_process_group()


def test_three_ranks_share_the_gpu_through_the_drivers(tmp_path):
    import torch.multiprocessing as mp
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    mp.spawn(_worker, args=(3, port, str(tmp_path)), nprocs=3, join=True)
    r = [np.load(tmp_path / f"r{k}.npz") for k in range(3)]
    for k in (1, 2):
        for key in r[0].files:
            assert np.array_equal(r[0][key], r[k][key], equal_nan=True), (k, key)
    _check_against_the_reference(r[0])
    whole = _calls()                                   # this process: no process group, one launch per driver
    for key in r[0].files:
        assert np.array_equal(r[0][key], whole[key], equal_nan=True), key     # small sweeps: the same kernel either way


def test_devices_list_through_the_drivers():
    """devices=[0, 0]: two host threads drive the same GPU through psa_rk4_sweep_f64 at once (the box has one GPU; on a node
    the list names distinct ones) -- blocks come back in order and equal the single-launch call."""
    many, one = _calls(devices=[0, 0]), _calls()
    _check_against_the_reference(many)
    for key in one:
        assert np.array_equal(many[key], one[key], equal_nan=True), key


def test_device_sweep_stages_its_outputs_to_pinned_host_memory_pass_after_pass(oracle):
    """DeviceSweep.stage_to_host: the copy of pass k runs on a second stream while pass k+1 integrates into the other
    record; what arrives on the host is pass k's record and gains, for every k (inputs change between passes)."""
    import torch
    from psa_amd.distributed import DeviceSweep
    n = 4099
    a0 = np.sqrt([0.5, 0.5, 1e-5, 1e-5]).astype(complex)
    ds = DeviceSweep(np.zeros(n), n_steps=600, z_max=60.0, save_every=10, gamma=0.0115, alpha=1.15e-4, a0=a0)
    rng = np.random.default_rng(3)
    dbs = [rng.uniform(-0.05, 0.05, n) for _ in range(5)]
    got = []
    for k, db in enumerate(dbs):
        ds.dbeta.copy_(torch.as_tensor(db).to(ds.device))
        ds.launch()
        ds.summarize(1e-5, mode="max", gain_db=True)
        ds.stage_to_host()
        if k >= 1:                                     # pass k-1's image is complete while pass k is still in flight
            prev = ds._last ^ 1
            ds._copied[prev].synchronize()
            got.append((ds._host[prev].numpy().copy(), ds._host_gain[prev][0].numpy().copy()))
    words, summ = ds.host_result()
    got.append((words.copy(), summ[0].copy()))
    torch.cuda.synchronize()
    for db, (w, gain) in zip(dbs, got):
        ref = oracle.sweep(db, z_max=60.0, n=600, save_every=10, gamma=0.0115, alpha=1.15e-4, a0=a0)
        a, pe, pm, fb = ds.layout.unpack(w, n)
        assert rel_err(a, ref["a_end"]) < RTOL_F64 and rel_err(pm, ref["p_max"]) < RTOL_F64 and (fb == -1).all()
        want = oracle.gain_from_summary(ref["p_max"], ref["first_bad_step"], 1e-5, "db")
        assert np.max(np.abs(gain - want)) < ATOL_DB
    assert np.array_equal(ds.result().a_end, ds.layout.unpack(got[-1][0], n)[0])
