"""Result files: same on-disk format as the reference's io_fwm.py -- NPZ keys z / A / metadata_json,
CSV columns z, P_*, phi_*, JSON metadata -- plus the sweep-summary bundle.  CPU only."""
import csv
import json

import numpy as np
import pytest

import psa_amd
from psa_amd import config, io_fwm
from psa_amd.sweep import SweepResult


def _run():
    z = np.linspace(0.0, 1.0, 6)
    A = (np.arange(24).reshape(6, 4) + 1j * np.arange(24).reshape(6, 4)[::-1]) * 0.1
    return z, A


def test_npz_roundtrip_and_reference_key_names(tmp_path):
    z, A = _run()
    cfg = config.custom_simulation_config(z_max=1.0, dz=0.2)
    p = io_fwm.save_result_npz(tmp_path / "run", z, A, metadata={"cfg": cfg, "gamma": np.float64(0.0115), "n": np.int64(5)})
    assert p.suffix == ".npz"
    with np.load(p, allow_pickle=False) as raw:
        assert set(raw.files) == {"z", "A", "metadata_json"}          # the reference's three keys (io_fwm.py:127-132)
    z2, A2, md = io_fwm.load_result_npz(p)
    assert np.array_equal(z2, z) and np.array_equal(A2, A) and A2.dtype == np.complex128
    assert md["cfg"]["z_max"] == 1.0 and md["gamma"] == 0.0115 and md["n"] == 5 and md["timestamp_utc"].endswith("Z")
    with pytest.raises(FileExistsError):
        io_fwm.save_result_npz(p, z, A)
    io_fwm.save_result_npz(p, z, A, overwrite=True)
    for bad in ((z[None], A), (z, A[0]), (z[:-1], A)):
        with pytest.raises(ValueError):
            io_fwm.save_result_npz(tmp_path / "bad", *bad)
    with pytest.raises(FileNotFoundError):
        io_fwm.load_result_npz(tmp_path / "missing.npz")


def test_csv_summary_and_bundle(tmp_path):
    z, A = _run()
    out = io_fwm.save_run_bundle(tmp_path / "b", "r1", z, A, metadata={"note": "x"})
    assert sorted(out) == ["csv", "json", "npz"] and all(p.exists() for p in out.values())
    rows = list(csv.reader(out["csv"].open()))
    assert rows[0] == ["z", "P_pump 1", "P_pump 2", "P_signal", "P_idler", "phi_pump 1", "phi_pump 2", "phi_signal", "phi_idler"]
    assert len(rows) == 7
    np.testing.assert_allclose([float(v) for v in rows[3][1:5]], np.abs(A[2]) ** 2)
    np.testing.assert_allclose([float(v) for v in rows[3][5:]], np.angle(A[2]))
    assert io_fwm.load_metadata_json(out["json"])["note"] == "x"
    assert json.loads(out["json"].read_text())["timestamp_utc"] == io_fwm.load_result_npz(out["npz"])[2]["timestamp_utc"]
    with pytest.raises(ValueError):
        io_fwm.save_summary_csv(tmp_path / "c", z, A[:, :3])


def test_sweep_summary_roundtrip(tmp_path):
    rng = np.random.default_rng(0)
    res = SweepResult(rng.normal(size=(7, 4)) + 1j * rng.normal(size=(7, 4)), rng.random(7), rng.random(7),
                      np.array([-1, -1, 3, -1, -1, -1, 12], np.int64), 1000, 10, 1.25)
    db = np.linspace(-1, 1, 7)
    p = io_fwm.save_sweep_npz(tmp_path / "sw", res, dbeta=db, gain=rng.random(7), metadata={"gamma": 0.0115})
    got, extra, md = io_fwm.load_sweep_npz(p)
    assert np.array_equal(got.a_end, res.a_end) and np.array_equal(got.first_bad_step, res.first_bad_step)
    assert got.n_steps == 1000 and got.save_every == 10 and got.elapsed_ms == 1.25
    assert np.array_equal(extra["dbeta"], db) and "x" not in extra and md["n_points"] == 7 and md["gamma"] == 0.0115
    with pytest.raises(ValueError):
        io_fwm.save_sweep_npz(tmp_path / "sw2", res, dbeta=db[:3])
    with pytest.raises(ValueError):
        io_fwm.load_sweep_npz(io_fwm.save_result_npz(tmp_path / "notasweep", *_run()))


def test_files_written_by_the_reference_load_here_and_match_ours(tmp_path):
    """tests/golden/G12_io_ref/ was written by the reference's io_fwm.save_run_bundle (gen_golden.py G12)."""
    import os
    ref_dir = os.path.join(os.path.dirname(__file__), "golden", "G12_io_ref")
    z, A = _run()
    z2, A2, md = io_fwm.load_result_npz(os.path.join(ref_dir, "run.npz"))
    assert np.array_equal(z2, z) and np.array_equal(A2, A) and md["note"] == "written by the reference"
    assert io_fwm.load_metadata_json(os.path.join(ref_dir, "run.json"))["gamma"] == 0.0115
    ours = io_fwm.save_run_bundle(tmp_path, "run", z, A, metadata={"gamma": 0.0115, "note": "written by the reference",
                                                                   "timestamp_utc": "2026-01-01T00:00:00Z"})
    assert ours["csv"].read_text() == open(os.path.join(ref_dir, "run.csv")).read()          # byte-identical CSV
    assert json.loads(ours["json"].read_text()) == json.load(open(os.path.join(ref_dir, "run.json")))
    with np.load(ours["npz"]) as a, np.load(os.path.join(ref_dir, "run.npz")) as b:
        assert set(a.files) == set(b.files) and str(a["metadata_json"]) == str(b["metadata_json"])
