"""Where the build-defined six-wave model CAN be tied to the reference, it is: with one signal/idler pair
dark the six-wave system is the reference's four-wave one, so the six-wave kernels must reproduce golden G8 -- numbers the
REFERENCE produced (tests/golden/gen_golden.py) -- directly, not via our own four-wave kernel:

  * pair 2 dark: waves [p1, p2, s, i, 0, 0] at (dbeta, anything)  -> G8's A_end / p_end / p_max in columns 0-3;
  * pair 1 dark: waves [p1, p2, 0, 0, s, i] at (anything, dbeta)  -> G8's A_end in columns 0, 1, 4, 5 (the kernel's summary
    wave is signal 1, so its p_end / p_max are 0 there: |A_s2|^2 is read from A_end);
  * exchanging the pairs -- (s1, i1, dbeta_1) <-> (s2, i2, dbeta_2) -- permutes the output (to rounding: see the test).

Both float64 lane layouts and the packed / scalar float32 kernels.  Beyond these reductions the six-wave model stays
"parity unpinned" (there is no six-wave reference)."""
import numpy as np
import pytest

import psa_amd._native as nat
from conftest import RTOL_F32, RTOL_F64, rel_err

pytestmark = pytest.mark.gpu

F64_LAYOUTS = [("one_lane", nat.OPT_ONE_LANE), ("two_lanes", nat.OPT_SPLIT_POINT), ("two_lanes_block64", nat.OPT_SPLIT_POINT | nat.OPT_BLOCK64)]
F32_LAYOUTS = [("packed", nat.OPT_F32_PACKED), ("scalar", nat.OPT_F32_SCALAR)]


def _run6(a0_six, db1, db2, g, alpha, n, flags, dtype=np.float64, **kw):
    return nat.sweep_host(db1, dbeta2=db2, n_steps=n, z_max=float(g["z_max"]), save_every=int(g["save_every"]),
                          gamma=float(g["gamma"]), alpha=alpha, a0=a0_six, extra_flags=flags, dtype=dtype, **kw)


@pytest.mark.parametrize("name,flags", F64_LAYOUTS)
@pytest.mark.parametrize("case", ["n1e4_a0", "n1e4_a1", "n1e5_a1"])
def test_dark_pair_reproduces_the_reference_vectors_float64(golden, name, flags, case):
    g = golden("G8")
    db = g["dbeta33"] if case.startswith("n1e5") else g["dbeta257"]
    n = 100_000 if case.startswith("n1e5") else 10_000
    alpha = float(g["alphas"][int(case[-1])])
    a4 = np.sqrt(g["p_in"]).astype(complex)
    other = 0.37 * db[::-1] + 0.011                         # the dark pair's mismatch must not matter
    # pair 2 dark
    got = _run6(np.concatenate([a4, [0, 0]]), db, other, g, alpha, n, flags)
    assert rel_err(got["a_end"][:, :4], g[case + "_A_end"]) < RTOL_F64 and np.all(got["a_end"][:, 4:] == 0)
    assert rel_err(got["p_end"], g[case + "_p_end"]) < RTOL_F64 and rel_err(got["p_max"], g[case + "_p_max"]) < RTOL_F64
    assert (got["first_bad_step"] == -1).all()
    # pair 1 dark: the live pair sits in columns 4, 5 and is driven by dbeta_2
    got = _run6(np.concatenate([a4[:2], [0, 0], a4[2:]]), other, db, g, alpha, n, flags)
    assert rel_err(got["a_end"][:, [0, 1, 4, 5]], g[case + "_A_end"]) < RTOL_F64 and np.all(got["a_end"][:, 2:4] == 0)
    assert rel_err(np.abs(got["a_end"][:, 4]) ** 2, g[case + "_p_end"]) < RTOL_F64
    assert np.all(got["p_max"] == 0) and np.all(got["p_end"] == 0)        # the summary wave (signal 1) is dark


@pytest.mark.parametrize("name,flags", F32_LAYOUTS)
def test_dark_pair_reproduces_the_reference_vectors_float32(golden, name, flags):
    g = golden("G8")
    db, case = g["dbeta257"], "n1e4_a1"
    alpha = float(g["alphas"][1])
    a4 = np.sqrt(g["p_in"]).astype(complex)
    other = 0.37 * db[::-1] + 0.011
    ref = g[case + "_A_end"]
    scale = np.abs(ref).max(axis=1, keepdims=True)           # float32: error relative to the point's largest wave
    got = _run6(np.concatenate([a4, [0, 0]]), db, other, g, alpha, 10_000, flags, dtype=np.float32)
    assert np.max(np.abs(got["a_end"][:, :4].astype(complex) - ref) / scale) < RTOL_F32 and np.all(got["a_end"][:, 4:] == 0)
    assert rel_err(got["p_max"].astype(float), g[case + "_p_max"]) < 5 * RTOL_F32
    got = _run6(np.concatenate([a4[:2], [0, 0], a4[2:]]), other, db, g, alpha, 10_000, flags, dtype=np.float32)
    assert np.max(np.abs(got["a_end"][:, [0, 1, 4, 5]].astype(complex) - ref) / scale) < RTOL_F32
    assert np.all(got["a_end"][:, 2:4] == 0)


@pytest.mark.parametrize("name,flags,dtype", [(n, f, np.float64) for n, f in F64_LAYOUTS] + [(n, f, np.float32) for n, f in F32_LAYOUTS])
def test_exchanging_the_pairs_permutes_the_output(name, flags, dtype):
    """(s1, i1, dbeta_1) <-> (s2, i2, dbeta_2): columns 2, 3 <-> 4, 5 of A_end and of every saved row.  The equations are
    symmetric under the exchange; the kernels are symmetric to ROUNDING, not bit for bit: the pumps' driving term is one FMA
    chain E_1 q_1 + E_2 q_2 (pair 1's products rounded first) and the two-lane layout forms A_p1 A_p2 with the lane's own
    pump as the FMA's exact factor -- making either bitwise symmetric would cost two more instructions per RHS evaluation.
    So the bar is the rounding noise of 1 500 steps: 1e-11 (float64), 2e-5 (float32), relative to the point's largest wave."""
    rng = np.random.default_rng(11)
    N = 203
    db1, db2 = rng.uniform(-0.05, 0.05, N), rng.uniform(-0.05, 0.05, N)
    p = np.column_stack([rng.uniform(0.2, 0.6, (N, 2)), 10 ** rng.uniform(-6, -3, (N, 4))])
    a0 = np.sqrt(p) * np.exp(1j * rng.uniform(-3, 3, (N, 6)))
    swap = [0, 1, 4, 5, 2, 3]
    kw = dict(n_steps=1500, z_max=150.0, save_every=10, gamma=0.0115, alpha=1.15e-4, extra_flags=flags, dtype=dtype,
              want_traj=True)
    a = nat.sweep_host(db1, dbeta2=db2, a0=a0, **kw)
    b = nat.sweep_host(db2, dbeta2=db1, a0=a0[:, swap], **kw)
    tol = 1e-11 if dtype == np.float64 else 2e-5
    ta, tb = a["traj"].astype(complex), b["traj"].astype(complex)
    scale = np.abs(ta).max(axis=2, keepdims=True)
    assert np.max(np.abs(ta[:, :, swap] - tb) / scale) < tol, name
    assert np.array_equal(a["first_bad_step"], b["first_bad_step"])
    # the summary follows signal 1: after the exchange it reports what was signal 2
    assert rel_err(b["p_end"].astype(float), np.abs(a["a_end"][:, 4].astype(complex)) ** 2) < 10 * tol
    assert rel_err(b["p_max"].astype(float), (np.abs(ta[:, :, 4]) ** 2).max(axis=1)) < 10 * tol
