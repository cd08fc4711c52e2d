"""The N > 1 path on CPU: two and four processes, ``gloo`` backend, rendezvous on 127.0.0.1.

The shard / pack / all_gather / unpack logic is what is under test; the local executor is injected (the CPU
oracle), because the product executor is the HIP kernel and needs a GPU.  The result gathered by every rank must
equal the unsharded run bit for bit, for even and ragged splits, per-point and broadcast arguments, and it must
carry a failing point's first_bad_step through the record unchanged -- for float64, float32 and six-wave records.
"""
import os
import socket
import sys

import numpy as np
import pytest
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _oracle_executor(dbeta, *, n_steps, z_max, save_every, gamma, alpha, a0, check_nan, dbeta2=None):
    import oracle as O
    return O.sweep(dbeta, z_max=z_max, n=n_steps, save_every=save_every, check_nan=check_nan, gamma=gamma,
                   alpha=alpha, a0=a0, dbeta2=dbeta2, threads=1)


def _oracle_executor_f32(dbeta, *, dtype=None, **kw):
    """A stand-in for the float32 kernel: the oracle's result rounded to float32 (what travels is float32 either way)."""
    assert np.dtype(dtype) == np.float32
    r = _oracle_executor(np.asarray(dbeta, dtype=np.float64), **kw)
    return dict(a_end=r["a_end"].astype(np.complex64), p_end=r["p_end"].astype(np.float32),
                p_max=r["p_max"].astype(np.float32), first_bad_step=r["first_bad_step"])


def _worker_variants(rank, world, port, n_points, out_dir):
    """float32 records and six-wave records through the same shard/gather path (BASELINE configs 4 and 5)."""
    for p in (ROOT, os.path.join(ROOT, "oracle")):
        sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from psa_amd.distributed import sweep_sharded
        rng = np.random.default_rng(23)
        db = rng.uniform(-0.05, 0.05, n_points)
        db2 = rng.uniform(-0.05, 0.05, n_points)
        gam = rng.uniform(5e-3, 2e-2, n_points)
        gam[n_points // 3] = 40.0
        a4 = np.sqrt(np.array([0.5, 0.4, 1e-5, 2e-5])).astype(complex)
        a6 = np.sqrt(rng.uniform(1e-6, 1.0, (n_points, 6))) * np.exp(1j * rng.uniform(-3, 3, (n_points, 6)))
        r32 = sweep_sharded(db, n_steps=200, z_max=20.0, save_every=7, gamma=gam, alpha=1e-4, a0=a4, dtype=np.float32,
                            executor=_oracle_executor_f32)
        r6 = sweep_sharded(db, dbeta2=db2, n_steps=200, z_max=20.0, save_every=7, gamma=gam, alpha=1e-4, a0=a6,
                           executor=_oracle_executor)
        np.savez(os.path.join(out_dir, f"rank{rank}.npz"), a32=r32.a_end, pe32=r32.p_end, pm32=r32.p_max, bad32=r32.first_bad_step,
                 a6=r6.a_end, pm6=r6.p_max, bad6=r6.first_bad_step, db=db, db2=db2, gam=gam, a4=a4, a6_in=a6)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,n_points", [(2, 37), (2, 64), (3, 5), (8, 45)])
def test_float32_and_six_wave_records_through_gloo(tmp_path, oracle, world, n_points):
    mp.spawn(_worker_variants, args=(world, _free_port(), n_points, str(tmp_path)), nprocs=world, join=True)
    r0 = np.load(tmp_path / "rank0.npz")
    for r in range(1, world):
        rr = np.load(tmp_path / f"rank{r}.npz")
        for k in ("a32", "pe32", "pm32", "bad32", "a6", "pm6", "bad6"):
            assert np.array_equal(r0[k], rr[k], equal_nan=True), (r, k)
    ref = oracle.sweep(r0["db"], z_max=20.0, n=200, save_every=7, gamma=r0["gam"], alpha=1e-4, a0=r0["a4"])
    assert r0["a32"].dtype == np.complex64 and r0["pm32"].dtype == np.float32 and r0["bad32"].dtype == np.int64
    assert np.array_equal(r0["a32"], ref["a_end"].astype(np.complex64), equal_nan=True)     # bit-equal to unsharded
    assert np.array_equal(r0["pe32"], ref["p_end"].astype(np.float32), equal_nan=True)
    assert np.array_equal(r0["pm32"], ref["p_max"].astype(np.float32), equal_nan=True)
    assert np.array_equal(r0["bad32"], ref["first_bad_step"])
    ref6 = oracle.sweep(r0["db"], dbeta2=r0["db2"], z_max=20.0, n=200, save_every=7, gamma=r0["gam"], alpha=1e-4, a0=r0["a6_in"])
    assert r0["a6"].shape == (n_points, 6)
    assert np.array_equal(r0["a6"], ref6["a_end"], equal_nan=True)
    assert np.array_equal(r0["pm6"], ref6["p_max"], equal_nan=True)
    assert np.array_equal(r0["bad6"], ref6["first_bad_step"])
    assert r0["bad6"][n_points // 3] >= 0


def _worker(rank, world, port, n_points, out_dir):
    for p in (ROOT, os.path.join(ROOT, "oracle")):
        sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from psa_amd.distributed import sweep_sharded
        rng = np.random.default_rng(11)
        db = rng.uniform(-0.05, 0.05, n_points)
        gam = rng.uniform(5e-3, 2e-2, n_points)
        gam[n_points // 2] = 40.0                      # past the RK4 stability edge: the record must carry the index
        a0 = np.sqrt(rng.uniform(1e-6, 1.0, (n_points, 4))) * np.exp(1j * rng.uniform(-3, 3, (n_points, 4)))
        res = sweep_sharded(db, n_steps=300, z_max=30.0, save_every=7, gamma=gam, alpha=1e-4, a0=a0,
                            executor=_oracle_executor)
        res_b = sweep_sharded(db, n_steps=50, z_max=5.0, save_every=10, gamma=0.0115, alpha=0.0, a0=a0[0],
                              executor=_oracle_executor)
        np.savez(os.path.join(out_dir, f"rank{rank}.npz"), a_end=res.a_end, p_end=res.p_end, p_max=res.p_max,
                 bad=res.first_bad_step, b_a_end=res_b.a_end, db=db, gam=gam, a0=a0)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,n_points", [(2, 64), (2, 37), (2, 1), (4, 37), (4, 3), (8, 67)])
def test_multi_rank_gloo_sweep_equals_unsharded(tmp_path, oracle, world, n_points):
    mp.spawn(_worker, args=(world, _free_port(), n_points, str(tmp_path)), nprocs=world, join=True)
    r0 = np.load(tmp_path / "rank0.npz")
    for r in range(1, world):
        rr = np.load(tmp_path / f"rank{r}.npz")
        for k in ("a_end", "p_end", "p_max", "bad", "b_a_end"):
            assert np.array_equal(r0[k], rr[k], equal_nan=True), (r, k)   # every rank holds the full result
    ref = oracle.sweep(r0["db"], z_max=30.0, n=300, save_every=7, gamma=r0["gam"], alpha=1e-4, a0=r0["a0"])
    assert np.array_equal(r0["a_end"], ref["a_end"], equal_nan=True)
    assert np.array_equal(r0["p_max"], ref["p_max"], equal_nan=True)
    assert np.array_equal(r0["bad"], ref["first_bad_step"]) and r0["bad"].dtype == np.int64
    if n_points > 1:
        assert r0["bad"][n_points // 2] >= 0 and (np.delete(r0["bad"], n_points // 2) == -1).all()
    ref_b = oracle.sweep(r0["db"], z_max=5.0, n=50, save_every=10, gamma=0.0115, alpha=0.0, a0=r0["a0"][0])
    assert np.array_equal(r0["b_a_end"], ref_b["a_end"])


def test_shard_bounds_partition():
    from psa_amd.distributed import shard_bounds
    for n in (0, 1, 7, 8, 65536, 1_048_577):
        for w in (1, 2, 3, 8):
            cuts = [shard_bounds(n, w, r) for r in range(w)]
            assert cuts[0][0] == 0 and cuts[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(cuts, cuts[1:]))
            sizes = [hi - lo for lo, hi in cuts]
            assert max(sizes) - min(sizes) <= 1
    assert shard_bounds(1 << 20, 8, 3) == (3 * 131072, 4 * 131072)          # config 4: 131 072 points per GPU
    with pytest.raises(ValueError):
        shard_bounds(10, 2, 2)


@pytest.mark.parametrize("n_waves,dtype,bytes_per_point", [(4, np.float64, 88), (4, np.float32, 48), (6, np.float64, 120),
                                                           (6, np.float32, 64)])
def test_record_roundtrip_keeps_int64_bits(n_waves, dtype, bytes_per_point):
    from psa_amd.distributed import RecordLayout, unpack_gathered, shard_bounds
    lay = RecordLayout(n_waves, dtype)
    assert lay.bytes_per_point() == bytes_per_point
    rng = np.random.default_rng(0)
    a = (rng.normal(size=(5, n_waves)) + 1j * rng.normal(size=(5, n_waves))).astype(lay.cdtype)
    pe, pm = rng.normal(size=5).astype(dtype), rng.normal(size=5).astype(dtype)
    bad = np.array([-1, 0, 2**40 + 3, -1, 7], np.int64)
    words = lay.pack(a, pe, pm, bad, pad_to=8)
    assert words.dtype == np.int64 and words.nbytes == bytes_per_point * 8
    a2, pe2, pm2, b2 = lay.unpack(words, 5)
    assert np.array_equal(a2, a) and np.array_equal(pe2, pe) and np.array_equal(pm2, pm) and np.array_equal(b2, bad)
    # a ragged 3-way split of 5 points (2 + 2 + 1), every image padded to the widest block
    parts = [lay.pack(a[lo:hi], pe[lo:hi], pm[lo:hi], bad[lo:hi], pad_to=2) for lo, hi in (shard_bounds(5, 3, r) for r in range(3))]
    assert len({p.size for p in parts}) == 1
    a3, pe3, pm3, b3 = unpack_gathered(lay, np.stack(parts), 5, 3)
    assert np.array_equal(a3, a) and np.array_equal(pm3, pm) and np.array_equal(b3, bad)


def test_record_layout_properties_hold_for_arbitrary_splits():
    """Property test (hypothesis): for any point count, world size, wave count and dtype, packing every rank's block with
    zero padding to the widest block and unpacking the stacked images returns the original arrays bit for bit, and every
    image has the same number of int64 words."""
    from hypothesis import given, settings, strategies as st
    from psa_amd.distributed import RecordLayout, shard_bounds, unpack_gathered

    @settings(max_examples=60, deadline=None, derandomize=True)
    @given(n=st.integers(0, 70), world=st.integers(1, 9), nw=st.sampled_from([4, 6]), f32=st.booleans(), seed=st.integers(0, 2**31))
    def check(n, world, nw, f32, seed):
        lay = RecordLayout(nw, np.float32 if f32 else np.float64)
        rng = np.random.default_rng(seed)
        a = (rng.normal(size=(n, nw)) + 1j * rng.normal(size=(n, nw))).astype(lay.cdtype)
        pe, pm = rng.normal(size=n).astype(lay.dtype), rng.normal(size=n).astype(lay.dtype)
        bad = rng.integers(-1, 2**40, n)
        width = (n + world - 1) // world
        parts = []
        for r in range(world):
            lo, hi = shard_bounds(n, world, r)
            assert 0 <= hi - lo <= width
            parts.append(lay.pack(a[lo:hi], pe[lo:hi], pm[lo:hi], bad[lo:hi], pad_to=width))
        assert len({p.size for p in parts}) == 1 and parts[0].size == lay.words(width)
        a2, pe2, pm2, b2 = unpack_gathered(lay, np.stack(parts), n, world)
        assert np.array_equal(a2, a) and np.array_equal(pe2, pe) and np.array_equal(pm2, pm) and np.array_equal(b2, bad)

    check()
