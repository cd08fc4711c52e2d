import os, sys, time
import numpy as np
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "oracle"))
import psa_amd._native as nat
import oracle as O
a0 = np.sqrt([0.5, 0.5, 1e-5, 1e-5]).astype(complex)
for N in (1 << 24, (1 << 24) + 12345):
    db = np.linspace(-0.05, 0.05, N)
    t = time.time()
    got = nat.sweep_host(db, n_steps=40, z_max=4.0, save_every=10, gamma=0.0115, alpha=1.15e-4, a0=a0)
    pick = np.array([0, 1, 255, 256, 65535, 65536, N // 2, N - 2, N - 1, (1 << 24) - 1 if N > (1 << 24) else N // 3])
    ref = O.sweep(db[pick], z_max=4.0, n=40, save_every=10, gamma=0.0115, alpha=1.15e-4, a0=a0)
    err = np.max(np.abs(got["a_end"][pick] - ref["a_end"]) / np.abs(ref["a_end"]))
    print(f"N={N}: kernel {got['elapsed_ms']:.2f} ms, wall {time.time() - t:.1f} s, sampled rel err {err:.2e}, all finite {bool((got['first_bad_step'] == -1).all())}, "
          f"p_max monotone check {bool(np.all(np.isfinite(got['p_max'])))}", flush=True)
    gain, bi, bg, nf = nat.gain_summary_host(got["p_max"], got["first_bad_step"], 1e-5)
    assert nf == N and bi == int(np.argmax(gain)), (nf, bi, int(np.argmax(gain)))
    print(f"   gain summary over {N} points: best index {bi}, n_finite {nf}", flush=True)
