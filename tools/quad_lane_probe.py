#!/usr/bin/env python3
"""Developer probe: four lanes per point (tools/quad_lane_probe.hip) against the library's two-lane and one-lane kernels --
step rate and agreement, 4 waves, float64, summary only (check_nan off so the library runs no finite test either).
Kept as profiles/r03_quad_lane_probe.log."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import psa_amd._native as nat  # noqa: E402

nat.lib()
Q = C.CDLL(os.path.join(ROOT, "tools", "libquad_probe.so"))
Q.quad_probe.restype = C.c_int
Q.quad_probe.argtypes = [C.c_longlong, C.c_int, C.c_double] + [C.c_void_p] * 2 + [C.c_double] * 2 + [C.c_void_p] * 2 + [C.c_int]
a0 = np.sqrt([0.5, 0.5, 1e-5, 1e-5]).astype(complex)
a0f = np.ascontiguousarray(np.stack([a0.real, a0.imag], 1).ravel())
n, L = 10_000, 1000.0
print(f"# {n} z-steps, 4 waves, float64, kernel ms (best of 3); lanes per point: 1 / 2 = the library (check_nan off), 4 = the probe")
for N in (1, 100, 4096, 16_384):
    db = np.ascontiguousarray(np.linspace(-0.02, 0.02, N))
    out = np.empty((N, 8))
    ms = C.c_double()
    assert Q.quad_probe(N, n, L, db.ctypes.data_as(C.c_void_p), a0f.ctypes.data_as(C.c_void_p), 0.0115, 1.15e-4,
                        out.ctypes.data_as(C.c_void_p), C.cast(C.byref(ms), C.c_void_p), 3) == 0
    quad = out[:, 0::2] + 1j * out[:, 1::2]
    kw = dict(n_steps=n, z_max=L, save_every=n, gamma=0.0115, alpha=1.15e-4, a0=a0, check_nan=False)
    t = {}
    for name, fl in (("one", nat.OPT_ONE_LANE), ("two", nat.OPT_SPLIT_POINT)):
        r = [nat.sweep_host(db, extra_flags=fl, **kw) for _ in range(4)]
        t[name] = min(x["elapsed_ms"] for x in r[1:])
        ref = r[-1]["a_end"]
    err = float(np.max(np.abs(quad - ref) / np.abs(ref)))
    print(f"N = {N:6d}: one lane {t['one']:7.3f}   two lanes {t['two']:7.3f}   four lanes {ms.value:7.3f}  "
          f"(x{ms.value / t['two']:.3f} of two lanes)   four-lane result vs two-lane: {err:.1e}", flush=True)
