#!/usr/bin/env python3
"""Where the first call of a process spends its time: library load, HIP initialisation (first API call), the first launch from
each code object (module load), steady state.   python tools/first_call.py"""
import os
import sys
import time

import numpy as np

t0 = time.perf_counter()
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import psa_amd._native as nat  # noqa: E402
t1 = time.perf_counter()
n_dev = nat.device_count()
t2 = time.perf_counter()
print(f"import + dlopen {1e3 * (t1 - t0):.1f} ms; psa_device_count() = {n_dev} (no context yet) {1e3 * (t2 - t1):.1f} ms")
a0 = np.sqrt([0.5, 0.5, 1e-5, 1e-5]).astype(complex)
kw = dict(n_steps=10, z_max=0.1, save_every=10, gamma=0.0115, alpha=1e-4, a0=a0)


def timed(label, **extra):
    t = time.perf_counter()
    nat.sweep_host(np.zeros(1), **{**kw, **extra})
    print(f"{label:58s} {1e3 * (time.perf_counter() - t):8.2f} ms")


timed("first float64 call (HIP context + float64 code object)")
timed("second float64 call")
timed("first float32 call (float32 code object)", dtype=np.float32)
timed("second float32 call", dtype=np.float32)
timed("first one-lane float64 call (same code object)", extra_flags=nat.OPT_ONE_LANE)
timed("steady state")
