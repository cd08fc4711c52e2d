#!/usr/bin/env python3
"""Developer probe, run under rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU: one launch per float64 lane layout (4 waves, 4 096
points x 20 000 steps, summary only) so that the instructions per wave and z-step of each layout can be read off."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import psa_amd._native as nat  # noqa: E402

db = np.linspace(-0.02, 0.02, 4096)
a0 = np.sqrt([0.5, 0.5, 1e-5, 1e-5]).astype(complex)
for fl in (nat.OPT_ONE_LANE, nat.OPT_SPLIT_POINT, nat.OPT_QUAD_POINT):
    r = nat.sweep_host(db, n_steps=20_000, z_max=2000.0, save_every=10, gamma=0.0115, alpha=1.15e-4, a0=a0, extra_flags=fl)
    print(fl, r["elapsed_ms"])
