#!/usr/bin/env python3
"""Developer probe: is trajectory mode (save_every = 1) slow because the chip has not ramped its clock up yet, or because
it throttles under FP64 + HBM-store load?  (1) 300 trajectory launches back to back, per-launch HIP-event times;
(2) the same right after 10 launches of the compute-only C2 kernel (0.6 s of full-rate FP64: clock certainly up)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from psa_amd.distributed import DeviceSweep

a0 = np.sqrt(np.array([0.5, 0.5, 1e-5, 1e-5])).astype(complex)
dev = torch.device("cuda", 0)
pts, nz = 262_144, 400
traj = DeviceSweep(np.linspace(-0.05, 0.05, pts), n_steps=nz, z_max=nz * 0.01, save_every=1, gamma=0.0115, alpha=1.15e-4, a0=a0, device=dev)
nbytes = traj.enable_trajectory()
summ = DeviceSweep(np.linspace(-0.05, 0.05, pts), n_steps=nz, z_max=nz * 0.01, save_every=1, gamma=0.0115, alpha=1.15e-4, a0=a0, device=dev)
heat = DeviceSweep(np.linspace(-0.05, 0.05, 65536), n_steps=100_000, z_max=1000.0, save_every=10, gamma=0.0115, alpha=1.15e-4, a0=a0, device=dev)


def timed(sweep, n):
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    for e0, e1 in ev:
        e0.record(); sweep.launch(); e1.record()
    torch.cuda.synchronize()
    return np.array([e0.elapsed_time(e1) for e0, e1 in ev])


traj.launch(); summ.launch(); heat.launch(); torch.cuda.synchronize()
show = [0, 1, 2, 5, 10, 20, 50, 100, 200, 299]
t = timed(traj, 300)
print("cold start, 300 trajectory launches:", " ".join(f"[{i}] {t[i]:.3f}" for i in show), f"| mean of last 100: {t[-100:].mean():.3f} ms = {nbytes / t[-100:].mean() / 1e6:.0f} GB/s", flush=True)
t = timed(summ, 300)
print("same sweep WITHOUT the trajectory (compute only):", " ".join(f"[{i}] {t[i]:.3f}" for i in show), f"| mean of last 100: {t[-100:].mean():.3f} ms", flush=True)
th = timed(heat, 10)
t = timed(traj, 300)
print(f"after 10 C2 launches ({th.mean():.1f} ms each):", " ".join(f"[{i}] {t[i]:.3f}" for i in show), f"| mean of last 100: {t[-100:].mean():.3f} ms = {nbytes / t[-100:].mean() / 1e6:.0f} GB/s", flush=True)
th = timed(heat, 10)
t = timed(summ, 300)
print(f"compute only after the same heater:", " ".join(f"[{i}] {t[i]:.3f}" for i in show), f"| mean of last 100: {t[-100:].mean():.3f} ms", flush=True)
