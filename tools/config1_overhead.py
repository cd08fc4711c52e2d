import os, sys, time
import numpy as np
sys.path.insert(0, os.getcwd())
import psa_amd._native as nat
from psa_amd import config, simulation
from psa_amd.phase_matching import PhaseMatchingConfig
a0 = np.sqrt([0.5, 0.5, 1e-5, 1e-5]).astype(complex)
kw = dict(n_steps=10000, z_max=1000.0, save_every=10, gamma=0.0115, alpha=2.0723e-4, a0=a0, want_traj=True, exact_step=True)
nat.sweep_host([3.9e-4], **kw)
def med(f, n=30):
    ts = []
    for _ in range(n):
        t = time.perf_counter(); r = f(); ts.append(time.perf_counter() - t)
    return np.median(ts) * 1e3, r
w, r = med(lambda: nat.sweep_host([3.9e-4], **kw))
print(f"sweep_host N=1 10k steps traj: wall {w:.3f} ms, kernel {r['elapsed_ms']:.3f} ms -> C-ABI overhead {w - r['elapsed_ms']:.3f} ms")
kw2 = dict(kw); kw2["n_steps"] = 10; kw2["z_max"] = 1.0
w, r = med(lambda: nat.sweep_host([3.9e-4], **kw2))
print(f"sweep_host N=1 10 steps: wall {w:.3f} ms, kernel {r['elapsed_ms']:.3f} ms")
cfg = config.custom_simulation_config(z_max=1000.0, dz=0.1)
pm = PhaseMatchingConfig(method="provided", provided_delta_beta=3.926290731647635e-4)
kws = dict(gamma=0.0115, alpha=2.0723e-4, omega=[1.2e15] * 4, p_in=[0.5, 0.5, 1e-5, 1e-5], phase_matching_cfg=pm)
w, _ = med(lambda: simulation.run_single_simulation(cfg, **kws))
print(f"run_single_simulation: wall {w:.3f} ms")
cfg2 = config.custom_simulation_config(z_max=1.0, dz=0.1)
w, _ = med(lambda: simulation.run_single_simulation(cfg2, **kws))
print(f"run_single_simulation 10 steps: wall {w:.3f} ms")
