#!/usr/bin/env python3
"""A (pump-2 wavelength x signal wavelength) gain map sharded over the GPUs of one node -- what the reference would
do with a doubly nested Python loop around run_single_simulation (scan_mismtach.py:694-738 per row).

    python examples/sharded_grid.py                                  # one GPU, no process group
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 examples/sharded_grid.py   # 8 GPUs

Every rank owns a contiguous block of the flattened grid.  It generates the block's phase mismatch ON ITS GPU from the two
wavelength axes (psa_dbeta_grid_f64_dev: no per-point input is scattered), integrates the block with the RK4 sweep kernel,
reduces the per-point gain and the block's best point on the device, and contributes its output record to ONE RCCL
all_gather.  Rank 0 prints the map's maximum.
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402  (before the native library: one HIP runtime per process)
import torch.distributed as dist  # noqa: E402

import psa_amd._native as nat  # noqa: E402
from psa_amd.dispersion import dispersion_params_from_D_S  # noqa: E402
from psa_amd.distributed import DeviceSweep, shard_bounds, unpack_gathered  # noqa: E402
from psa_amd.phase_matching import PhaseMatchingConfig  # noqa: E402


def main():
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        dist.init_process_group("nccl", device_id=dev)

    lam_p1 = 1550e-9
    lam_p2 = np.linspace(1552e-9, 1562e-9, 256)          # rows
    lam_sig = np.linspace(1540e-9, 1565e-9, 512)         # columns
    disp = dispersion_params_from_D_S(1554e-9, 0.1, 0.02, 0.0, D_units="ps/nm/km", S_units="ps/nm^2/km",
                                      dSdlmbd_units="ps/nm^3/km")
    p_in = np.array([0.1, 0.1, 1e-7, 1e-7])
    N = lam_p2.size * lam_sig.size
    lo, hi = shard_bounds(N, world, rank)

    shard = DeviceSweep(n_local=hi - lo, n_steps=50_000, z_max=500.0, save_every=10, gamma=0.0115, alpha=1.15e-4,
                        a0=np.sqrt(p_in).astype(complex), device=dev, pad_to=(N + world - 1) // world)
    shard.fill_dbeta_grid(nat.dbeta_model(disp, PhaseMatchingConfig()), lam_p1, lam_p2, lam_sig, first=lo)
    shard.launch()                                       # the whole z-loop of every point of the block: one kernel
    shard.summarize(float(p_in[2]), mode="max", gain_db=True)
    if world > 1:
        words = shard.gather()                           # the single collective of the path
        torch.cuda.synchronize()
        a_end, p_end, p_max, first_bad = unpack_gathered(shard.layout, words.cpu().numpy(), N, world)
    else:
        torch.cuda.synchronize()
        r = shard.result()
        a_end, p_max, first_bad = r.a_end, r.p_max, r.first_bad_step
    if rank == 0:
        gain = 10.0 * np.log10(np.where(first_bad < 0, p_max / p_in[2], np.nan)).reshape(lam_p2.size, lam_sig.size)
        iy, ix = np.unravel_index(np.nanargmax(gain), gain.shape)
        print(f"{N} points on {world} GPU(s): max gain {gain[iy, ix]:.3f} dB at lambda_p2 = {lam_p2[iy] * 1e9:.3f} nm, "
              f"lambda_signal = {lam_sig[ix] * 1e9:.3f} nm; this rank's best: index {int(shard.best[0])} "
              f"({float(shard.best_gain):.3f} dB), {int((first_bad >= 0).sum())} failed points")
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
