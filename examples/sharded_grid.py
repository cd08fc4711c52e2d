#!/usr/bin/env python3
"""A (pump-2 wavelength x signal wavelength) gain map over the GPUs of one node -- what the reference would do with a
doubly nested Python loop around run_single_simulation (scan_mismtach.py:694-738 per row).

    python examples/sharded_grid.py                                  # one GPU
    python examples/sharded_grid.py --devices 0,1,2,3                # one process, one host thread per GPU
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 examples/sharded_grid.py   # one process per GPU

The driver call is the same in all three: under a process group ``scan_gain_grid`` gives every rank a contiguous block of
the flattened grid; the rank generates the block's phase mismatch ON ITS GPU from the two wavelength axes (no per-point
input is scattered), integrates it with the RK4 sweep kernel and contributes its output record to ONE RCCL all_gather;
every rank returns the whole map.
"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--devices", default=None, help="comma-separated GPU ordinals for the one-process form")
    args = ap.parse_args()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1:
        import torch
        import torch.distributed as dist
        torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
        dist.init_process_group("nccl", device_id=torch.device("cuda", torch.cuda.current_device()))   # nccl == RCCL
    from psa_amd import config, scan_mismtach
    from psa_amd.dispersion import dispersion_params_from_D_S

    lam_p2 = np.linspace(1552e-9, 1562e-9, 256)          # rows
    lam_sig = np.linspace(1540e-9, 1565e-9, 512)         # columns
    disp = dispersion_params_from_D_S(1554e-9, 0.1, 0.02, 0.0, D_units="ps/nm/km", S_units="ps/nm^2/km",
                                      dSdlmbd_units="ps/nm^3/km")
    out = scan_mismtach.scan_gain_grid(cfg=config.custom_simulation_config(z_max=500.0, dz=0.01), lambda_p1_m=1550e-9,
                                       lambda_p2_m=lam_p2, lambda_signal_m=lam_sig, gamma=0.0115, alpha=1.15e-4,
                                       p_in=[0.1, 0.1, 1e-7, 1e-7], dispersion=disp,
                                       devices=[int(d) for d in args.devices.split(",")] if args.devices else None)
    if int(os.environ.get("RANK", "0")) == 0:
        iy, ix = out["best_index"]
        print(f"{out['gain'].size} points on {world} rank(s): max gain {out['best_gain']:.3f} dB at lambda_p2 = "
              f"{lam_p2[iy] * 1e9:.3f} nm, lambda_signal = {lam_sig[ix] * 1e9:.3f} nm; {out['n_finite']} finite points")
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
