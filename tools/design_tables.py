#!/usr/bin/env python3
"""Developer tool: the measured tables of DESIGN.md section 5 as markdown, straight from the committed files
(profiles/r03_bench_*.json, profiles/kernels.json, profiles/r03_store_ceiling.log, profiles/r03_small_sweeps.log)."""
import json
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = lambda *a: os.path.join(ROOT, "profiles", *a)  # noqa: E731
K = json.load(open(P("kernels.json")))


def bench(name):
    return json.load(open(P(f"r03_bench_{name}.json")))


print("### main table")
rows = [("c2", "c2", "**c2** 65 536 × 4 × 1e5 f64 (headline; per-step check semantics)"), ("c2_block_check", "c2_block_check", "c2 `--block-check`"),
        ("c3", "c3", "c3 1 048 576 × 4 × 1e5 f64"), ("c4", "c4", "c4 131 072 × 4 × 1e6 f32 (one GPU's eighth)"),
        ("c5", "c5", "c5 32 768 × 6 × 1e5 f64 (one GPU's eighth)"), ("c5_one_lane", "c5one", "c5 forced to one lane per point")]
for f, key, label in rows:
    d, k = bench(f), K[key]
    r = d["roofline"]
    kern = k["kernel"].replace("void psa::", "").split("(")[0].replace(", ", ",")
    print(f"| {label} | `{kern}` | {r['kernel_ms_avg']:.2f} | {d['value'] / 1e9:.1f} | {r['achieved']:.1f} / {r['peak']} = **{r['frac']:.2f}** | "
          f"{k['valu_insts_per_wave_step']:.1f} | **{r['issue_frac_nominal']:.3f}** | {r['executed']['flops_per_lane_step']:.1f} flops/lane-step → "
          f"{r['executed']['achieved']:.1f} TF = **{r['executed']['frac']:.2f}** | {k.get('held_clock_ghz', 0):.2f} GHz |")
    print(f"    ms/step {d['ms_per_step']:.3f} resident {d['device_resident_ms_per_step']:.3f} traffic {k['hbm_bytes_per_launch'] / 1e6:.2f} MB")

print("\n### trajectories")
ceil = {}
cur = None
for line in open(P("r03_store_ceiling.log")):
    m = re.match(r"-- store-only ceiling, (\d+) points", line)
    if m:
        cur = int(m.group(1))
        ceil[cur] = {"dense": [], "padded": []}
    m = re.search(r"100 launches back to back ([\d.]+) ms each -> (\d+) GB/s", line)
    if m and cur and ", 0 dependent" in line and not line.startswith("BLOCKED"):
        ceil[cur]["dense"].append(int(m.group(2)))
    if m and cur and line.startswith("PADDED ld = n + 272"):
        ceil[cur]["padded"].append(int(m.group(2)))
for f, shape in (("traj", 262144), ("traj_f32", 524288), ("traj_six", 262144), ("traj_split4", 32768), ("traj_split6", 32768)):
    d = bench(f)
    r = d["roofline"]
    c = ceil[shape]
    dn, pd = (min(c["dense"]), max(c["dense"])), (min(c["padded"]), max(c["padded"]))
    print(f"{f:12s} {r['algorithmic_bytes_per_launch'] / 1e9:.2f} GB  {r['kernel_ms_avg']:.3f} ms -> {r['achieved'] / 1e3:.2f} TB/s ({r['frac']:.2f})  "
          f"ceiling dense {dn[0] / 1e3:.2f}-{dn[1] / 1e3:.2f} padded {pd[0] / 1e3:.2f}-{pd[1] / 1e3:.2f}  "
          f"ratio dense {r['achieved'] / dn[1]:.2f}-{r['achieved'] / dn[0]:.2f} padded {r['achieved'] / pd[1]:.2f}-{r['achieved'] / pd[0]:.2f}")

print("\n### small sweeps")
for line in open(P("r03_small_sweeps.log")):
    if line.startswith(("G1", "G2", "G3", "psa_rk4_sweep")):
        print(line.rstrip())
