#!/usr/bin/env python3
"""Condense the rocprofv3 output of tools/profile_r02.sh into what gets committed under profiles/:

    python tools/profile_summary.py gpurun_out/<dir> r02        ->  profiles/r02_<cfg>_kernel_stats.csv   (copied as is)
                                                                    profiles/r02_<cfg>_pmc.csv            (dominant kernel's counters per dispatch)
                                                                    profiles/kernels.json                 (read by bench.py)

kernels.json, per configuration: the dominant kernel's symbol, average / min duration, VALU instructions per wave per
z-step (SQ_INSTS_VALU / SQ_WAVES / n_zsteps), held clock (GRBM_GUI_ACTIVE / 8 XCDs / duration), and HBM bytes per launch
from the FETCH_SIZE / WRITE_SIZE passes with the guide's gfx950 correction (bytes = counter x 1024; FETCH_SIZE doubled).
"""
import csv
import glob
import json
import os
import shutil
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
N_Z = {"c2": 100_000, "c2blk": 100_000, "c3": 100_000, "c4": 1_000_000, "c5": 100_000, "c5one": 100_000, "traj": 400,
       "trajf32": 400, "traj6": 400, "traj4s": 3200, "traj6s": 2000}
LANES_PER_POINT = {"c2": 1.0, "c2blk": 1.0, "c3": 1.0, "c4": 0.5, "c5": 2.0, "c5one": 1.0, "traj": 1.0, "trajf32": 0.5, "traj6": 1.0,
                   "traj4s": 2.0, "traj6s": 2.0}
N_PTS = {"c2": 65_536, "c2blk": 65_536, "c3": 1_048_576, "c4": 131_072, "c5": 32_768, "c5one": 32_768, "traj": 262_144,
         "trajf32": 524_288, "traj6": 262_144, "traj4s": 32_768, "traj6s": 32_768}
KEY = {"traj": "trajectory", "trajf32": "traj_c4", "traj6": "traj_c5", "traj4s": "traj_c2split", "traj6s": "traj_c5split",
       "c2blk": "c2_block_check"}


def one(pattern):
    hits = sorted(glob.glob(pattern, recursive=True))
    return hits[0] if hits else None


def db_kernel_stats(db):
    """rocprofv3's default output on this image is a rocpd SQLite database: the per-kernel table of --stats from it."""
    import sqlite3
    con = sqlite3.connect(db)
    per = {}
    for name, dur in con.execute("select name, duration from kernels"):
        per.setdefault(name, []).append(float(dur))
    total = sum(sum(v) for v in per.values())
    rows = []
    for name, v in sorted(per.items(), key=lambda kv: -sum(kv[1])):
        rows.append({"Name": name, "Calls": len(v), "TotalDurationNs": sum(v), "AverageNs": sum(v) / len(v),
                     "Percentage": 100.0 * sum(v) / total, "MinNs": min(v), "MaxNs": max(v),
                     "StdDev": statistics.pstdev(v) if len(v) > 1 else 0.0})
    return rows


def db_counters(db, kernel):
    """-> ({counter: {dispatch: value}}, {dispatch: duration_ms}) of one kernel (values summed over XCDs / instances)."""
    import sqlite3
    con = sqlite3.connect(db)
    counters, durations = {}, {}
    for disp, cname, value, start, end in con.execute(
            "select dispatch_id, counter_name, value, start, end from counters_collection where kernel_name = ?", (kernel,)):
        counters.setdefault(cname, {})[int(disp)] = counters.get(cname, {}).get(int(disp), 0.0) + float(value)
        durations[int(disp)] = (int(end) - int(start)) / 1e6
    return counters, durations


def main(src, tag):
    out_dir = os.path.join(ROOT, "profiles")
    facts = {}
    path = os.path.join(out_dir, "kernels.json")
    if os.path.exists(path):
        facts = json.load(open(path))
    for cfg in N_Z:
        stats = one(os.path.join(src, f"{cfg}_stats", "**", "*_kernel_stats.csv"))
        stats_db = one(os.path.join(src, f"{cfg}_stats", "**", "*_results.db"))
        if stats:
            rows = list(csv.DictReader(open(stats)))
            shutil.copy(stats, os.path.join(out_dir, f"{tag}_{cfg}_kernel_stats.csv"))
        elif stats_db:
            rows = db_kernel_stats(stats_db)
            with open(os.path.join(out_dir, f"{tag}_{cfg}_kernel_stats.csv"), "w", newline="") as f:
                w = csv.DictWriter(f, fieldnames=list(rows[0]), quoting=csv.QUOTE_NONNUMERIC)
                w.writeheader()
                w.writerows(rows)
        else:
            continue
        dom = max(rows, key=lambda r: float(r["TotalDurationNs"]))
        kernel = dom["Name"]
        rec = {"kernel": kernel, "calls": int(dom["Calls"]), "avg_ms": float(dom["AverageNs"]) / 1e6,
               "min_ms": float(dom["MinNs"]) / 1e6, "lanes_per_point": LANES_PER_POINT[cfg],
               "source": f"rocprofv3 --kernel-trace --stats / --pmc passes of tools/profile_{tag}.sh ({tag}), "
                         f"profiles/{tag}_{cfg}_kernel_stats.csv + profiles/{tag}_{cfg}_pmc.csv"}
        counters = {}
        durations = {}
        for grp in ("sq", "grbm", "fetch", "write", "flops"):
            cc = one(os.path.join(src, f"{cfg}_pmc_{grp}", "**", "*_counter_collection.csv"))
            cdb = one(os.path.join(src, f"{cfg}_pmc_{grp}", "**", "*_results.db"))
            if cc:
                for r in csv.DictReader(open(cc)):
                    if r["Kernel_Name"] != kernel:
                        continue
                    counters.setdefault(r["Counter_Name"], {})[int(r["Dispatch_Id"])] = float(r["Counter_Value"])
                    durations.setdefault(grp, {})[int(r["Dispatch_Id"])] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
            elif cdb:
                c, d = db_counters(cdb, kernel)
                counters.update(c)
                durations[grp] = d
        if counters:
            with open(os.path.join(out_dir, f"{tag}_{cfg}_pmc.csv"), "w", newline="") as f:
                w = csv.writer(f)
                w.writerow(["kernel", "counter", "dispatches", "median", "min", "max"])
                for name, per in sorted(counters.items()):
                    v = list(per.values())
                    w.writerow([kernel, name, len(v), statistics.median(v), min(v), max(v)])
            med = {k: statistics.median(list(v.values())) for k, v in counters.items()}
            low = {k: min(v.values()) for k, v in counters.items()}
            rec["counters_median"] = med
            # waves of one launch = what the grid asks for.  SQ_WAVES can exceed it: bench.py copies pass k's outputs to the
            # host on a second stream under pass k+1's kernel, and for shards that fill the chip several times over (c3) the
            # hardware scheduler context-switches the resident waves once per pass to serve that queue (compute wave
            # save/restore): the restored waves are counted again and their register image shows up in FETCH / WRITE_SIZE.
            # Per-wave figures are therefore normalised by the grid's wave count, and the kernel's OWN traffic is read from
            # the dispatch that was not switched (the minimum); the switched one is kept as cwsr_bytes_per_launch.
            waves = -(-int(round(N_PTS[cfg] * LANES_PER_POINT[cfg])) // 64)
            rec["waves"] = float(waves)
            if "SQ_INSTS_VALU" in med:
                rec["valu_insts_per_wave_step"] = med["SQ_INSTS_VALU"] / waves / N_Z[cfg]
            if med.get("SQ_WAVES") and max(counters["SQ_WAVES"].values()) > waves:
                rec["sq_waves_max"] = max(counters["SQ_WAVES"].values())
            # executed floating-point work: wave-instruction counts of the FMA / MUL / ADD classes (an FMA = 2 flops per lane;
            # a packed float32 instruction does two per lane and is counted by SQ_INSTS_VALU_FLOPS_FP32 accordingly), per lane
            # per z-step.  The dedicated gfx950 counter SQ_INSTS_VALU_FLOPS_FP64/32 is kept beside the sum as a cross-check.
            for suf in ("F64", "F32"):
                fma, mul, add = (med.get(f"SQ_INSTS_VALU_{k}_{suf}") for k in ("FMA", "MUL", "ADD"))
                if fma is None:
                    continue
                if suf == "F32" and cfg not in ("c4", "trajf32"):
                    continue
                if suf == "F64" and cfg in ("c4", "trajf32"):
                    continue
                per_wave_step = (2 * fma + (mul or 0) + (add or 0)) / waves / N_Z[cfg]
                pack = 2.0 if suf == "F32" else 1.0          # v_pk_* : one wave-instruction = two lanes' worth per lane
                rec["executed_flops_per_lane_step"] = per_wave_step * pack
                rec["executed_flops_counters"] = {k: med[k] for k in med if k.startswith("SQ_INSTS_VALU_") and
                                                  k.split("_")[-1] in ("F64", "F32", "FP64", "FP32", "CVT")}
                flc = med.get("SQ_INSTS_VALU_FLOPS_FP64" if suf == "F64" else "SQ_INSTS_VALU_FLOPS_FP32")
                if flc:
                    rec["flops_counter_per_wave_step"] = flc / waves / N_Z[cfg]
            if "GRBM_GUI_ACTIVE" in med and "grbm" in durations:
                dur = statistics.median(list(durations["grbm"].values()))
                rec["held_clock_ghz"] = med["GRBM_GUI_ACTIVE"] / 8 / (dur * 1e-3) / 1e9
            if "FETCH_SIZE" in med and "WRITE_SIZE" in med:
                rec["fetch_bytes_corrected"] = low["FETCH_SIZE"] * 1024 * 2
                rec["write_bytes"] = low["WRITE_SIZE"] * 1024
                rec["hbm_bytes_per_launch"] = rec["fetch_bytes_corrected"] + rec["write_bytes"]
                switched = max(counters["FETCH_SIZE"].values()) * 2048 + max(counters["WRITE_SIZE"].values()) * 1024
                if switched > 1.5 * rec["hbm_bytes_per_launch"]:
                    rec["cwsr_bytes_per_launch"] = switched - rec["hbm_bytes_per_launch"]
                rec["traffic_correction"] = ("MI355X_MICROARCH.md HBM section: bytes = counter x 1024; FETCH_SIZE doubled on "
                                             "gfx950 (128-B requests tallied at 64 B)")
        facts[KEY.get(cfg, cfg)] = rec
        print(cfg, json.dumps({k: v for k, v in rec.items() if k not in ("counters_median", "source")}))
    json.dump(facts, open(path, "w"), indent=1, sort_keys=True)
    print("wrote", path)


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else "r02")
