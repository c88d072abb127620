"""Fold the rocprofv3 outputs of tools/profile.sh into three small files (kernel_stats.csv, pmc.csv, pmc.json).

pmc.json carries the HBM traffic of one k_pairs launch the way MI355X_MICROARCH.md prescribes for gfx950:
FETCH_SIZE / WRITE_SIZE count kilobytes; FETCH_SIZE under-counts wide streaming reads by 2x on gfx950,
this kernel's reads are 4-16 B per lane (uncalibrated), so both the raw and the doubled figure are
recorded and bench.py quotes the doubled (conservative) one.
"""
import csv
import glob
import json
import os
import sys


def main(out):
    # 1. kernel stats
    rows = []
    for f in glob.glob(os.path.join(out, "stats", "**", "*kernel_stats.csv"), recursive=True):
        with open(f) as fh:
            rows += list(csv.DictReader(fh))
    if rows:
        with open(os.path.join(out, "kernel_stats.csv"), "w", newline="") as fh:
            w = csv.DictWriter(fh, fieldnames=list(rows[0].keys()))
            w.writeheader()
            w.writerows(rows)
    # 2. counters: sum over the dispatches of each kernel
    acc = {}
    for f in glob.glob(os.path.join(out, "pmc_*", "**", "*counter_collection.csv"), recursive=True):
        with open(f) as fh:
            for r in csv.DictReader(fh):
                k = (r["Kernel_Name"], r["Counter_Name"])
                a = acc.setdefault(k, [set(), 0.0])
                a[0].add(r["Dispatch_Id"])
                a[1] += float(r["Counter_Value"])
    pairs, workload = None, {"genomes": None, "seed": None, "slab_rows": None}
    try:
        with open(os.path.join(out, "stats.log")) as fh:
            for line in fh:
                if line.startswith("{") and '"config"' in line:
                    cfg = json.loads(line)["config"]
                    pairs = cfg.get("pairs_per_step")          # one k_pairs launch = one step = one slab
                    workload = {"genomes": cfg.get("genomes"), "seed": cfg.get("seed"), "slab_rows": cfg.get("slab_rows"),
                                "params": "default" if cfg.get("params", {}).get("mal") == 11 else cfg.get("params")}
    except Exception:
        pass
    with open(os.path.join(out, "pmc.csv"), "w", newline="") as fh:
        w = csv.writer(fh)
        w.writerow(["kernel", "counter", "launches", "sum_value", "per_launch", "per_pair"])
        for (kn, cn), (ds, v) in sorted(acc.items()):
            if "k_pairs" not in kn:
                continue
            per = v / max(1, len(ds))
            w.writerow([kn.split("(")[0], cn, len(ds), v, per, (per / pairs) if pairs else ""])
    fetch = [v / max(1, len(ds)) for (kn, cn), (ds, v) in acc.items() if "k_pairs" in kn and cn == "FETCH_SIZE"]
    write = [v / max(1, len(ds)) for (kn, cn), (ds, v) in acc.items() if "k_pairs" in kn and cn == "WRITE_SIZE"]
    if fetch and write:
        rec = {
            "source": "tools/profile.sh: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, per k_pairs launch",
            "workload": workload,
            "FETCH_SIZE_KB": fetch[0], "WRITE_SIZE_KB": write[0],
            "correction": "gfx950: FETCH_SIZE counts 64 B per 128 B request on wide streaming reads "
                          "(MI355X_MICROARCH.md, HBM); this kernel's reads are 4-16 B per lane, uncalibrated, "
                          "so both the raw and the doubled figure are given",
            "traffic_bytes_raw": (fetch[0] + write[0]) * 1024.0,
            "traffic_bytes_fetch_doubled": (2 * fetch[0] + write[0]) * 1024.0,
        }
        with open(os.path.join(out, "pmc.json"), "w") as fh:
            json.dump(rec, fh, indent=1)
    print("summary written to", out)


if __name__ == "__main__":
    main(sys.argv[1])
