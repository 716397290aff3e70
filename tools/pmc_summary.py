#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter_collection.csv files per kernel: mean counter value per launch.

HBM bytes per launch follow /opt/skills/guides/MI355X_MICROARCH.md §HBM: FETCH_SIZE and WRITE_SIZE are in
KiB, collected in SEPARATE passes; on gfx950 FETCH_SIZE reads exactly 1/2 of the bytes of a wide
(16 B/lane) coalesced stream, so the read side is doubled:
    hbm_bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024
Usage: pmc_summary.py <fetch_counter_collection.csv> <write_counter_collection.csv> [out.json]
"""
import csv
import json
import sys
from collections import defaultdict


def per_kernel(path, counter):
    acc = defaultdict(lambda: [0.0, 0])
    with open(path) as f:
        for row in csv.DictReader(f):
            if row["Counter_Name"] != counter:
                continue
            a = acc[row["Kernel_Name"]]
            a[0] += float(row["Counter_Value"])
            a[1] += 1
    return {k: (v[0] / v[1], v[1]) for k, v in acc.items()}


def main():
    f = per_kernel(sys.argv[1], "FETCH_SIZE")
    w = per_kernel(sys.argv[2], "WRITE_SIZE")
    out = []
    for k in sorted(set(f) | set(w), key=lambda k: -(2 * f.get(k, (0, 0))[0] + w.get(k, (0, 0))[0]) * max(f.get(k, (0, 1))[1], 1)):
        fk, n = f.get(k, (0.0, 0))
        wk, _ = w.get(k, (0.0, 0))
        out.append({"kernel": k, "launches": n, "fetch_size_kib_per_launch": fk, "write_size_kib_per_launch": wk,
                    "hbm_bytes_per_launch_corrected": (2 * fk + wk) * 1024})
    s = json.dumps(out, indent=1)
    if len(sys.argv) > 3:
        open(sys.argv[3], "w").write(s)
    print(s)


if __name__ == "__main__":
    main()
