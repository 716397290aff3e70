#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc passes of the SQ / GRBM counters per kernel (mean per launch) and derive what the guide defines
(/opt/skills/guides/MI355X_MICROARCH.md "rocprofv3 PMC slots", "DVFS give-back"):

    clock under load    = GRBM_GUI_ACTIVE / 8 / kernel duration            (the counter is summed over the 8 XCDs; reads high below ~0.3 ms)
    mfma_busy_frac      = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 * CUs * 4 SIMDs)      (matrix pipe busy cycles over the cycles the chip ran)
    wave-state split    = SQ_WAIT_ANY : SQ_WAIT_INST_ANY : SQ_ACTIVE_INST_ANY over SQ_WAVE_CYCLES (quad-cycles; disjoint), SQ_WAIT_INST_LDS a
                          sub-bucket of WAIT_INST_ANY
Usage: pmc_mfma_summary.py out.json counter_collection.csv [more.csv ...]   (kernels of < 1 % of the summed GRBM_GUI_ACTIVE are dropped)"""
import csv
import json
import sys
from collections import defaultdict

CUS = 256


def main():
    out_path, paths = sys.argv[1], sys.argv[2:]
    acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
    dur = defaultdict(lambda: [0.0, 0])
    for p in paths:
        seen = set()
        with open(p) as f:
            for row in csv.DictReader(f):
                k = row["Kernel_Name"]
                a = acc[k][row["Counter_Name"]]
                a[0] += float(row["Counter_Value"])
                a[1] += 1
                did = (p, row.get("Dispatch_Id"))
                if did not in seen and row.get("Start_Timestamp") and row.get("End_Timestamp"):
                    seen.add(did)
                    d = dur[k]
                    d[0] += float(row["End_Timestamp"]) - float(row["Start_Timestamp"])
                    d[1] += 1
    rows = []
    for k, cs in acc.items():
        m = {c: v[0] / v[1] for c, v in cs.items()}
        n = max(v[1] for v in cs.values())
        r = {"kernel": k, "launches_per_pass": n, "counters_per_launch": m}
        gui = m.get("GRBM_GUI_ACTIVE")
        if dur[k][1]:
            r["avg_duration_us_under_pmc"] = dur[k][0] / dur[k][1] / 1e3
        if gui:
            cyc = gui / 8.0
            if "SQ_VALU_MFMA_BUSY_CYCLES" in m:
                r["mfma_busy_frac"] = m["SQ_VALU_MFMA_BUSY_CYCLES"] / (cyc * CUS * 4)
            if "SQ_BUSY_CYCLES" in m:
                r["sq_busy_cycles_over_chip_cycles"] = m["SQ_BUSY_CYCLES"] / cyc
            if dur[k][1]:
                r["clock_ghz_under_load"] = cyc / (dur[k][0] / dur[k][1])
        wc = m.get("SQ_WAVE_CYCLES")
        if wc:
            for c, name in (("SQ_WAIT_ANY", "wave_parked_frac"), ("SQ_WAIT_INST_ANY", "issue_stall_frac"), ("SQ_ACTIVE_INST_ANY", "issuing_frac"),
                            ("SQ_WAIT_INST_LDS", "lds_issue_stall_frac")):
                if c in m:
                    r[name] = m[c] / wc
        r["_weight"] = (gui or 0.0) * n
        rows.append(r)
    tot = sum(r["_weight"] for r in rows) or 1.0
    rows = [r for r in sorted(rows, key=lambda r: -r["_weight"]) if r["_weight"] >= 0.01 * tot or tot == 1.0]
    for r in rows:
        r["share_of_gpu_cycles"] = r.pop("_weight") / tot
    s = json.dumps(rows, indent=1)
    open(out_path, "w").write(s)
    print(s)


if __name__ == "__main__":
    main()
