"""Clock and MFMA-pipe occupancy of a kernel's dispatches from ONE rocprofv3 run with counters and the kernel trace:
  rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace --output-format csv -d DIR -o p -- <cmd>
GRBM_GUI_ACTIVE (summed over the 8 XCDs) / 8 / the dispatch's duration = the shader clock the chip held; MFMA busy cycles (summed
over the 1024 SIMDs) / (GUI_ACTIVE / 8 * 1024) = the share of cycles the matrix pipe was busy.  achieved rate = busy share x the
datasheet rate scaled to that clock (78.6 TFLOP/s at 2.4 GHz for fp64).
usage: python tools/pmc_clock.py DIR kernel-name-substring [flops_per_dispatch_avg]"""
import csv
import glob
import json
import os
import sys


def main():
    d, name = sys.argv[1], sys.argv[2]
    flops = float(sys.argv[3]) if len(sys.argv) > 3 else None
    cc = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    kt = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)
    if not cc or not kt:
        raise SystemExit("need both *counter_collection.csv and *kernel_trace.csv under " + d)
    dur = {}
    for r in csv.DictReader(open(kt[0])):
        if name in r["Kernel_Name"]:
            dur[r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-9
    acc = {}
    for r in csv.DictReader(open(cc[0])):
        if name in r["Kernel_Name"] and r["Dispatch_Id"] in dur:
            acc.setdefault(r["Dispatch_Id"], {})[r["Counter_Name"]] = acc.get(r["Dispatch_Id"], {}).get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    rows = []
    for k, v in acc.items():
        if "GRBM_GUI_ACTIVE" in v and "SQ_VALU_MFMA_BUSY_CYCLES" in v and dur[k] > 0:
            cyc = v["GRBM_GUI_ACTIVE"] / 8.0
            rows.append((dur[k], cyc / dur[k] / 1e9, v["SQ_VALU_MFMA_BUSY_CYCLES"] / (cyc * 1024.0)))
    if not rows:
        raise SystemExit("no dispatch of %r with both counters" % name)
    tot = sum(r[0] for r in rows)
    clk = sum(r[0] * r[1] for r in rows) / tot
    busy = sum(r[0] * r[2] for r in rows) / tot
    out = {"kernel": name, "dispatches": len(rows), "sum_duration_ms": tot * 1e3, "clock_ghz_duration_weighted": clk,
           "mfma_busy_frac_duration_weighted": busy, "fp64_matrix_peak_at_that_clock_tflops": 78.6 * clk / 2.4,
           "busy_x_peak_at_clock_tflops": busy * 78.6 * clk / 2.4,
           "clock_ghz_min_max": [min(r[1] for r in rows), max(r[1] for r in rows)]}
    if flops:
        out["achieved_tflops_from_durations"] = flops * len(rows) / tot / 1e12
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
