"""The dominant kernel's rate WHILE IT IS ON THE CHIP, from a rocprofv3 kernel trace alone (round-5 verdict, item 3: under
the hybrid schedule two update streams and the panel kernel overlap, so rocprofv3's per-kernel average duration understates
the kernel's rate -- 61 x 25.2 ms > the 1396 ms step it was traced in -- and the only figures that isolated it came from the
library's own HIP events).  Per step of the traced bench.py command (a step starts with the assembly kernel):
  union of the begin / end stamps of the named kernel's dispatches   -> busy_ms
  sum of their durations                                              -> what `--stats` averages
  step span (first dispatch of the step to the last one's end)
and, with the bench line of the same command (its roofline.algorithmic_flops_per_launch_avg x launches = flops per step),
achieved_while_busy = flops / busy_ms.  usage:
  python tools/trace_busy.py <kernel_trace.csv> <bench_line.json> [kernel substring = gemm_nt_dma_kernel<1>] > out.json"""
import csv
import json
import sys

PEAK = 78.6


def union(iv):
    iv = sorted(iv)
    if not iv:
        return 0
    tot, (cs, ce) = 0, iv[0]
    for s, e in iv[1:]:
        if s > ce:
            tot += ce - cs
            cs, ce = s, e
        else:
            ce = max(ce, e)
    return tot + ce - cs


def main():
    path, bench = sys.argv[1], sys.argv[2]
    pat = sys.argv[3] if len(sys.argv) > 3 else "gemm_nt_dma_kernel<1>"
    rows = []
    for r in csv.DictReader(open(path)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    rows.sort()
    starts = [i for i, r in enumerate(rows) if "assemble_block" in r[2]]
    # a step starts with its assembly launches (one per block pair and term group: a CONTIGUOUS run of dispatches); keep the
    # first launch of every run
    firsts = [i for k, i in enumerate(starts) if k == 0 or i != starts[k - 1] + 1]
    line = json.loads(open(bench).read().strip().splitlines()[-1])
    roof = line.get("roofline") or {}
    flops = (roof.get("algorithmic_flops_per_launch_avg") or 0.0) * (roof.get("launches") or 0)
    steps = []
    for k, i in enumerate(firsts):
        j = firsts[k + 1] if k + 1 < len(firsts) else len(rows)
        seg = rows[i:j]
        iv = [(s, e) for s, e, n in seg if pat in n]
        if not iv:
            continue
        busy = union(iv) / 1e6
        steps.append({"dispatches_of_kernel": len(iv), "busy_ms": busy, "sum_of_durations_ms": sum(e - s for s, e in iv) / 1e6,
                      "avg_duration_ms": sum(e - s for s, e in iv) / 1e6 / len(iv),
                      "step_span_ms": (max(e for s, e, n in seg) - seg[0][0]) / 1e6,
                      "achieved_while_busy_tflops": (flops / (busy * 1e-3) / 1e12) if flops and busy > 0 else None})
    n_l = roof.get("launches")
    full = [s for s in steps if not n_l or s["dispatches_of_kernel"] == n_l] or steps
    full.sort(key=lambda s: s["busy_ms"])
    med = full[len(full) // 2] if full else None
    out = {"trace": path.split("/")[-1], "kernel": pat, "steps_in_trace": len(steps), "flops_per_step_of_the_kernel": flops,
           "source_of_flops": "roofline.algorithmic_flops_per_launch_avg x roofline.launches of the traced command's own line",
           "median_step": med,
           "achieved_while_busy_frac_of_78.6": (med["achieved_while_busy_tflops"] / PEAK) if med and med["achieved_while_busy_tflops"] else None,
           "in_situ_frac_from_avg_duration": ((flops / (med["sum_of_durations_ms"] * 1e-3) / 1e12 / PEAK) if med and flops else None),
           "all_steps": steps}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
