"""Per-queue busy time of the last bench step in a rocprofv3 kernel_trace.csv: which stream bounds
the step (the trailing-update stream or the panel stream of the look-ahead)."""
import collections
import csv
import sys


def union(iv):
    iv = sorted(iv)
    tot, (cs, ce) = 0, iv[0]
    for s, e in iv[1:]:
        if s > ce:
            tot += ce - cs
            cs, ce = s, e
        else:
            ce = max(ce, e)
    return tot + ce - cs


def main(path):
    rows = list(csv.DictReader(open(path)))
    for r in rows:
        r["s"], r["e"] = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    asm = [i for i, r in enumerate(rows) if "assemble_block" in r["Kernel_Name"]]
    step = rows[asm[-1]:]
    t0, t1 = step[0]["s"], max(r["e"] for r in step)
    print(f"last step: {(t1 - t0) / 1e6:.2f} ms, {len(step)} kernel dispatches")
    by = collections.defaultdict(list)
    for r in step:
        name = r["Kernel_Name"].split("(")[0].replace("sgp::", "").replace("void ", "")
        by[(r["Queue_Id"], name)].append((r["s"], r["e"]))
    for (q, name), iv in sorted(by.items(), key=lambda kv: -sum(e - s for s, e in kv[1]))[:8]:
        print(f"  queue {q} {name:28s} n={len(iv):5d} sum {sum(e - s for s, e in iv) / 1e6:9.2f} ms")
    for q in sorted({r["Queue_Id"] for r in step}):
        iv = [(r["s"], r["e"]) for r in step if r["Queue_Id"] == q]
        print(f"  queue {q}: busy {union(iv) / 1e6:.2f} ms of {(t1 - t0) / 1e6:.2f} ({union(iv) / (t1 - t0):.3f})")


if __name__ == "__main__":
    main(sys.argv[1])
