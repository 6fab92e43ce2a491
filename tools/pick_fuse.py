"""Chooses the SGP_FUSE_POTRF value for the rest of tools/r02_pass10.sh from the A/B lines and the bit-identity
test log: a fused variant is eligible only if tests/test_gpu_fused_potrf.py passed and every line of it holds
parity (< 1e-10 against the CPU goldens); it wins if it is faster than the separate launches at N = 16 384 and not
slower (1 %) at the small sizes.  Prints the reasoning, then the chosen value on the last line.
usage: pick_fuse.py fuse.txt pytest_fused.log"""
import sys


def main(fuse_txt, pytest_log):
    t = {}
    for ln in open(fuse_txt):
        p = ln.split()
        if len(p) < 4 or p[2] == "FAILED":
            print("unusable line:", ln.strip())
            continue
        f = int(p[0].split("=")[1])
        try:
            t[(f, p[1])] = (float(p[2]), float(p[3]))
        except ValueError:
            print("unusable line:", ln.strip())
    log = open(pytest_log).read()
    tests_ok = "rc=0" in log and "passed" in log      # pytest's exit code is 0 only if nothing failed or errored
    print("fused tests passed:", tests_ok)
    best, best_ms = 0, None
    if tests_ok and all((0, c) in t for c in ("c1", "n4k", "c2")):
        best_ms = t[(0, "c2")][0]
        for f in (3, 11):
            if not all((f, c) in t for c in ("c1", "n4k", "c2")):
                print(f"fuse={f}: incomplete")
                continue
            if any(t[(f, c)][1] > 1e-10 for c in ("c1", "n4k", "c2")):
                print(f"fuse={f}: parity lost", [t[(f, c)] for c in ("c1", "n4k", "c2")])
                continue
            small_ok = all(t[(f, c)][0] <= 1.01 * t[(0, c)][0] for c in ("c1", "n4k"))
            print(f"fuse={f}:", {c: t[(f, c)][0] for c in ("c1", "n4k", "c2")}, "vs", {c: t[(0, c)][0] for c in ("c1", "n4k", "c2")},
                  "small sizes ok:", small_ok)
            if small_ok and t[(f, "c2")][0] < best_ms:
                best, best_ms = f, t[(f, "c2")][0]
    print(best)


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
