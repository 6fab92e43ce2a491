"""Relative error of the GPU logpdf (and of the oracle's LAPACK path) against the committed 60-digit values of
tests/golden/illcond_truth.json -- the numbers behind test_logpdf_ill_conditioned_against_60_digit_reference, printed
(A/B of variants of the diagonal-block factorisation)."""
import json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry
P = entry.load_package()
import oracle.abstractgps as agp, oracle.kernelfunctions as kf, oracle.stheno as st
for c in json.load(open(os.path.join(ROOT, "tests", "golden", "illcond_truth.json")))["cases"]:
    x, y, s2, truth = np.array(c["x"]), np.array(c["y"]), c["noise"], float(c["logpdf"])
    try:
        lo = agp.logpdf(st.atomic(agp.GP(kf.SEKernel()), st.GPC())(x, s2), y)
    except Exception:
        lo = float("nan")
    try:
        lp = P.logpdf(P.atomic(P.GP(P.SEKernel()), P.GPC())(x, s2), y)
    except Exception as e:
        lp = float("nan")
    print(f"N={c['N']} noise={s2:g}: gpu rel err {abs(lp - truth) / abs(truth):.2e}, lapack {abs(lo - truth) / abs(truth):.2e}", flush=True)
