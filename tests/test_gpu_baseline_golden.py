"""GPU parity at the BASELINE.json sizes (run with -m gpu): the HIP path through the C-ABI against
CPU known-answer values committed in tests/golden/baseline_configs.json.

The goldens come from tests/golden/make_baseline_golden.py -- a standalone NumPy/SciPy statement of
the same models (blocked Cholesky over <= 4096-wide potrf / trsm / gemm; it imports neither the
product nor oracle/), on inputs in the style of /root/reference/test/gp/util.jl:15-20,76-88 (seeded
standard normals, sigma^2 = 0.1).  north_star tolerance: logpdf and posterior mean / var within 1e-8
relative in fp64; the assertions below hold 1e-10 (the observed agreement is 1e-13 .. 1e-15).

These are the sizes bench.py times: c5 and `target` run the n_pad >= 32768 branch of the blocked
Cholesky (outer panels of 1024) that no oracle-sized test reaches.
"""
import os
import sys

import numpy as np
import pytest

import stheno_jl_amd as P

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench_configs as bc  # noqa: E402

pytestmark = pytest.mark.gpu

REL = 1e-10
DENSE = ["c1", "n4k", "c2", "c3", "n32k", "c5", "target", "w4k"]


def _rel(a, b):
    a, b = np.asarray(a, dtype=float), np.asarray(b, dtype=float)
    return float(np.max(np.abs(a - b)) / np.max(np.abs(b)))


@pytest.mark.parametrize("name", DENSE)
def test_logpdf_and_posterior_match_cpu_golden(name):
    g = bc.golden(name)
    assert g is not None, f"tests/golden/baseline_configs.json has no record for {name}"
    w = bc.build(P, name)
    assert g["N"] == w["N"]
    lp = P.logpdf(w["fx"], w["y"])
    assert abs(lp - g["logpdf"]) <= REL * abs(g["logpdf"]), (name, lp, g["logpdf"])
    post = P.posterior(w["fx"], w["y"])
    # alpha = C^-1 (y - m): head and norm against the CPU back-substitution
    assert _rel(post.alpha[:8], g["alpha_head"]) < 1e-8
    assert abs(float(post.alpha @ post.alpha) - g["alpha_norm2"]) <= 1e-9 * g["alpha_norm2"]
    m, v = post.mean_and_var(w["xs_new"])
    assert _rel(m, g["post_mean"]) < 1e-8, (name, _rel(m, g["post_mean"]))
    assert _rel(v, g["post_var"]) < 1e-8, (name, _rel(v, g["post_var"]))


def test_elbo_matches_cpu_golden():
    g = bc.golden("c4")
    assert g is not None
    w = bc.build(P, "c4")
    e = P.elbo(w["vfe"], w["fx"], w["y"])
    assert abs(e - g["elbo"]) <= REL * abs(g["elbo"]), (e, g["elbo"])


def test_many_columns_at_16k_match_golden_column():
    """logpdf(fx, Y) column-wise (AbstractGPs logpdf(fx, Y::Matrix)): column 0 is the golden y."""
    g = bc.golden("c2")
    w = bc.build(P, "c2")
    rng = np.random.default_rng(5)
    Y = np.column_stack([w["y"], rng.standard_normal((w["N"], 2))])
    out = P.logpdf(w["fx"], Y)
    assert abs(out[0] - g["logpdf"]) <= REL * abs(g["logpdf"])
    for j in (1, 2):
        assert abs(out[j] - P.logpdf(w["fx"], Y[:, j])) <= 1e-12 * abs(out[j])
