"""Generates tests/golden/sklearn_gpr.json: known-answer vectors from an independent third
implementation (scikit-learn GaussianProcessRegressor, exact GP regression with fixed
hyper-parameters) for the logpdf / posterior path.

The reference (Stheno.jl) cannot be run in the build container (no Julia) and its own tests
hold no numeric golden vectors, so these vectors are what pins both the
CPU oracle (tests/test_oracle_vs_sklearn.py) and the HIP path (tests/test_gpu_parity.py).
Run:  python tests/golden/make_golden.py   (deterministic; commits the JSON next to it)."""
import json
import os

import numpy as np
from sklearn.gaussian_process import GaussianProcessRegressor
from sklearn.gaussian_process.kernels import RBF, Matern


def main():
    rng = np.random.default_rng(123456)
    cases = []
    for (kernel, N, D, l, s2) in [("se", 64, 2, 1.0, 0.1), ("matern52", 200, 3, 1.7, 0.05),
                                  ("matern32", 150, 1, 0.6, 0.2), ("se", 400, 8, np.sqrt(8.0), 0.1)]:
        X = rng.standard_normal((N, D))
        y = rng.standard_normal(N)
        Xs = rng.standard_normal((17, D))
        k = {"se": RBF(length_scale=l), "matern52": Matern(length_scale=l, nu=2.5),
             "matern32": Matern(length_scale=l, nu=1.5)}[kernel]
        gpr = GaussianProcessRegressor(kernel=k, alpha=s2, optimizer=None, normalize_y=False).fit(X, y)
        mean, std = gpr.predict(Xs, return_std=True)
        cases.append({"kernel": kernel, "lengthscale": float(l), "sigma2": s2, "X": X.tolist(), "y": y.tolist(),
                      "Xs": Xs.tolist(), "lml": float(gpr.log_marginal_likelihood_value_),
                      "mean": mean.tolist(), "std": std.tolist()})
    out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "sklearn_gpr.json")
    json.dump({"generator": "tests/golden/make_golden.py", "sklearn": __import__("sklearn").__version__,
               "cases": cases}, open(out, "w"))
    print("wrote", out, os.path.getsize(out), "bytes")


if __name__ == "__main__":
    main()
