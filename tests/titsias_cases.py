"""Small sparse-GP cases shared by the CPU and GPU dense-Titsias tests: inputs, inducing points z != x, observations,
noise (isotropic and non-isotropic), test points.  Sizes N <= 600, M <= 80: the dense statements are O(N^3)."""
import numpy as np


def single_gp_case(noise_kind, seed=20260926):
    rng = np.random.default_rng(seed)
    D, N, M, NS = 3, 600, 80, 37
    X = np.asfortranarray(rng.standard_normal((D, N)))
    Z = np.asfortranarray(1.3 * rng.standard_normal((D, M)))      # inducing points are NOT data points
    Xs = np.asfortranarray(rng.standard_normal((D, NS)))
    y = rng.standard_normal(N) + 0.4
    if noise_kind == "scalar":
        noise = 0.3
        sy = np.full(N, 0.3)
    else:
        noise = 0.05 + rng.random(N)                               # non-isotropic diagonal Sigma_y
        sy = noise
    return dict(X=X, Z=Z, Xs=Xs, y=y, noise=noise, sy=sy, ell=1.7, coef=2.5, mean=0.4, jitter=1e-6)


def gppp_case(seed=7):
    """x observed in f3 = f1 + f2 of the docstring model, inducing points in f1 (a different process of the programme),
    predictions in f2: every covariance is a cross-covariance."""
    rng = np.random.default_rng(seed)
    N, M, NS = 500, 60, 29
    return dict(x=np.sort(rng.uniform(-4, 4, N)), z=np.linspace(-4.5, 4.5, M) + 0.01 * rng.standard_normal(M),
                xs=np.sort(rng.uniform(-4, 4, NS)), y=rng.standard_normal(N), noise=0.05 + rng.random(N), jitter=1e-6)
