"""Model recipes written once against an "API namespace", so that the same Stheno programme can
be built with the oracle (oracle.*) and with the product (stheno_jl_amd) and compared.
The recipes are the models the reference's tests and examples use (cited per recipe)."""
import types

import numpy as np


def oracle_api():
    import oracle.abstractgps as agp
    import oracle.kernelfunctions as kf
    import oracle.stheno as st
    ns = types.SimpleNamespace()
    for name in ("GPC", "atomic", "stretch", "select", "periodic", "shift", "compose", "cross",
                 "additive_gp", "GPPP", "GPPPInput", "BlockData"):
        setattr(ns, name, getattr(st, name))
    for name in ("SEKernel", "Matern12Kernel", "Matern32Kernel", "Matern52Kernel", "WhiteKernel",
                 "ConstantKernel", "ScaledKernel", "KernelSum", "with_lengthscale", "ColVecs", "PeriodicTransform",
                 "ScaleTransform", "TransformedKernel"):
        setattr(ns, name, getattr(kf, name))
    ns.GP = agp.GP
    ns.is_oracle = True
    return ns


def product_api():
    import stheno_jl_amd as p
    ns = types.SimpleNamespace()
    for name in ("GPC", "atomic", "stretch", "select", "periodic", "shift", "compose", "cross",
                 "additive_gp", "GPPP", "GPPPInput", "BlockData", "SEKernel", "Matern12Kernel",
                 "Matern32Kernel", "Matern52Kernel", "WhiteKernel", "ConstantKernel", "ScaledKernel",
                 "KernelSum", "with_lengthscale", "ColVecs", "GP", "PeriodicTransform", "ScaleTransform",
                 "TransformedKernel"):
        setattr(ns, name, getattr(p, name))
    ns.is_oracle = False
    return ns


def _sumsin(x):
    return float(np.sum(np.sin(x)))


def _sumcos(x):
    return float(np.sum(np.cos(x)))


# ---------------------------------------------------------------------------------------------
# each recipe: api -> (dict name -> process, gpc)
# ---------------------------------------------------------------------------------------------
def gppp_docstring(api):
    """@gppp docstring model, gaussian_process_probabilistic_programme.jl:145-149."""
    gpc = api.GPC()
    f1 = api.atomic(api.GP(api.SEKernel()), gpc)
    f2 = api.atomic(api.GP(api.Matern52Kernel()), gpc)
    return {"f1": f1, "f2": f2, "f3": f1 + f2}, gpc


def toy_gppp(api):
    """test/gaussian_process_probabilistic_programme.jl:19-26: f3 = f1 + 3 f2 with means."""
    gpc = api.GPC()
    f1 = api.atomic(api.GP(np.sin, api.SEKernel()), gpc)
    f2 = api.atomic(api.GP(np.cos, api.Matern52Kernel()), gpc)
    return {"f1": f1, "f2": f2, "f3": f1 + 3 * f2}, gpc


def correlated_sums(api):
    """test/affine_transformations/addition.jl:3-9: f4 = f1 + f3, f5 = f3 + f4."""
    gpc = api.GPC()
    f1 = api.atomic(api.GP(1, api.SEKernel()), gpc)
    f2 = api.atomic(api.GP(2, api.SEKernel()), gpc)
    f3 = f1 + f2
    f4 = f1 + f3
    f5 = f3 + f4
    return {"f1": f1, "f2": f2, "f3": f3, "f4": f4, "f5": f5, "f6": f2 - f1}, gpc


def scaled(api):
    """test/affine_transformations/product.jl:8-14,55-60: constant and function scaling."""
    gpc = api.GPC()
    g1 = api.atomic(api.GP(1, api.SEKernel()), gpc)
    c, c2 = -4.3, 2.1
    g2, g2p = c * g1, g1 * c2
    g3, g3p = c * g2, g2p * c2
    h2, h2p = _sumsin * g1, g1 * _sumcos
    h3 = _sumsin * h2
    return {"g1": g1, "g2": g2, "g2p": g2p, "g3": g3, "g3p": g3p, "h2": h2, "h2p": h2p, "h3": h3,
            "mix": h2 + g3p}, gpc


def warped(api):
    """test/affine_transformations/compose.jl: stretch / shift / select / arbitrary maps."""
    gpc = api.GPC()
    f = api.atomic(api.GP(np.sin, api.SEKernel()), gpc)
    h = api.atomic(api.GP(np.exp, api.Matern12Kernel()), gpc)
    return {"f": f, "h": h, "fs": api.stretch(f, 0.51), "fsh": api.shift(f, 0.3),
            "fcos": api.compose(f, np.cos), "sum": api.stretch(f, 0.51) + 2.0 * h,
            "hs": api.stretch(h, 2.0)}, gpc


def warped_colvecs(api, D=3):
    gpc = api.GPC()
    f = api.atomic(api.GP(1.3, api.SEKernel()), gpc)
    g = api.atomic(api.GP(api.Matern32Kernel()), gpc)
    lam = np.array([0.7, 1.3, 0.4])[:D]
    A = np.array([[0.5, 0.1, 0.0], [0.2, 1.1, -0.3], [0.0, 0.4, 0.9]])[:D, :D]
    a = np.array([0.1, -0.2, 0.3])[:D]
    return {"f": f, "g": g, "fs": api.stretch(f, 0.6), "fv": api.stretch(f, lam), "fA": api.stretch(g, A),
            "fsh": api.shift(f, a), "fsel": api.select(g, [0, D - 1]), "fsel1": api.select(f, 1),
            "add": api.additive_gp([f, g, f][:D])}, gpc


def composite_kernels(api):
    """examples/*: scaled / summed / lengthscaled leaf kernels (KernelFunctions composites)."""
    gpc = api.GPC()
    k1 = api.ScaledKernel(api.with_lengthscale(api.SEKernel(), 0.7), 2.5)
    k2 = api.KernelSum([api.Matern32Kernel(), api.ScaledKernel(api.WhiteKernel(), 0.1)])
    k3 = api.KernelSum([api.ConstantKernel(0.4), api.with_lengthscale(api.Matern52Kernel(), 1.9)])
    f1 = api.atomic(api.GP(k1), gpc)
    f2 = api.atomic(api.GP(0.5, k2), gpc)
    f3 = api.atomic(api.GP(k3), gpc)
    return {"f1": f1, "f2": f2, "f3": f3, "s": f1 + 0.5 * f2 - f3}, gpc


def periodic_model(api):
    gpc = api.GPC()
    f = api.atomic(api.GP(api.SEKernel()), gpc)
    g = api.atomic(api.GP(api.Matern32Kernel()), gpc)
    return {"g": g, "p": api.periodic(f, 2.0), "s": api.periodic(f, 0.5) + g, "pp": api.periodic(f, 0.5) + api.periodic(f, 2.0)}, gpc


def mauna_loa(api):
    """examples/extended_mauna_loa/script.jl:118-137: shared trend, kernel-level PeriodicTransform,
    ConstantKernel offsets, two output processes."""
    gpc = api.GPC()
    trend = api.stretch(api.atomic(api.GP(api.SEKernel()), gpc), 0.3)
    wig = 0.4 * api.stretch(api.atomic(api.GP(api.SEKernel()), gpc), 3.0)
    per = 0.8 * api.atomic(api.GP(api.SEKernel() @ api.PeriodicTransform(1.3)), gpc)
    per2 = api.atomic(api.GP(api.with_lengthscale(api.Matern32Kernel() @ api.PeriodicTransform(0.5), 2.0)), gpc)
    co2 = 1.7 * trend + wig + per + 0.6 * api.atomic(api.GP(api.ConstantKernel()), gpc)
    temp = 0.9 * trend + 0.5 * api.stretch(api.atomic(api.GP(api.SEKernel()), gpc), 2.0) + per2
    return {"trend": trend, "co2": co2, "T": temp, "per": per, "per2": per2}, gpc


def sensor_fusion(api):
    """examples/sensor_fusion/script.jl:33-45: a latent process seen through two sensors -- white noise with a known,
    input-dependent mean (`GP + function`, addition.jl:73-86) and white noise with a constant bias."""
    gpc = api.GPC()
    f = api.atomic(api.GP(api.SEKernel()), gpc)
    noise1 = np.sqrt(1e-2) * api.atomic(api.GP(api.WhiteKernel()), gpc) + (lambda x: float(np.sin(x) - 5.0 + np.sqrt(abs(x))))
    noise2 = np.sqrt(1e-1) * api.atomic(api.GP(3.5, api.WhiteKernel()), gpc)
    return {"f": f, "noise1": noise1, "noise2": noise2, "y1": f + noise1, "y2": f + noise2}, gpc


def time_varying_blr(api):
    """examples/time_varying_blr/script.jl:22-29: time-varying basis functions times slowly varying GP weights plus
    rough temporally correlated noise (function-scaled processes, product.jl:25-48)."""
    gpc = api.GPC()
    w1 = api.stretch(api.atomic(api.GP(api.SEKernel()), gpc), 0.2)
    w2 = api.stretch(api.atomic(api.GP(api.SEKernel()), gpc), 1.0)
    f = (lambda x: float(np.sum(x)) / 4) * w1 + (lambda x: float(np.sum(np.cos(x)))) * w2
    y = f + 0.3 * api.atomic(api.GP(api.Matern12Kernel()), gpc)
    return {"w1": w1, "w2": w2, "f": f, "y": y}, gpc


def pseudo_points(api):
    """examples/gppp_and_pseudo_points/script.jl:9-13: f1 = periodic(GP(SE), w), f2 = GP(0.1 * SE), f3 = f1 + f2."""
    gpc = api.GPC()
    f1 = api.periodic(api.atomic(api.GP(api.SEKernel()), gpc), 1.0)
    f2 = api.atomic(api.GP(api.ScaledKernel(api.SEKernel(), 0.1)), gpc)
    return {"f1": f1, "f2": f2, "f3": f1 + f2}, gpc


# recipes that so far only the CPU suites use (flattening against the recursion, the host mirror on the NumPy double);
# they join RECIPES_1D -- and with it the -m gpu covariance tests -- once they have run on the device
RECIPES_1D_CPU_ONLY = [sensor_fusion, time_varying_blr, pseudo_points]

RECIPES_1D = [gppp_docstring, toy_gppp, correlated_sums, warped, composite_kernels, periodic_model, mauna_loa]
RECIPES_ND = [gppp_docstring, correlated_sums, scaled, warped_colvecs, composite_kernels]
