"""GPPP façade (host side): /root/reference/src/gaussian_process_probabilistic_programme.jl.

  GPPP(fs, gpc)              :13-18
  extract_components         :25-43   (incl. the generic tuple-vector path that regroups by key)
  @gppp let ... end          :166-201 -> `gppp(build)` / the `Programme` builder below: Python has
                             no macros, so the rewrite "GP(...) -> atomic(GP(...), gpc)" is done by
                             handing the model function a `GP` constructor bound to one fresh GPC.
"""
from __future__ import annotations

import numpy as np

from . import gp as _gp
from .inputs import BlockData, ColVecs, GPPPInput, regroup_pairs  # noqa: F401


class GPPP:
    """A group of related GPs interpreted as one GP; index it with GPPPInput / BlockData."""

    def __init__(self, fs, gpc):
        self.fs = dict(fs)
        self.gpc = gpc
        for f in self.fs.values():
            assert f.gpc is gpc, "all processes of a GPPP must share its GPC"

    def __call__(self, x, noise=1e-18):
        from .finite_gp import FiniteGP
        return FiniteGP(self, x, noise)


def gppp(build):
    """`f = gppp(lambda GP: dict(f1=GP(SEKernel()), ...))` -- the @gppp macro's effect.

    `build` receives a leaf constructor that wraps every GP(...) as atomic(GP(...), gpc) with one
    fresh GPC (gppp.jl:189-197) and returns an ordered mapping name -> process."""
    gpc = _gp.GPC()

    def wrapped_GP(*args):
        return _gp.atomic(_gp.GP(*args), gpc)

    fs = build(wrapped_GP)
    return GPPP(fs, gpc)


def gppp_sum_model():
    """The @gppp docstring model (gppp.jl:145-149): f1 ~ SE, f2 ~ Matern52, f3 = f1 + f2."""
    from .kernels import Matern52Kernel, SEKernel

    def build(GP):
        f1 = GP(SEKernel())
        f2 = GP(Matern52Kernel())
        return {"f1": f1, "f2": f2, "f3": f1 + f2}

    return gppp(build)


def extract_components(f, x):
    """-> (node, inputs): a single process + its inputs, or cross(fs) + BlockData."""
    if isinstance(x, GPPPInput):
        return f.fs[x.p], x.x
    if isinstance(x, BlockData):
        pairs = [extract_components(f, b) for b in x.X]
        return _gp.cross([p[0] for p in pairs]), BlockData([p[1] for p in pairs])
    # generic vector of (key, value): regroup by unique key in order of first appearance.
    # NOTE: like the reference (gppp.jl:32-43) this changes the element order.
    return extract_components(f, regroup_pairs(list(x)))
