"""stheno.jl_amd -- MI355X-native dense GP inference behind the Stheno.jl operator surface.

Host-side mirror of the reference's exported names (/root/reference/src/Stheno.jl:46-48 plus
the AbstractGPs / KernelFunctions names it re-exports, :4-6), so code written against Stheno.jl
reads the same here; all covariance / Cholesky arithmetic runs in libsthenomi.so (HIP, gfx950)
through the C-ABI of include/sthenomi.h.  Import name: `stheno_jl_amd` (see __graft_entry__).
"""
from . import lib  # noqa: F401
from .lib import PosDefException, SthenoMIError  # noqa: F401
from .inputs import BlockData, ColVecs, GPPPInput, blocks, split, vcat  # noqa: F401
from .kernels import (ConstantKernel, ExponentialKernel, KernelSum, Matern12Kernel,  # noqa: F401
                      Matern32Kernel, Matern52Kernel, PeriodicTransform, ScaledKernel, ScaleTransform,
                      ScaleTransformedKernel, TransformedKernel,
                      SEKernel, SqExponentialKernel, WhiteKernel, with_lengthscale)
from .gp import (GP, GPC, AtomicGP, DerivedGP, Periodic, Select, Shift, Stretch,  # noqa: F401
                 additive_gp, atomic, compose, cross, mean_vector, periodic, select, shift,
                 stretch)
from .gppp import GPPP, extract_components, gppp, gppp_sum_model  # noqa: F401
from .finite_gp import (VFE, ApproxPosteriorGP, FiniteGP, PosteriorGP, SparseFiniteGP,  # noqa: F401
                        cov, elbo, elbo_and_gradient, logpdf, logpdf_and_gradient, logpdf_batch, logpdf_f32, marginals, mean, mean_and_cov, mean_and_var,
                        posterior, posterior_mean_and_var_f32, prior_cov, prior_mean, prior_var, rand, sparse_cov, var)
from .flatten import build_spec  # noqa: F401
from .ordering import block_atoms, fill_reducing_order, permute_blocks, suggest_order_capi  # noqa: F401
