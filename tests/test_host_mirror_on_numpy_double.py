"""The host mirror's logic WITHOUT a GPU: the bodies of the GPU parity suite (tests/test_gpu_parity.py) re-run with
libsthenomi.so replaced by the NumPy double of its C-ABI (tests/np_capi.py).

What this proves: that the Python side -- flattening (flatten.py), kernel expansion (kernels.py), argument
marshalling and result mapping (finite_gp.py: FiniteGP noise kinds, posterior / sequential conditioning, rand,
SparseFiniteGP / VFE dispatch, gradient records with mirror terms and function scales, the input-gradient chain rule
through the model's warps, PosDefException(info), ragged and empty blocks) -- turns a Stheno programme into the calls
whose documented results (include/sthenomi.h) equal the oracle's.  What it does NOT prove: anything about the HIP
kernels; the same bodies run against the real library under `-m gpu`.  The product itself has no CPU path."""
import numpy as np
import pytest

import np_capi
import test_gpu_parity as G


@pytest.fixture(autouse=True)
def _numpy_double(monkeypatch):
    np_capi.install(monkeypatch)


# the test bodies, collected here WITHOUT the module-level gpu mark of tests/test_gpu_parity.py
_REUSED = [
    "test_cov_mean_var_blockdata_1d", "test_cov_colvecs", "test_cov_and_logpdf_high_dimensional_inputs",
    "test_cov_many_terms_per_block_pair_accumulate_path", "test_warps_colvecs", "test_exact_identities_from_reference_tests",
    "test_logpdf_single_gp", "test_logpdf_rand_posterior_gppp", "test_posterior_external_consistency",
    "test_sequential_conditioning_posterior_of_posterior", "test_non_positive_definite_raises_posdef",
    "test_rand_sum_model_consistency", "test_sparse_finite_gp_reference_properties", "test_elbo_equals_logpdf_when_z_is_x",
    "test_elbo_gppp_with_diag_noise", "test_golden_sklearn_vectors", "test_ragged_and_empty_blocks",
    "test_many_right_hand_sides", "test_input_dimension_limit_is_reported", "test_logpdf_gradient_terms_noise_y_mean",
    "test_logpdf_gradient_matches_finite_differences_of_hyperparameters",
    "test_logpdf_gradient_records_with_scale_only_differences", "test_logpdf_gradient_wrt_function_scales",
    "test_elbo_gradient_wrt_function_scales",
    "test_logpdf_gradient_wrt_input_points", "test_logpdf_input_gradient_matches_finite_differences",
    "test_input_gradients_chain_through_model_transformations", "test_elbo_gradient_against_oracle_cotangents",
    "test_elbo_gradient_matches_finite_differences_of_hyperparameters", "test_elbo_input_gradients_match_finite_differences",
    "test_elbo_gradient_with_dense_inducing_noise",
    "test_gradients_at_tiny_and_tile_boundary_sizes",
]
# left to the GPU suite: test_library_is_native_and_loaded (about the .so), test_rand_statistics (100 000 samples),
# test_elbo_and_sparse_posterior_row_chunked_path (a device memory-layout knob),
# test_logpdf_ill_conditioned_against_60_digit_reference (about the device's panel solves)
for _name in _REUSED:
    globals()[_name] = getattr(G, _name)
del _name

# Float32 tagging and type stability of the host mirror (test/gp/util.jl:76-88): the bodies of tests/test_gpu_f32.py
import test_gpu_f32 as F32  # noqa: E402

for _name in ("test_logpdf_f32_single_gp", "test_logpdf_f32_gppp_blocks_diag_noise_and_means", "test_cov_and_mean_f32",
              "test_posdef_failure_f32", "test_rand_f32_type_stable_and_close_to_fp64",
              "test_posterior_moments_f32_and_fp64_factor_on_demand", "test_one_output_type_rule_across_the_operator_surface"):
    globals()[_name] = getattr(F32, _name)
del _name


def test_double_is_installed_and_the_product_has_no_cpu_path_of_its_own():
    import stheno_jl_amd as P
    assert isinstance(P.lib.default_context(), np_capi.FakeContext)
    x = np.linspace(0.0, 1.0, 5)
    f = P.atomic(P.GP(P.SEKernel()), P.GPC())
    K = P.prior_cov(f, x)
    assert np.allclose(K, np.exp(-0.5 * (x[:, None] - x[None, :]) ** 2), rtol=0, atol=1e-15)


# the example models / host features of round 2, whose bodies now live in the -m gpu suite (tests/test_gpu_examples.py)
import test_gpu_examples as EX  # noqa: E402

for _name in [n for n in dir(EX) if n.startswith("test_")]:
    globals()[_name] = getattr(EX, _name)
del _name

# round 4: the dense Titsias statements (oracle/titsias_dense.py) against the host mirror's VFE dispatch on the double
import test_gpu_dense_titsias as TD  # noqa: E402

for _name in [n for n in dir(TD) if n.startswith("test_")]:
    globals()[_name] = getattr(TD, _name)
del _name


def test_logpdf_batch_host_mirror_marshalling():
    """Round 6: `logpdf_batch(fxs, ys)` -> sgp_logpdf_batch: the arrays of spec / mean / noise / y pointers the host builds, the
    NaN + info convention for a member that is not positive definite, and the member-by-member route for mixed noise kinds --
    against each member's own `logpdf` and the oracle (the pooled factorisation itself is tests/test_gpu_batch.py's business)."""
    import stheno_jl_amd as P
    from oracle import reference_model as orm
    rng = np.random.default_rng(4)
    F = P.gppp_sum_model()
    fxs, ys, refs = [], [], []
    for b in range(4):
        xs = [np.asfortranarray(rng.standard_normal((2, n))) for n in (30, 25, 40)]
        x = P.BlockData([P.GPPPInput(k, P.ColVecs(v)) for k, v in zip(("f1", "f2", "f3"), xs)])
        y = rng.standard_normal(95)
        fxs.append(F(x, 0.1 + 0.05 * b))
        ys.append(y)
        refs.append(orm.gppp_sum_logpdf(xs, y, 0.1 + 0.05 * b))
    got = P.logpdf_batch(fxs, ys)
    np.testing.assert_allclose(got, refs, rtol=1e-10)
    assert np.array_equal(got, np.array([P.logpdf(fx, y) for fx, y in zip(fxs, ys)]))
    bad = list(fxs)
    bad[1] = F(fxs[1].x, -4.0)
    vals, infos = P.logpdf_batch(bad, ys, return_infos=True)
    assert np.isnan(vals[1]) and infos[1] >= 1 and not infos[[0, 2, 3]].any() and np.array_equal(vals[[0, 2, 3]], got[[0, 2, 3]])
    mixed = list(fxs)
    mixed[2] = F(fxs[2].x, 0.05 + rng.random(95))                      # a diagonal-noise member among scalar ones
    assert np.array_equal(P.logpdf_batch(mixed, ys), np.array([P.logpdf(fx, y) for fx, y in zip(mixed, ys)]))
    assert P.logpdf_batch([], []).shape == (0,)
    with pytest.raises(ValueError):
        P.logpdf_batch(fxs, ys[:2])
