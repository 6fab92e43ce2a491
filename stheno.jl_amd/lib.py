"""ctypes binding of libsthenomi.so (include/sthenomi.h).

This is the only place the Python host touches native code.  There is deliberately no
CPU fallback: if the HIP library is missing or no gfx950 device is visible, every entry
point raises (`SthenoMIError`), so a silently-eager test run is impossible.
"""
from __future__ import annotations

import ctypes as C
import os
import threading

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "libsthenomi.so")
BENCH_LIB_PATH = os.path.join(_HERE, "csrc", "libsthenomi_bench.so")

# kernel kinds / noise kinds (sthenomi.h enums)
SE, MATERN12, MATERN32, MATERN52, WHITE, CONST = range(6)
NOISE_SCALAR, NOISE_DIAG, NOISE_DENSE = range(3)


class SthenoMIError(RuntimeError):
    pass


class PosDefException(SthenoMIError):
    """Mirrors LinearAlgebra.PosDefException(info) thrown by `cholesky` on the reference path."""

    def __init__(self, info, msg=""):
        super().__init__(f"PosDefException: matrix is not positive definite; "
                         f"Cholesky factorization failed (info={info}). {msg}")
        self.info = info


class sgp_input(C.Structure):
    _fields_ = [("dim", C.c_int64), ("n", C.c_int64), ("ld", C.c_int64),
                ("x", C.POINTER(C.c_double))]


class sgp_term(C.Structure):
    _fields_ = [("kind", C.c_int32), ("row_input", C.c_int32), ("col_input", C.c_int32),
                ("reserved", C.c_int32), ("coef", C.c_double), ("param", C.c_double),
                ("row_scale", C.POINTER(C.c_double)), ("col_scale", C.POINTER(C.c_double))]


class sgp_cov_spec(C.Structure):
    _fields_ = [("n_row_blocks", C.c_int32), ("n_col_blocks", C.c_int32),
                ("row_len", C.POINTER(C.c_int64)), ("col_len", C.POINTER(C.c_int64)),
                ("n_inputs", C.c_int32), ("inputs", C.POINTER(sgp_input)),
                ("term_ptr", C.POINTER(C.c_int32)), ("terms", C.POINTER(sgp_term)),
                ("symmetric", C.c_int32), ("reserved", C.c_int32)]


class sgp_panel_src(C.Structure):
    """a factored column panel, packed (sgp_dev_panel_update_batch)"""
    _fields_ = [("base", C.c_void_p), ("ld", C.c_int64), ("row0", C.c_int64), ("w", C.c_int64)]


class sgp_panel_dst(C.Structure):
    """an owned column panel, packed, and the range of sources applied to it"""
    _fields_ = [("base", C.c_void_p), ("ld", C.c_int64), ("c0", C.c_int64), ("w", C.c_int64),
                ("src_first", C.c_int32), ("src_count", C.c_int32)]


_lib = None
_lib_lock = threading.Lock()

_P = C.c_void_p
_D = C.POINTER(C.c_double)
_SIGS = {
    "sgp_abi_version": (C.c_int, []),
    "sgp_ctx_create": (C.c_int, [C.c_int, C.POINTER(_P)]),
    "sgp_ctx_create_multi": (C.c_int, [C.POINTER(C.c_int), C.c_int, C.POINTER(_P)]),
    "sgp_ctx_ndev": (C.c_int, [_P]),
    "sgp_ctx_transport": (C.c_char_p, [_P]),
    "sgp_ctx_factor_schedule": (C.c_char_p, [_P, C.c_int64]),
    "sgp_ctx_factor_work": (C.c_int, [_P, C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "sgp_cov_spec_suggest_order": (C.c_int, [C.POINTER(sgp_cov_spec), C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "sgp_ctx_multi_stats": (C.c_int, [_P, _D, C.c_int64, C.POINTER(C.c_int64)]),
    "sgp_ctx_multi_owners": (C.c_int, [_P, C.POINTER(C.c_int32), C.c_int64, C.POINTER(C.c_int64)]),
    "sgp_ctx_multi_profile": (C.c_int, [_P, C.c_int]),
    "sgp_ctx_multi_profile_get": (C.c_int, [_P, _D, C.c_int64, C.POINTER(C.c_int64)]),
    "sgp_ctx_destroy": (C.c_int, [_P]),
    "sgp_ctx_trim": (C.c_int, [_P]),
    "sgp_ctx_stage_timing": (C.c_int, [_P, C.c_int]),
    "sgp_ctx_stage_ms": (C.c_int, [_P, _D]),
    "sgp_last_error": (C.c_char_p, []),
    "sgp_kernelmatrix": (C.c_int, [_P, C.POINTER(sgp_cov_spec), _D, C.c_int64]),
    "sgp_kernelmatrix_diag": (C.c_int, [_P, C.POINTER(sgp_cov_spec), _D]),
    "sgp_logpdf": (C.c_int, [_P, C.POINTER(sgp_cov_spec), _D, C.c_int, _D, _D, C.c_int64,
                             C.c_int64, _D]),
    "sgp_logpdf_batch": (C.c_int, [_P, C.c_int, C.POINTER(C.POINTER(sgp_cov_spec)), C.POINTER(_D), C.c_int, C.POINTER(_D),
                                   C.POINTER(_D), _D, C.POINTER(C.c_int)]),
    "sgp_logpdf_f32": (C.c_int, [_P, C.POINTER(sgp_cov_spec), _D, C.c_int, _D, _D, _D]),
    "sgp_kernelmatrix_f32": (C.c_int, [_P, C.POINTER(sgp_cov_spec), C.POINTER(C.c_float), C.c_int64]),
    "sgp_rand_f32": (C.c_int, [_P, C.POINTER(sgp_cov_spec), _D, C.c_int, _D, _D, C.c_int64, C.c_int64,
                               C.POINTER(C.c_float), C.c_int64]),
    "sgp_posterior_mean_var_f32": (C.c_int, [_P, C.POINTER(sgp_cov_spec), _D, C.c_int, _D, _D, C.POINTER(sgp_cov_spec),
                                             C.POINTER(sgp_cov_spec), _D, C.POINTER(C.c_float), C.POINTER(C.c_float)]),
    "sgp_logpdf_grad": (C.c_int, [_P, C.POINTER(sgp_cov_spec), _D, C.c_int, _D, _D, _D, _D, _D, _D, _D, _D]),
    "sgp_logpdf_grad_x": (C.c_int, [_P, C.POINTER(sgp_cov_spec), _D, C.c_int, _D, _D, _D, _D, _D, _D, _D, _D,
                                    C.POINTER(_D)]),
    "sgp_logpdf_grad_xs": (C.c_int, [_P, C.POINTER(sgp_cov_spec), _D, C.c_int, _D, _D, _D, _D, _D, _D, _D, _D,
                                     C.POINTER(_D), C.POINTER(_D)]),
    "sgp_rand": (C.c_int, [_P, C.POINTER(sgp_cov_spec), _D, C.c_int, _D, _D, C.c_int64, C.c_int64,
                           _D, C.c_int64]),
    "sgp_posterior_create": (C.c_int, [_P, C.POINTER(sgp_cov_spec), _D, C.c_int, _D, _D, _D,
                                       C.POINTER(_P)]),
    "sgp_posterior_predict": (C.c_int, [_P, C.POINTER(sgp_cov_spec), C.POINTER(sgp_cov_spec), _D,
                                        _D, _D, _D, C.c_int64]),
    "sgp_posterior_predict_explicit": (C.c_int, [_P, _D, C.c_int64, C.c_int64, _D, _D, C.c_int64, _D, _D, _D, _D, C.c_int64]),
    "sgp_posterior_destroy": (C.c_int, [_P]),
    "sgp_elbo_grad": (C.c_int, [_P, C.POINTER(sgp_cov_spec), C.POINTER(sgp_cov_spec), _D, _D, C.c_int, _D, C.c_int,
                                _D, _D, _D, _D, _D, _D, _D, _D, _D, _D, _D, _D]),
    "sgp_elbo_grad_x": (C.c_int, [_P, C.POINTER(sgp_cov_spec), C.POINTER(sgp_cov_spec), _D, _D, C.c_int, _D, C.c_int,
                                  _D, _D, _D, _D, _D, _D, _D, _D, _D, _D, _D, _D, C.POINTER(_D), C.POINTER(_D)]),
    "sgp_elbo_grad_xs": (C.c_int, [_P, C.POINTER(sgp_cov_spec), C.POINTER(sgp_cov_spec), _D, _D, C.c_int, _D, C.c_int,
                                   _D, _D, _D, _D, _D, _D, _D, _D, _D, _D, _D, _D, C.POINTER(_D), C.POINTER(_D),
                                   C.POINTER(_D), C.POINTER(_D), C.POINTER(_D)]),
    "sgp_kernelmatrix_diag_grad_xs": (C.c_int, [_P, C.POINTER(sgp_cov_spec), _D, _D, _D, C.POINTER(_D), C.POINTER(_D),
                                                C.POINTER(_D)]),
    "sgp_kernelmatrix_diag_grad": (C.c_int, [_P, C.POINTER(sgp_cov_spec), _D, _D, _D]),
    "sgp_kernelmatrix_diag_grad_x": (C.c_int, [_P, C.POINTER(sgp_cov_spec), _D, _D, _D, C.POINTER(_D)]),
    "sgp_elbo": (C.c_int, [_P, C.POINTER(sgp_cov_spec), C.POINTER(sgp_cov_spec), _D, _D, C.c_int,
                           _D, C.c_int, _D, _D, _D]),
    "sgp_sparse_posterior_create": (C.c_int, [_P, C.POINTER(sgp_cov_spec),
                                              C.POINTER(sgp_cov_spec), _D, C.c_int, _D, C.c_int,
                                              _D, _D, C.POINTER(_P)]),
    "sgp_sparse_posterior_predict": (C.c_int, [_P, C.POINTER(sgp_cov_spec),
                                               C.POINTER(sgp_cov_spec), _D, _D, _D, _D, C.c_int64]),
    "sgp_sparse_posterior_destroy": (C.c_int, [_P]),
    "sgp_dspec_create": (C.c_int, [_P, C.POINTER(sgp_cov_spec), C.POINTER(_P)]),
    "sgp_dspec_destroy": (C.c_int, [_P]),
    "sgp_geometry": (C.c_int, [C.c_int64, C.c_int64, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "sgp_dev_logpdf": (C.c_int, [_P, _P, _P, _P, C.c_int, _D, _P, _P, C.c_int64, C.c_int64, _D, _D]),
    "sgp_dev_assemble_cols": (C.c_int, [_P, _P, C.c_int64, C.c_int64, C.c_int64, _P, C.c_int64,
                                        C.c_int64, _P, C.c_int, _D, _P, _P, C.c_int64, C.c_int64,
                                        _P]),
    "sgp_dev_panel_factor": (C.c_int, [_P, _P, C.c_int64, C.c_int64, C.c_int64, C.c_int64, _P, _P,
                                       _P]),
    "sgp_dev_panel_update": (C.c_int, [_P, _P, C.c_int64, C.c_int64, C.c_int64, _P, C.c_int64,
                                       C.c_int64, C.c_int64, C.c_int64, _P]),
    "sgp_dev_panel_update_batch": (C.c_int, [_P, _P, C.c_int, _P, C.c_int, C.c_int64, _P]),
    "sgp_dev_rowsumsq": (C.c_int, [_P, _P, C.c_int64, C.c_int64, C.c_int64, _P, _P]),
    "sgp_elbo_part_len": (C.c_int, [C.c_int64, C.POINTER(C.c_int64)]),
    "sgp_dev_elbo_partial": (C.c_int, [_P, C.POINTER(sgp_cov_spec), C.POINTER(sgp_cov_spec), _D, _D, C.c_int, _D,
                                       C.c_int, _D, _D, _P, C.c_int64]),
    "sgp_dev_elbo_finish": (C.c_int, [_P, C.c_int64, C.c_int64, _P, _D]),
    "sgp_dev_assemble_cross_rows": (C.c_int, [_P, _P, C.c_int64, C.c_int64, _P, C.c_int64, C.c_int64, _P]),
    "sgp_dev_rows_dot": (C.c_int, [_P, _P, C.c_int64, C.c_int64, C.c_int64, _P, _P, _P, _P]),
    "sgp_dev_rows_gram": (C.c_int, [_P, _P, C.c_int64, C.c_int64, C.c_int64, _P, C.c_int64, _P]),
}
# include/sthenomi_bench.h: micro-benchmark / diagnosis / test hooks, exported by libsthenomi_bench.so (round 6), NOT by the product
# library -- bench.py, tools/ and the GPU tests reach them through bench_lib() / Context.bench
_SIGS_BENCH = {
    "sgp_bench_df_fallbacks": (C.c_int, [_P, C.POINTER(C.c_int64)]),
    "sgp_bench_multi_fault": (C.c_int, [_P, C.c_int, C.c_int64]),
    "sgp_bench_multi_broken": (C.c_int, [_P, C.POINTER(C.c_int)]),
    "sgp_bench_multi_profile_pieces": (C.c_int, [_P, _D, C.c_int64, C.POINTER(C.c_int64)]),
    "sgp_bench_multi_stall": (C.c_int, [_P, C.c_int, C.c_int64, C.c_double]),
    "sgp_bench_mfma_f64": (C.c_int, [_P, C.c_int, _D, _D]),
    "sgp_bench_hbm": (C.c_int, [_P, C.c_int64, C.c_int, _D, _D]),
    "sgp_bench_potrf": (C.c_int, [_P, C.c_int, _D, C.POINTER(C.c_longlong)]),
    "sgp_bench_cumask": (C.c_int, [_P, C.POINTER(C.c_uint32), C.c_int, C.c_int, C.POINTER(C.c_uint)]),
    "sgp_bench_potrf_contended": (C.c_int, [_P, C.c_int64, C.c_int64, C.c_int, C.c_int, _D, C.POINTER(C.c_longlong), C.POINTER(C.c_int)]),
    "sgp_bench_gemm_stamps": (C.c_int, [_P, C.c_int64, C.c_int64, C.POINTER(C.c_longlong), C.c_int64, C.POINTER(C.c_int64)]),
    "sgp_bench_gemm": (C.c_int, [_P, C.c_int64, C.c_int64, C.c_int64, C.c_int, C.c_int, _D, _D]),
}


def exported_symbols():
    """Names include/sthenomi.h declares (used by the CPU-side symbol test)."""
    return sorted(_SIGS)


def bench_symbols():
    """Names include/sthenomi_bench.h declares: the entry points of libsthenomi_bench.so."""
    return sorted(_SIGS_BENCH)


_bench = None


def bench_lib():
    """dlopen libsthenomi_bench.so (micro-benchmarks, diagnosis and test hooks: include/sthenomi_bench.h).  It links against
    the product library and works on contexts created there; nothing on a product path loads it."""
    global _bench
    load()
    with _lib_lock:
        if _bench is not None:
            return _bench
        if not os.path.exists(BENCH_LIB_PATH):
            raise SthenoMIError(f"{BENCH_LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'`")
        lib = C.CDLL(BENCH_LIB_PATH, mode=C.RTLD_GLOBAL)
        for name, (res, args) in _SIGS_BENCH.items():
            fn = getattr(lib, name)
            fn.restype = res
            fn.argtypes = args
        _bench = lib
        return lib


def load():
    """dlopen libsthenomi.so (after torch, so both share one HIP runtime) and type its symbols."""
    global _lib
    with _lib_lock:
        if _lib is not None:
            return _lib
        if not os.path.exists(LIB_PATH):
            raise SthenoMIError(
                f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; "
                f"g.build()'` (hipcc --offload-arch=gfx950).  There is no CPU fallback.")
        try:  # torch bundles its own libamdhip64 (SONAME libamdhip64.so.7): load it first
            import torch  # noqa: F401
        except Exception:  # pragma: no cover - torch is optional for the pure C-ABI
            pass
        lib = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
        for name, (res, args) in _SIGS.items():
            if not hasattr(lib, name) and os.environ.get("SGP_ALLOW_MISSING_SYMBOLS"):
                continue        # (A/B runs against an older build of the library: tools/r05_call7.sh)
            fn = getattr(lib, name)
            fn.restype = res
            fn.argtypes = args
        _lib = lib
        return lib


def last_error():
    return load().sgp_last_error().decode("utf-8", "replace")


def check(rc, what=""):
    if rc == 0:
        return
    msg = last_error()
    if rc > 0:
        raise PosDefException(rc, msg)
    raise SthenoMIError(f"{what} failed (rc={rc}): {msg}")


def dptr(a):
    """double* of a numpy array (or NULL)."""
    if a is None:
        return C.cast(None, _D)
    return a.ctypes.data_as(_D)


class Context:
    """One sgp_ctx (one GPU, one stream).  `default_context()` gives a process-wide one."""

    def __init__(self, device=0, devices=None):
        """device: one GPU.  devices=[...]: a multi-GPU context (sgp_ctx_create_multi): logpdf, rand, the posterior, the
        gradients, the covariance entry points, the ELBO and its gradients are sharded over the listed GPUs inside the
        library (include/sthenomi.h lists what is not)."""
        lib = load()
        h = _P()
        if devices is not None:
            arr = (C.c_int * len(devices))(*[int(d) for d in devices])
            check(lib.sgp_ctx_create_multi(arr, len(devices), C.byref(h)), "sgp_ctx_create_multi")
            device = int(devices[0])
        else:
            check(lib.sgp_ctx_create(int(device), C.byref(h)), "sgp_ctx_create")
        self.handle = h
        self.device = device
        self.lib = lib

    @property
    def bench(self):
        """libsthenomi_bench.so (sthenomi_bench.h): `ctx.bench.sgp_bench_*(ctx.handle, ...)`"""
        return bench_lib()

    @property
    def ndev(self):
        return int(self.lib.sgp_ctx_ndev(self.handle))

    @property
    def transport(self):
        return self.lib.sgp_ctx_transport(self.handle).decode()

    def factor_schedule(self, N):
        """Which schedule the blocked Cholesky of an N-point covariance runs on this context."""
        return self.lib.sgp_ctx_factor_schedule(self.handle, int(N)).decode()

    def factor_work(self):
        """(executed, dense) tile products of the last factorisation's contractions: they differ when the model has
        independent components whose exact zero blocks the factorisation skipped (sthenomi.h: sgp_ctx_factor_work)."""
        e, d = C.c_double(), C.c_double()
        check(self.lib.sgp_ctx_factor_work(self.handle, C.byref(e), C.byref(d)), "sgp_ctx_factor_work")
        return e.value, d.value

    def close(self):
        if getattr(self, "handle", None):
            self.lib.sgp_ctx_destroy(self.handle)
            self.handle = None

    def __del__(self):  # pragma: no cover
        try:
            self.close()
        except Exception:
            pass


_default_ctx = None


def set_default_context(ctx):
    """Route the host mirror's calls (logpdf, posterior, ...) through `ctx`; returns the previous one.
    With a multi-GPU context (Context(devices=[...])) logpdf is sharded over its GPUs."""
    global _default_ctx
    prev, _default_ctx = _default_ctx, ctx
    return prev


def default_context():
    global _default_ctx
    if _default_ctx is None:
        devs = os.environ.get("SGP_DEVICES")         # e.g. "0,1,2,3,4,5,6,7": one process, all GPUs
        if devs:
            _default_ctx = Context(devices=[int(d) for d in devs.split(",") if d.strip() != ""])
            return _default_ctx
        dev = int(os.environ.get("LOCAL_RANK", "0"))
        try:
            import torch
            if torch.cuda.is_available() and dev >= torch.cuda.device_count():
                dev = 0
        except Exception:
            pass
        _default_ctx = Context(dev)
    return _default_ctx


class Spec:
    """Owns the numpy buffers and ctypes arrays behind one sgp_cov_spec.

    row_len / col_len : block lengths
    inputs            : list of (D x n) float64 Fortran-ordered arrays (ColVecs layout)
    pairs             : dict (I, J) -> list of term tuples
                        (kind, row_input, col_input, coef, param, row_scale|None, col_scale|None)
    """

    def __init__(self, row_len, col_len, inputs, pairs, symmetric):
        self.row_len = np.asarray(row_len, dtype=np.int64)
        self.col_len = np.asarray(col_len, dtype=np.int64)
        self.N = int(self.row_len.sum())
        self.M = int(self.col_len.sum())
        self.symmetric = bool(symmetric)
        self._keep = []
        self.inputs = []
        for x in inputs:
            x = np.asarray(x, dtype=np.float64)
            if x.ndim == 1:
                x = x.reshape(1, -1)
            x = np.asfortranarray(x)
            self.inputs.append(x)
        nrb, ncb = len(self.row_len), len(self.col_len)
        self._in_arr = (sgp_input * max(1, len(self.inputs)))()
        for k, x in enumerate(self.inputs):
            self._in_arr[k].dim = x.shape[0]
            self._in_arr[k].n = x.shape[1]
            self._in_arr[k].ld = x.shape[0]
            self._in_arr[k].x = dptr(x)
        term_ptr = [0]
        terms = []
        for I in range(nrb):
            for J in range(ncb):
                terms.extend(pairs.get((I, J), []))
                term_ptr.append(len(terms))
        self.n_terms = len(terms)
        self._term_ptr = np.asarray(term_ptr, dtype=np.int32)
        self._terms = (sgp_term * max(1, len(terms)))()
        # identity of each term's scale vectors (the flattener's own merge key): a term of pair (I, J)
        # with scales (rs, cs) is mirrored in pair (J, I) by the term with (cs, rs)
        self.term_scale_ids = [(None if rs is None else id(rs), None if cs is None else id(cs))
                               for (_, _, _, _, _, rs, cs) in terms]
        self.term_row_scale = [rs for (_, _, _, _, _, rs, _) in terms]   # the flattener's own vectors (with factors)
        self.term_col_scale = [cs for (_, _, _, _, _, _, cs) in terms]
        self._keep.extend(terms)
        for k, (kind, ri, ci, coef, param, rs, cs) in enumerate(terms):
            t = self._terms[k]
            t.kind, t.row_input, t.col_input, t.reserved = int(kind), int(ri), int(ci), 0
            t.coef, t.param = float(coef), float(param)
            if rs is not None:
                rs = np.ascontiguousarray(rs, dtype=np.float64)
                self._keep.append(rs)
            if cs is not None:
                cs = np.ascontiguousarray(cs, dtype=np.float64)
                self._keep.append(cs)
            t.row_scale = dptr(rs)
            t.col_scale = dptr(cs)
        s = sgp_cov_spec()
        s.n_row_blocks, s.n_col_blocks = nrb, ncb
        s.row_len = self.row_len.ctypes.data_as(C.POINTER(C.c_int64))
        s.col_len = self.col_len.ctypes.data_as(C.POINTER(C.c_int64))
        s.n_inputs = len(self.inputs)
        s.inputs = self._in_arr
        s.term_ptr = self._term_ptr.ctypes.data_as(C.POINTER(C.c_int32))
        s.terms = self._terms
        s.symmetric = 1 if symmetric else 0
        s.reserved = 0
        self.c = s

    def ref(self):
        return C.byref(self.c)

    def f32_supported(self):
        """The fp32 device kernels (csrc/f32.hip: assemble_f32) take input dimension <= 16 and, per block pair,
        (number of terms) x (dimension rounded up to a power of two) <= 64; anything else runs on the fp64 path."""
        tp = self._term_ptr
        for p in range(len(tp) - 1):
            t0, t1 = int(tp[p]), int(tp[p + 1])
            if t1 == t0:
                continue
            d = max(self.inputs[int(self._terms[t].row_input)].shape[0] for t in range(t0, t1))
            dmax = 1
            while dmax < d:
                dmax *= 2
            if dmax > 16 or (t1 - t0) * dmax > 64:
                return False
        return True


def _noise_args(noise, N):
    """(kind, buffer) for a FiniteGP noise: scalar -> s2*I, vector -> Diagonal, matrix -> dense."""
    a = np.asarray(noise, dtype=np.float64)
    if a.ndim == 0:
        return NOISE_SCALAR, np.array([float(a)], dtype=np.float64)
    if a.ndim == 1:
        if a.shape[0] != N:
            raise ValueError("diagonal noise has the wrong length")
        return NOISE_DIAG, np.ascontiguousarray(a)
    if a.shape != (N, N):
        raise ValueError("dense noise has the wrong shape")
    return NOISE_DENSE, np.asfortranarray(a)
