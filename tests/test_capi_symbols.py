"""CPU-side checks of the drop-in boundary: the in-tree libsthenomi.so loads, exports every
symbol include/sthenomi.h declares, and refuses to run without a gfx950 device (no CPU path)."""
import ctypes
import os
import re

import pytest

import stheno_jl_amd as P

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    txt = open(os.path.join(ROOT, "include", "sthenomi.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(sgp_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    lib = P.lib.load()
    syms = header_symbols()
    assert len(syms) >= 25
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in sthenomi.h but not exported"
    # and the Python binding types exactly that set
    assert sorted(P.lib.exported_symbols()) == syms
    assert lib.sgp_abi_version() == 1


def test_geometry_is_pure_host_arithmetic():
    lib = P.lib.load()
    n, m = ctypes.c_int64(), ctypes.c_int64()
    for N, S, en, em in [(1, 1, 128, 256), (128, 0, 128, 128), (129, 3, 256, 384), (65536, 1, 65536, 65664)]:
        assert lib.sgp_geometry(N, S, ctypes.byref(n), ctypes.byref(m)) == 0
        assert (n.value, m.value) == (en, em)


def test_no_cpu_fallback():
    """Without a HIP device every product entry point fails loudly."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present; the loud-failure path is exercised on the CPU builder")
    with pytest.raises(P.SthenoMIError) as ei:
        P.lib.Context(0)
    assert "no HIP device" in str(ei.value) or "no CPU path" in str(ei.value)
    f = P.atomic(P.GP(P.SEKernel()), P.GPC())
    import numpy as np
    with pytest.raises(P.SthenoMIError):
        P.logpdf(f(np.arange(4.0), 0.1), np.zeros(4))


def test_spec_struct_layout_matches_header():
    """sizeof checks guard the ctypes mirror of sgp_input / sgp_term / sgp_cov_spec."""
    assert ctypes.sizeof(P.lib.sgp_input) == 32
    assert ctypes.sizeof(P.lib.sgp_term) == 48
    assert ctypes.sizeof(P.lib.sgp_cov_spec) == 64
