"""-m gpu: the C-ABI driven from C (tests/capi_logpdf.c) -- no Python between the caller and libsthenomi.so, as a Julia
`ccall` would be (round-3 verdict: the calling convention had only ever been exercised through the ctypes mirror).  This
wrapper compiles the program against include/sthenomi.h (the product header alone), writes the c1 inputs and the committed
CPU golden into one binary file and runs it; the program builds the sgp_cov_spec itself."""
import math
import os
import struct
import subprocess

import numpy as np
import pytest

import bench_configs as bc
import stheno_jl_amd as P

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def test_c_program_runs_logpdf_and_posterior_against_the_c1_golden(tmp_path):
    exe = str(tmp_path / "capi_logpdf")
    subprocess.check_call(["gcc", "-std=c99", "-pedantic", "-Wall", "-Werror", "-O1", "-o", exe,
                           os.path.join(HERE, "capi_logpdf.c"), "-ldl", "-lm"])
    kind, N, D = bc.CONFIGS["c1"]
    assert kind == "se"
    X, y = bc.make_inputs(N, D)
    g = bc.golden("c1")
    ls = math.sqrt(D)
    XS = bc.xs_points(D)
    NS = XS.shape[1]
    case = str(tmp_path / "c1.bin")
    with open(case, "wb") as fh:
        fh.write(struct.pack("<qqq", N, D, NS))
        fh.write(struct.pack("<dd", bc.SIGMA2, g["logpdf"]))
        for a in (np.asfortranarray(X / ls).ravel(order="F"), y, np.asfortranarray(XS / ls).ravel(order="F"),
                  np.asarray(g["post_mean"]), np.asarray(g["post_var"])):
            fh.write(np.ascontiguousarray(a, dtype="<f8").tobytes())
    r = subprocess.run([exe, P.lib.LIB_PATH, case], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert r.stdout.strip().splitlines()[-1] == "OK", r.stdout
    assert "not positive definite" in r.stdout
