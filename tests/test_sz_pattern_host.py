"""The tile pattern of the Cholesky factor of a programme with independent components (structural zeros:
stheno.jl_amd/csrc/sz_pattern.h, used by capi.hip: sz_pattern) is host-side integer work that must be exact -- a tile
wrongly declared zero silently drops a product.  Compiled for the host with g++ and checked on random programmes
(tests/sz_pattern_host.cpp): against a brute-force boolean statement of the elimination rule, and numerically -- every
non-zero of a plain Cholesky factor lies inside the pattern, and leaving the dead tile products out does not change a bit of
the factor.  The GPU suite (tests/test_gpu_struct_zeros.py) checks the same on the device."""
import os
import subprocess
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))


def test_pattern_against_brute_force_and_numerical_factorisations():
    with tempfile.TemporaryDirectory() as d:
        exe = os.path.join(d, "sz_pattern_host")
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-Wall", "-Werror", os.path.join(HERE, "sz_pattern_host.cpp"), "-o", exe])
        r = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:]
    last = r.stdout.strip().splitlines()[-1].split()
    assert last[0] == "cases" and int(last[1]) > 300 and int(last[3]) > 100 and int(last[5]) == 0, r.stdout[-500:]
