"""The workgroup -> tile enumeration of the GEMM launches (stheno.jl_amd/csrc/tilemap.h) is integer work that must be
exact: compiled for the host with g++ and checked exhaustively over launch shapes (tests/tilemap_host.cpp) -- every
live tile once, nothing dead, tile rows bound to their XCD, id 0 = tile (0, 0) (the fused update + potrf_diag launch
of the blocked Cholesky depends on it).  The GPU suites only see the shapes their problem sizes produce."""
import os
import subprocess
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))


def test_tile_enumeration_is_exact_for_every_launch_shape():
    with tempfile.TemporaryDirectory() as d:
        exe = os.path.join(d, "tilemap_host")
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-Wall", "-Werror", os.path.join(HERE, "tilemap_host.cpp"), "-o", exe])
        r = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:]
    last = r.stdout.strip().splitlines()[-1].split()
    assert last[0] == "shapes" and int(last[1]) > 20000 and int(last[3]) == 0, r.stdout[-500:]
    # dead workgroups cost dispatcher time.  What is left: up to 7 tile rows of padding (owned rows come in eights per
    # XCD) -- a quarter of a 65-row launch, below 3 % from 512 tile rows (N = 65536) on
    assert float(last[5]) < 0.25 and float(last[11]) < 0.03, r.stdout[-500:]
