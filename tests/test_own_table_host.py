"""The panel ownership table of the sharded factorisation (stheno.jl_amd/csrc/own_table.h, used by multi.hip: make_geometry)
is host-side work: compiled for the host with g++ and checked in tests/own_table_host.cpp -- one panel per rank and round,
equal costs give the cyclic deal, never worse than the cyclic deal, and the north-star model's per-rank loads (from the
symbolic tile pattern, sz_pattern.h) come out within 3 % of their mean where the cyclic deal is off by -7 % / +10 %.  The GPU
suite (tests/test_gpu_multi.py) checks that a non-cyclic table gives the cyclic deal's bits."""
import os
import subprocess
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))


def test_owner_table_properties_and_the_north_star_balance():
    with tempfile.TemporaryDirectory() as d:
        exe = os.path.join(d, "own_table_host")
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-Wall", "-Werror", os.path.join(HERE, "own_table_host.cpp"), "-o", exe])
        r = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:]
    last = r.stdout.strip().splitlines()[-1].split()
    assert last[0] == "cases" and int(last[1]) > 2000 and int(last[3]) == 0, r.stdout[-500:]
