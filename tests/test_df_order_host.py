"""The task order of the dataflow factorisation (stheno.jl_amd/csrc/df_order.h: task id -> tile, column-major) is integer
work that must be exact, and the kernel's freedom from deadlock rests on one property of it: every input of a task belongs
to a task with a smaller id.  Compiled for the host with g++ (tests/df_order_host.cpp): the decode exhaustively for every
shape up to 160 tile columns and at the column boundaries of large ones, and a replay of the schedule with 1 ... 5000
simulated workgroups, which must always run to completion with the tiles of every row becoming final in column order.
Round 4: the same replay for the XCD-affine order (eight in-order queues of tile patches, df_build_queues) over patch shapes,
bordered rows and 8 ... 5000 workgroups, plus "every tile exactly once" and "a tile row lives in one queue"."""
import os
import subprocess
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))


def test_task_order_is_exact_and_always_makes_progress():
    with tempfile.TemporaryDirectory() as d:
        exe = os.path.join(d, "df_order_host")
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-Wall", "-Werror", os.path.join(HERE, "df_order_host.cpp"), "-o", exe])
        r = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:]
    last = r.stdout.strip().splitlines()[-1].split()
    # "shapes S replays R queue replays Q bad B": Q = replays of the XCD-affine queues (df_build_queues), workgroup w serving
    # queue w % 8 in order and the others only once its own is exhausted
    assert last[0] == "shapes" and int(last[1]) > 600 and int(last[3]) > 100 and int(last[6]) > 3000 and int(last[8]) == 0, \
        r.stdout[-500:]
