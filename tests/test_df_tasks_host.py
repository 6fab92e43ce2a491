"""The task order of the dataflow factorisation (stheno.jl_amd/csrc/df_tasks.h: task id -> tile, column-major; a batch of
matrices round robin) is integer work that must be exact, and the kernel's freedom from deadlock rests on one property of
it: every input of a task belongs to a task with a smaller id.  Compiled for the host with g++ (tests/df_tasks_host.cpp):
the decode exhaustively for every shape up to 160 tile columns and at the column boundaries of large ones, and a replay of
the kernel's task loop with 1 ... 5000 simulated workgroups, which must always run to completion with the tiles of every
row becoming final in column order -- round 6: also for launches that factor only their first T_f tile columns and merely
update the others (the sharded factorisation's sub-panel launches) and for batches of 2 ... 16 independent matrices."""
import os
import subprocess
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))


def test_task_order_is_exact_and_always_makes_progress():
    with tempfile.TemporaryDirectory() as d:
        exe = os.path.join(d, "df_tasks_host")
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-Wall", "-Werror", os.path.join(HERE, "df_tasks_host.cpp"), "-o", exe])
        r = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:]
    last = r.stdout.strip().splitlines()[-1].split()
    # "shapes S replays R partial P batch B bad X"
    assert last[0] == "shapes" and int(last[1]) > 600 and int(last[3]) > 100 and int(last[5]) > 300 and int(last[7]) > 250 \
        and int(last[9]) == 0, r.stdout[-500:]
