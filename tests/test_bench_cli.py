"""bench.py --gpus N must use N GPUs or refuse (round-2 verdict: `--gpus 8` without torch.distributed.run silently ran
and reported 1 GPU).  CPU part: the device-resolution rule and the exit code on a box without enough GPUs; GPU part
(-m gpu): the in-process multi-GPU code path the plain `python bench.py --gpus N` takes, driven on the 1-GPU box with
two loopback ranks, checked through the printed JSON line."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(*args, env=None):
    e = dict(os.environ)
    e.pop("WORLD_SIZE", None)
    e.pop("RANK", None)
    e.update(env or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], capture_output=True, text=True, env=e,
                          timeout=600)


def test_resolve_devices_never_falls_back():
    import bench
    assert bench.resolve_devices(1, "", 1, 1) == [0]
    assert bench.resolve_devices(8, "", 8, 1) == list(range(8))
    assert bench.resolve_devices(2, "0,0", 1, 1) == [0, 0]          # loopback ranks: the 1-GPU test hook
    assert bench.resolve_devices(2, "", 8, 2) is None               # under torch.distributed.run: LOCAL_RANK decides
    for bad in ((2, "", 1, 1), (8, "", 4, 1), (2, "0,1", 1, 1), (2, "0", 4, 1), (1, "", 0, 1), (4, "", 8, 2),
                (2, "0,1", 8, 2)):
        with pytest.raises(SystemExit) as e:
            bench.resolve_devices(*bad)
        assert e.value.code not in (0, None), bad


@pytest.mark.parametrize("schedule", [None, "hybrid"])
@pytest.mark.parametrize("N", [8192, 16384, 32768, 65536, 40000])
def test_update_launch_model_covers_the_trailing_matrix(N, schedule):
    """bench.update_launches mirrors the driver's panel rule (it prices `roofline.algorithmic_bytes_per_launch_avg`): whatever
    the schedule, every column of the factor must receive exactly the columns to its left that lie in EARLIER panels of its
    own level as contraction depth -- for the look-ahead schedules (one level: outer panels of W): column c gets K = the
    first column of its panel; for the serial-deep schedule the halving inside a panel adds the panel's own earlier halves."""
    import bench
    n_pad = (N + 127) // 128 * 128
    ls = bench.update_launches(N, schedule)
    got = [0] * (n_pad // 128)
    for m, nc, k in ls:
        assert m % 128 == 0 and nc % 128 == 0 and k % 128 == 0 and nc > 0 and k > 0 and m >= nc
        c0 = n_pad + 128 - m                      # the launch's first column: its rows run to the bordered row
        for t in range(c0 // 128, (c0 + nc) // 128):
            got[t] += k
    deep = schedule is None and n_pad >= 65536
    W = 2048 if schedule == "hybrid" else (n_pad if n_pad <= 4096 else 1024 if n_pad <= 8192 else 4096 if deep
                                          else 1024 if n_pad >= 32768 else 512)
    for t, g in enumerate(got):
        c = t * 128
        want = c // W * W                                                    # everything left of its outer panel
        if deep:                                                             # + the earlier 1024-blocks of its own panel
            want += (c % W) // 1024 * 1024
        assert g == want, (N, schedule, t, g, want)
    if schedule == "hybrid":
        assert all(k == 2048 for _, _, k in ls[:-2])
        assert len(ls) == 2 * ((n_pad + 2047) // 2048 - 1) - (1 if (n_pad + 2047) // 2048 >= 2 else 0)


@pytest.mark.parametrize("gpus", ["2", "8"])
def test_gpus_2_exits_nonzero_without_two_gpus(gpus):
    import torch
    if torch.cuda.is_available() and torch.cuda.device_count() >= int(gpus):
        pytest.skip("this box has that many GPUs")
    r = _bench("--gpus", gpus, "--config", "c1", "--steps", "1", "--warmup", "0", "--cpu-sample", "0")
    assert r.returncode != 0
    assert r.stdout.strip() == ""                     # no result line for a run that did not happen
    assert "GPU" in r.stderr or "HIP device" in r.stderr


@pytest.mark.gpu
def test_inprocess_multi_gpu_line_on_loopback_ranks():
    r = _bench("--gpus", "2", "--devices", "0,0", "--config", "c1", "--steps", "2", "--warmup", "1", "--cpu-sample", "0")
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    # two loopback ranks on ONE physical GPU: n_gpus counts GPUs, `ranks` the ranks, and the line says it is a loopback run
    assert line["n_gpus"] == 1 and line["ranks"] == 2 and line["loopback"] is True
    assert line["metric"] == "logpdf_per_sec" and line["scaling"] == "strong"
    mg = line["multi_gpu"]
    assert mg["ranks"] == 2 and mg["devices"] == [0, 0] and mg["transport"] == "loopback"
    assert len(mg["per_rank"]) == 2 and sum(p["panels_factored"] for p in mg["per_rank"]) == mg["panels"]
    assert line["parity_rel"] is not None and line["parity_rel"] < 1e-10
    assert line["roofline"]["bound"] == "mfma" and len(line["roofline"]["per_rank_update_tflops"]) == 2


@pytest.mark.gpu
def test_transport_probe_code_path_on_loopback_ranks():
    """On a node with > 2 GPUs bench.py times one untimed logpdf per panel transport and keeps the faster one; the
    probe's code path runs here with loopback ranks (both candidates resolve to same-device copies)."""
    r = _bench("--gpus", "3", "--devices", "0,0,0", "--config", "c1", "--steps", "1", "--warmup", "1", "--cpu-sample", "0",
               env={"SGP_BENCH_FORCE_PROBE": "1", "SGP_MULTI_PANEL": "512"})
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    probe = line["multi_gpu"]["transport_probe_ms"]
    assert set(probe) == {"p2p", "rccl"} and all(v > 0 for v in probe.values())
    assert line["n_gpus"] == 1 and line["ranks"] == 3 and line["parity_rel"] < 1e-10


@pytest.mark.gpu
@pytest.mark.parametrize("gpus", ["2", "8"])
def test_gpus_2_refuses_on_the_one_gpu_box(gpus):
    """(round 6: and `--gpus 8`, the command the driver issues on a node, on the 1-GPU box: exit != 0, no JSON line)"""
    import torch
    if torch.cuda.device_count() >= int(gpus):
        pytest.skip("this box has that many GPUs")
    r = _bench("--gpus", gpus, "--config", "c1", "--steps", "1", "--warmup", "0", "--cpu-sample", "0")
    assert r.returncode != 0 and r.stdout.strip() == "" and "refusing" in r.stderr
