"""The bodies of tests/test_gpu_random_programmes.py WITHOUT a GPU: libsthenomi.so replaced by the NumPy double of its C-ABI
(tests/np_capi.py).  Proves the host side on random programmes at the sizes the device suite uses (a few hundred to ~1500
points, ragged blocks) and that the test bodies themselves are sound; says nothing about the HIP kernels -- the same bodies
run against the real library under `-m gpu`."""
import pytest

import np_capi
import oracle.kernelfunctions as okf
import test_gpu_random_programmes as T


@pytest.fixture(autouse=True)
def _numpy_double_and_direct_distances(monkeypatch):
    np_capi.install(monkeypatch)
    orig = okf.pairwise_sqeuclidean
    monkeypatch.setattr(okf, "pairwise_sqeuclidean", lambda X, Y=None, faithful=True: orig(X, Y, False))


# (test_random_programme_structural_zeros_change_no_bit compares schedules of the real library: device only)
for _n in [n for n in dir(T) if n.startswith("test_") and "structural_zeros" not in n]:
    globals()[_n] = getattr(T, _n)
del _n
