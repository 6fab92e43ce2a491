"""The in-tree libraries travel to the GPU box as built (they are git-ignored, not gpurun-ignored): a library older than its
sources would be measured and sha-stamped in place of the code that is committed (round 6: it happened once -- an edit of
capi.hip without a rebuild before a collection).  `make -q` says whether the libraries are up to date with every source and
header they are built from; a tree without built libraries (a fresh clone before `__graft_entry__.build()`) has nothing to
check."""
import os
import subprocess

import pytest

CSRC = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "stheno.jl_amd", "csrc")


def test_the_built_libraries_are_up_to_date_with_their_sources():
    if not (os.path.exists(os.path.join(CSRC, "libsthenomi.so")) and os.path.exists(os.path.join(CSRC, "capi.o"))):
        pytest.skip("no in-tree build to check")
    r = subprocess.run(["make", "-q", "-C", CSRC, "all"], capture_output=True, text=True)
    assert r.returncode == 0, ("stheno.jl_amd/csrc: a source or header is newer than the built library -- run "
                               "`python -c 'import __graft_entry__ as g; g.build()'` (make -q: rc %d)\n%s" % (r.returncode, r.stdout[-500:]))
