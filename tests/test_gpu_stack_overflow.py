"""The two-level traversal stack of the render launches (dev_intersect.hpp::stack_put / stack_get, trc_abi.hip::
plan_launch_lds): by default a lane keeps 16 entries in LDS and only deeper ones in the workgroup's global rows, so the
small test scenes never leave LDS.  Here the same parity tests run with 1 and 3 LDS entries per lane
(TRC_STACK_LDS_LEVELS, read once per process, hence the child processes): nearly every pending sibling then lives in the
overflow rows, and the frames must still equal the oracle's bit for bit.
"""
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu

SELECTION = [
    "tests/test_gpu_parity.py::test_render_bit_exact",
    "tests/test_gpu_lbvh.py::test_render_through_device_tree",
    "tests/test_gpu_volume.py::test_homogeneous_medium_in_mesh_and_cube",
    "tests/test_gpu_fuzz.py::test_generated_scene",
    "tests/test_gpu_fullsize.py::test_one_sample_per_launch_like_the_reference",
]


@pytest.mark.parametrize("levels", [1, 3])
def test_parity_with_the_stack_mostly_in_the_overflow_rows(gpu, levels):
    env = dict(os.environ, TRC_STACK_LDS_LEVELS=str(levels))
    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-m", "gpu", "-p", "no:cacheprovider"] + SELECTION,
                       cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=1500)
    assert r.returncode == 0, r.stdout[-3000:]
    assert " passed" in r.stdout and "failed" not in r.stdout, r.stdout[-2000:]
