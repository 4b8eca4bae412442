"""The block-level backward kernels of a frame batch (blend_bwd_mfma_kernel with the forward's cull flags,
blend_bwd_sets_kernel) are what SPLAT_BWD_QUARTERS=0 selects; by default a batch runs the quarter-list kernels.  The switch
is read once per process, so the oracle tests of the batch path run again in a child process with the block-level kernels."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_frame_batch_oracle_tests_with_block_level_backward():
    env = dict(os.environ, SPLAT_BWD_QUARTERS="0", PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""))
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_gpu_frames_oracle.py"), "-q", "-x", "-m", "gpu",
                        "-k", "render_against or render_sets_against or per_frame_cameras or wide", "-p", "no:cacheprovider"],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert " passed" in r.stdout
