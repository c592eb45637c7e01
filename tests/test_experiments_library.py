"""The measurement-only forms of the threshold + corner pass live in librcc_hip_exp.so (make -C csrc EXPERIMENTS=1), not in the
product library: the two-kernel variant 3 (k_dense_runs.hip) and the gang form of k_dense_wave.  Their
bit-identity is checked against THAT library in a child process (one HIP library per process), and the child's digests of the
default form must equal the product library's on the same frames -- so "identical to the experiments library's default form"
means "identical to what ships"."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXP = os.path.join(ROOT, "robot_camera_calibration_amd", "librcc_hip_exp.so")


@pytest.mark.gpu
def test_experiment_forms_bit_identical_to_the_product():
    if not os.path.exists(EXP):
        pytest.skip("librcc_hip_exp.so has not been built (python __graft_entry__.py)")
    import torch
    assert torch.cuda.is_available()
    from robot_camera_calibration_amd import abi, api, synth
    from tests import exp_variants_check as X
    assert os.path.basename(api.library_path()) == "librcc_hip.so"
    mine = {(k, w, h, n): X.digest_default(torch, abi, api, synth, k, w, h, n, False) for k, w, h, n in X.GEOMS}
    env = dict(os.environ, RCC_LIBRARY=EXP)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "exp_variants_check.py")], capture_output=True, text=True, env=env, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    theirs = {}
    for line in out.stdout.splitlines():
        if line.startswith("DIGEST "):
            _, k, w, h, n, d = line.split()
            theirs[(k, int(w), int(h), int(n))] = d
    assert theirs == mine
