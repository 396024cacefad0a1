"""Occupancy guard: the eval kernels sit right at a register-file step (gradient: 128 VGPRs = 4 waves/SIMD, one more
register drops to 3; likelihood: 72 VGPRs = 7 waves/SIMD, which its 106 SGPRs allow anyway).  A harmless-looking edit has crossed that step before and cost
8 % of the headline rate, so the cross-compiled resource usage is checked on the CPU."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "tamcmc-c-_amd", "csrc")


@pytest.mark.skipif(shutil.which("hipcc") is None and not os.path.exists("/opt/rocm/bin/hipcc"), reason="hipcc not available")
def test_eval_kernel_register_budget():
    r = subprocess.run(["make", "-C", CSRC, "resource-usage"], capture_output=True, text=True, timeout=600)
    txt = r.stdout + r.stderr
    usage = {}
    for m in re.finditer(r"Function Name: (\S+).*?VGPRs: (\d+).*?ScratchSize \[bytes/lane\]: (\d+).*?VGPRs Spill: (\d+)", txt, flags=re.S):
        usage[m.group(1)] = (int(m.group(2)), int(m.group(4)), int(m.group(3)))       # VGPRs, spilled VGPRs, scratch bytes
    grad = [v for k, v in usage.items() if "tamcmc_eval_kernelILb1" in k]
    fwd = [v for k, v in usage.items() if "tamcmc_eval_kernelILb0" in k]
    assert grad and fwd, txt[-2000:]
    assert grad[0][0] <= 128 and grad[0][1] == 0 and grad[0][2] == 0, grad
    assert fwd[0][0] <= 72 and fwd[0][1] == 0 and fwd[0][2] == 0, fwd
