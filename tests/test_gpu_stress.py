"""A short run of the adversarial differential stress tools (tools/stress_lk_parity.py,
tools/stress_parity.py): binary / noise / stripe textures, extreme thresholds, degenerate
triangulation and PnP scenes -- zero mismatches against the oracle allowed."""
import os
import subprocess
import sys

import pytest

import conftest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("tool,seeds", [("stress_lk_parity.py", "4"), ("stress_lk_parity.py", "4 sse2"), ("stress_parity.py", "3")])
def test_adversarial_stress(tool, seeds):
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    r = subprocess.run([sys.executable, os.path.join(conftest.ROOT, "tools", tool)] + seeds.split(), capture_output=True, timeout=900)
    out = r.stdout.decode()
    assert r.returncode == 0 and "stress result: OK" in out, out[-3000:] + r.stderr.decode()[-2000:]
