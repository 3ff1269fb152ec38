"""A short run of tools/gpu_fuzz.py inside the GPU suite: random [Network] shapes, batch sizes and kernel-variant
switches through the C ABI against the fp64 oracle (layer outputs, latents, cosine, arg-max / upright / top-k)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
@pytest.mark.parametrize('seed', [11, 12])
def test_random_networks_and_variants(seed):
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'tools', 'gpu_fuzz.py'), '10', str(seed)],
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    assert '"cases": 10' in out.stdout.splitlines()[-1]
