"""Randomised hazard hunts (tools/fuzz_blocks.py, tools/fuzz_conv.py, tools/fuzz_bwd.py) at a size that runs in seconds: random shapes,
dilations, flags; every residual-block mode twice (bit-identical), against the exact fp32 kernel; every conv arithmetic
against a float64 convolution; the fused backward kernels of both arithmetic modes against each other.  Round 1 ran 9 000 block and 8 600 conv cases clean."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("tool,cases", [("fuzz_blocks.py", 250), ("fuzz_conv.py", 400), ("fuzz_bwd.py", 60)])
def test_randomised_shapes(tool, cases):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", tool), str(cases), "11"], capture_output=True, text=True,
                       timeout=600)
    print(r.stdout[-2000:])
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
