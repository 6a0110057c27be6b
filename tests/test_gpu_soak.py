"""Randomised consistency soak (tools/soak.py): culled == strict scan, fused == drop-in loss,
carried-over target scan == full evaluation, over random ragged shapes and scales."""
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("seed", [1, 2])
def test_soak(seed):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "soak.py"), str(seed), "80"],
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "0 mismatches" in r.stdout
