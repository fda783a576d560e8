"""The `pytorch_lightning` branch of the mirror modules (ModelBase derives from pl.LightningModule when the package imports;
`_raw_optimizers` unwraps LightningOptimizer so that grad_mul / grad_scale reach the optimizer whose step() runs): executed
against the reference's training_step recordings with a stand-in package (tools/debug/pl_stub: pytorch-lightning is not
installed in this image), in a subprocess so that the stand-in never leaks into the other tests."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_lightning_module_branch_matches_reference_recordings():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "debug", "pl_stub_check.py")], cwd=ROOT, capture_output=True,
                       text=True, timeout=600)
    out = r.stdout + r.stderr
    assert r.returncode == 0, out[-3000:]
    assert "PL_BRANCH_OK" in out and "MISMATCH" not in out, out[-3000:]
