"""The sweep of tests/stress_emu.py through the C ABI on the device (`--gpu`): random graph shapes x error / N rates x read lengths x
region offsets x four kinds of position hints, every record (and, with --stream, every score word, call and phase flag) compared
with the oracle's.  Two seeds of each kind here -- the ranges of record are run by hand."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("args", [["7300", "7302"], ["--stream", "7310", "7312"]], ids=["align", "stream"])
def test_stress_sweep_on_the_device(args):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "stress_emu.py"), "--gpu"] + args, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "ok" in r.stdout and "FAIL" not in r.stdout
