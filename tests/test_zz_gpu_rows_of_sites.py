"""The GPU twin of test_emu_parity.test_rows_of_seven_and_eight_snp_sites (the file's name puts it behind the other GPU tests: it was
written when the round's GPU time was spent and runs on the device for the first time in the round's closing run)."""
import pytest

import harness
from test_emu_parity import rows_of_sites_case


@pytest.mark.gpu
def test_rows_of_seven_and_eight_snp_sites_on_the_device():
    assert rows_of_sites_case(harness.GpuBackend) == 800
