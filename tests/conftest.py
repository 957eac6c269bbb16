import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # Build the product library and the oracle once, if they are not there yet (seconds).
    if not os.path.exists(os.path.join(ROOT, "kmersgwas_amd", "lib", "libkgwas.so")):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "kmersgwas_amd", "csrc")], stdout=subprocess.DEVNULL)
    if not os.path.exists(os.path.join(ROOT, "oracle", "liboracle.so")):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")], stdout=subprocess.DEVNULL)


@pytest.fixture(scope="session")
def have_gpu():
    import kmersgwas_amd as kg
    return kg.device_count() > 0
