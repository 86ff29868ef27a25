import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # The CPU checker (oracle) is tiny C: build it on demand.  The HIP library is NOT built here:
    # __graft_entry__.build() does that, and GPU tests fail loudly if it is missing.
    so = os.path.join(ROOT, "oracle", "_build", "libatsc_oracle.so")
    src = os.path.join(ROOT, "oracle", "atsc_oracle.c")
    if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "_build/libatsc_oracle.so"])


@pytest.fixture(scope="session")
def golden():
    import json

    import numpy as np
    g = json.load(open(os.path.join(ROOT, "tests", "golden", "golden.json")))
    w = np.load(os.path.join(ROOT, "tests", "golden", "windows.npz"))
    return g, w
