"""pytest configuration: registers the `gpu` marker and builds the checker."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line(
        "markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # The oracle (CPU checker) is test infrastructure: build it on demand.
    so = os.path.join(ROOT, "oracle", "liboracle.so")
    src = os.path.join(ROOT, "oracle", "od_oracle.c")
    if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"),
                        "liboracle.so"], check=True, capture_output=True)
