import os
import sys

import pytest
import torch  # noqa: F401  -- before liblzani_hip.so: torch must load its own bundled HIP runtime first (same soname)

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("lz-ani_amd", "oracle", "tools", "tests"):
    sys.path.insert(0, os.path.join(ROOT, p))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def root():
    return ROOT
