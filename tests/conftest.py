import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def golden_cases():
    return sorted(f[:-4] for f in os.listdir(GOLDEN) if f.endswith(".npz") and f not in ("meshes.npz", "screenshots.npz") and not f.startswith(("gl_", "gl1_")))


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN
