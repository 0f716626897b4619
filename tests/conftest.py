import os
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))

GOLDEN = ROOT / "tests" / "golden"


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


@pytest.fixture(scope="session", autouse=True)
def _built_library():
    """The C-ABI library is a build artefact (git-ignored): (re)build it when missing or stale.
    hipcc cross-compiles for gfx950 without a GPU, so this also works in the CPU-only container."""
    from adsorbdiff_amd import build as _build

    if _build.needs_build():
        _build.build()
    return _build.LIB


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN
