"""pytest configuration: the ``gpu`` marker, import helpers for the hyphenated package and the oracle."""
import importlib
import os
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
for p in (str(ROOT), str(ROOT / "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)
os.environ.setdefault("TORCHDYNAMO_DISABLE", "1")

PKG = "matcha-tts-24k_amd"
GOLDEN = ROOT / "tests" / "golden"


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def sub(name: str):
    """Import ``matcha-tts-24k_amd.<name>``."""
    return importlib.import_module(f"{PKG}.{name}" if name else PKG)


@pytest.fixture(scope="session")
def pkg():
    return importlib.import_module(PKG)


@pytest.fixture(scope="session")
def hparams():
    return sub("hparams")


@pytest.fixture(scope="session")
def synthetic():
    return sub("synthetic")


@pytest.fixture(scope="session")
def oracle():
    import matcha_oracle
    return matcha_oracle
