import importlib.util
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
PKG_DIR = os.path.join(ROOT, "spartan-bn254_amd")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_pkg():
    """import the hyphenated package directory as `spartan_bn254_amd`"""
    if "spartan_bn254_amd" in sys.modules:
        return sys.modules["spartan_bn254_amd"]
    spec = importlib.util.spec_from_file_location("spartan_bn254_amd", os.path.join(PKG_DIR, "__init__.py"), submodule_search_locations=[PKG_DIR])
    mod = importlib.util.module_from_spec(spec)
    sys.modules["spartan_bn254_amd"] = mod
    spec.loader.exec_module(mod)
    return mod


@pytest.fixture(scope="session", autouse=True)
def _built():
    # the checker (C oracle) and the product library; both are plain `make` (hipcc cross-compiles without a GPU)
    if not os.path.exists(os.path.join(ROOT, "oracle", "libsbn_oracle.so")):
        subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle")], check=True)
    if not os.path.exists(os.path.join(PKG_DIR, "libsbn254_hip.so")):
        subprocess.run(["make", "-s", "-C", PKG_DIR], check=True)


@pytest.fixture(scope="session")
def sbn():
    return load_pkg()


@pytest.fixture(scope="session")
def ol():
    import oracle_lib
    return oracle_lib


@pytest.fixture(scope="session")
def pr():
    import pyref
    return pyref


def golden(name):
    with open(os.path.join(ROOT, "tests", "golden", name)) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def ctx(sbn):
    """one device context for the whole GPU session (tests run in one process, as the GPU box requires)"""
    c = sbn.Context(0)
    yield c
    c.close()


def fr_bytes(vals):
    import pyref
    return b"".join(pyref.scalar_to_bytes(v) for v in vals)


def rand_scalars(n, seed):
    import numpy as np
    import pyref
    rng = np.random.default_rng(seed)
    raw = rng.integers(0, 2**64, size=(n, 4), dtype=np.uint64)
    out = bytearray()
    for row in raw:
        v = (int(row[0]) | int(row[1]) << 64 | int(row[2]) << 128 | int(row[3]) << 192) % pyref.R
        out += v.to_bytes(32, "little")
    return bytes(out)
