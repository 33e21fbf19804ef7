"""Shared fixtures.  `-m "not gpu"` runs here (no GPU): oracle vs golden vectors, host logic, C-ABI exports.
`-m gpu` runs on the MI355X box: parity of the HIP path (through the C ABI) against the oracle and the goldens.

/root/reference does not exist on the GPU box: nothing here reads it.  The oracle (oracle/) is test
infrastructure: the product never imports it.
"""
import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(rel):
    with open(os.path.join(GOLDEN, rel), encoding="utf-8") as f:
        return json.load(f)


@pytest.fixture(scope="session")
def golden():
    return load_golden


@pytest.fixture(scope="session")
def ref_dir():
    return os.path.join(GOLDEN, "ref")


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as O

    O.build()
    O.lib()
    return O


@pytest.fixture(scope="session")
def swt():
    """The product package with libswt_hip.so built (hipcc cross-compiles without a GPU)."""
    import subword_tokenizers_amd as S
    from subword_tokenizers_amd import _build

    _build.build()
    return S


@pytest.fixture(scope="session")
def native(swt):
    from subword_tokenizers_amd import _native

    _native.lib()
    return _native


@pytest.fixture(scope="session")
def corpora(golden):
    return {"pan": golden("ref/data/pan_tadeusz.json"), "t5k": golden("ref/data/train-5K.json"),
            "pan_tokens": golden("ref/data/pan_tadeusz.tokens.json")}
