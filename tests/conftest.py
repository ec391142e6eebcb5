import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN_DIR = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # a fresh checkout has no built artefacts (they are git-ignored): build them once (hipcc
    # cross-compiles gfx950 without a GPU; gcc builds the oracle and the generators)
    built = [os.path.join(ROOT, "rust-msbwt_amd", "libmsbwt_hip.so"), os.path.join(ROOT, "oracle", "libmsbwt_oracle.so"),
             os.path.join(ROOT, "synth", "libmsbwt_synth.so")]
    if not all(os.path.exists(p) for p in built):
        import __graft_entry__
        __graft_entry__.build()


@pytest.fixture(scope="session")
def golden():
    with open(os.path.join(GOLDEN_DIR, "reference_vectors.json")) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN_DIR


def expand_case(case):
    """A golden case gives its text either literally or as [[char, count], ...]."""
    if "text" in case:
        return case["text"]
    return "".join(ch * n for ch, n in case["repeat"])
