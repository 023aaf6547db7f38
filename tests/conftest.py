import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# The model raises FileNotFoundError, like the reference (transformer.py:622-624), when `pretrained` (default True) finds no ViT
# checkpoint; the tests construct models with seeded weights and no checkpoint on disk, so they opt in to the documented escape
# (tests/test_layout.py::test_missing_pretrained_checkpoint_raises checks the default behaviour with the variable removed).
os.environ.setdefault("VTAMIQ_ALLOW_MISSING_WEIGHTS", "1")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def _libraries_are_current():
    """Both libraries -- the product's and the fp8 experiment's -- are (re)built from the tree's sources before the first test when they are missing
    or older than a source (vtamiq_amd.build: a no-op otherwise; hipcc is on the CPU container and on the GPU boxes), so that no test runs against
    a stale .so.  The tests themselves never build on demand: a missing library is a failure there (no fallback)."""
    from vtamiq_amd import build
    build.build(verbose=False)
    build.build(verbose=False, fp8=True)


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")
