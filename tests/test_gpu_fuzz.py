"""Randomised end-to-end sweep: topology (layers, register tokens, scales, head on/off), batch, patch count, numerics mode,
FR pairs and pairwise triplets drawn from a seeded generator, against the oracle on the host (tools/fuzz_parity.py)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("seed", [2, 3])
def test_random_configurations_against_oracle(seed):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fuzz_parity.py"), "--cases", "8", "--seed", str(seed)],
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert "misses: 0" in r.stdout
