"""-m gpu: randomised parity sweep (tools/fuzz_parity.py): random m, p, d in 1..3, n around the 64-row tile boundaries, random kernel
kinds / hyper-parameters / test-point counts; logpdf, posterior marginals, posterior logpdf and the dense-H logpdf against the
oracle at the rtol 1e-6 bar (observed: 1e-9 or better)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_randomised_parity_sweep():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fuzz_parity.py"), "30", "7"], cwd=ROOT, capture_output=True,
                         text=True, timeout=600)
    assert out.returncode == 0 and "fuzz parity OK" in out.stdout, out.stdout[-2000:] + out.stderr[-2000:]
