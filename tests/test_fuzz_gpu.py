"""A bounded run of the randomised parity sweep (scripts/fuzz_parity.py): random sizes, k, start radii
and point distributions, every kernel against the replay checker."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_randomised_parity_sweep():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "fuzz_parity.py"), "25", "7"], capture_output=True,
                       text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    last = r.stdout.strip().splitlines()[-1]
    assert last.startswith("fuzz:") and int(last.split()[1]) >= 10, last


@pytest.mark.gpu
def test_randomised_dbscan_sweep():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "fuzz_dbscan.py"), "15", "5"], capture_output=True,
                       text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    last = r.stdout.strip().splitlines()[-1]
    assert last.startswith("fuzz:") and int(last.split()[1]) >= 3, last
