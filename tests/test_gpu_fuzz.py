"""A short run of the randomised parity campaign (tools/fuzz_parity.py): random matrices,
formats, ABFT modes, layouts (streaming / forced panels), SpMV in one or two parts, single
and double bit flips -- stored words, y (bit for bit, two passes) and event streams against
the CPU oracle.  The long form (`python tools/fuzz_parity.py 300`: ~20 000 cases) is run by
hand; its last result is quoted in DESIGN.md."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_short_fuzz_campaign():
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fuzz_parity.py"), "20", "100000"],
                       capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stdout[-3000:] + p.stderr[-1000:]
    assert "0 failures" in p.stdout


def test_short_call_sequence_campaign():
    """tools/fuzz_sequence.py: random interleavings of spmv / dot / calc_xr / calc_p / copy /
    map / unmap against a numpy model -- the fused dot and the deferred x update must never
    show (vectors bit-identical whenever read)."""
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fuzz_sequence.py"), "15", "500000"],
                       capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stdout[-3000:] + p.stderr[-1000:]
    assert " 0 failures" in p.stdout
