"""The committed fixtures are what the reference's own code produces TODAY: oracle/gen_golden.py is re-run against /root/reference into a temporary
directory and every array / JSON document compared with tests/golden/ (tools/check_golden_regen.py).  Runs only where the reference tree exists (this
container; never on the GPU box: the GPU tests read fixtures only)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(not os.path.isdir("/root/reference"), reason="the reference tree is not present on this machine")
def test_fixtures_regenerate_identically():
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_golden_regen.py")], capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.strip()]
    assert len(lines) == 14 and all(l.startswith("identical") for l in lines), p.stdout
