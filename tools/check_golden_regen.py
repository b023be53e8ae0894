#!/usr/bin/env python3
"""Regenerate every fixture from the reference (oracle/gen_golden.py, needs /root/reference) into a temporary directory and
compare with the committed tests/golden/: arrays must be equal element for element, JSON documents equal.  CPU only."""
import json
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")


def main():
    bad = 0
    with tempfile.TemporaryDirectory() as tmp:
        subprocess.check_call([sys.executable, os.path.join(ROOT, "oracle", "gen_golden.py")], env=dict(os.environ, ISX_GOLDEN_OUT=tmp),
                              stdout=subprocess.DEVNULL)
        names = sorted(set(os.listdir(tmp)) | set(os.listdir(GOLDEN)))
        for n in names:
            a, b = os.path.join(GOLDEN, n), os.path.join(tmp, n)
            if not (os.path.exists(a) and os.path.exists(b)):
                print("MISSING  ", n); bad += 1; continue
            if n.endswith(".json"):
                same = json.load(open(a)) == json.load(open(b))
            else:
                x, y = np.load(a), np.load(b)
                same = sorted(x.files) == sorted(y.files) and all(np.array_equal(x[k], y[k], equal_nan=True) for k in x.files)
            print("identical" if same else "DIFFERS  ", n)
            bad += not same
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
