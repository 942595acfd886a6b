#!/usr/bin/env python3
"""Run a pytest selection N times in one process and keep the full report of every failing round
(usage: python tools/loop_tests.py N out_dir <pytest args...>)."""
import io
import os
import sys
import time
from contextlib import redirect_stderr, redirect_stdout

import pytest

n, out_dir, args = int(sys.argv[1]), sys.argv[2], sys.argv[3:]
os.makedirs(out_dir, exist_ok=True)
bad = 0
for i in range(n):
    buf = io.StringIO()
    t0 = time.time()
    with redirect_stdout(buf), redirect_stderr(buf):
        rc = pytest.main(["-q", "-p", "no:cacheprovider"] + args)
    tail = buf.getvalue().strip().splitlines()[-1:]
    print("round %d rc=%d %.1fs %s" % (i, int(rc), time.time() - t0, tail), flush=True)
    if int(rc) != 0:
        bad += 1
        with open(os.path.join(out_dir, "fail_round_%d.log" % i), "w") as f:
            f.write(buf.getvalue())
print("%d of %d rounds failed" % (bad, n))
