#!/usr/bin/env python3
"""One-off soak: the seeded differential fuzz tests of tests/test_gpu_fuzz.py over many more
seeds than the test suite runs.  Prints the seeds that fail.
    python tools/long_fuzz.py 40 400 [test_fuzz_solve_and_warp ...]"""
import os
import sys
import traceback

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import contextlib
import io

import torch  # noqa: F401  (before the library is loaded: one HIP runtime per process, cvx_proj_amd/_native.py)
import test_gpu_fuzz as T
from cvx_proj_amd import _native


def main():
    lo, hi = int(sys.argv[1]), int(sys.argv[2])
    _native.lib()
    bad = []
    for name in (sys.argv[3:] or ("test_fuzz_solve_and_warp", "test_fuzz_equalize", "test_fuzz_ransac")):
        fn = getattr(T, name)
        for seed in range(lo, hi):
            try:
                with contextlib.redirect_stdout(io.StringIO()):
                    fn(_native, seed)
            except Exception as e:      # noqa: BLE001
                bad.append((name, seed, repr(e)[:200]))
                traceback.print_exc(limit=1)
            if (seed - lo) % 250 == 249:      # a long silent run looks hung to the GPU box's watchdog
                print(f"{name}: seed {seed}, failures so far {len(bad)}", flush=True)
        print(f"{name}: seeds {lo}..{hi - 1} done, failures so far {len(bad)}", flush=True)
    for b in bad:
        print("FAIL", *b)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
