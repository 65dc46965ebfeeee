#!/usr/bin/env python3
"""Random LP shapes (1 ... 4 000 rows / columns: fewer rows than ranks, fewer columns than ranks x chunks included) through
the one-process-per-GPU routes as REAL processes on one GPU over the test-only RCCL stand-in (tests/fake_rccl): world 2 / 3 /
4, both ingest forms, reduce-scatter / per-slice reduce, the all-gather of xbar whole or in column chunks.  Every case is
tests/workers/dist_fake_worker.py's `traj` or `agtraj` check: all ranks hold the same bits, bitwise the in-process group,
decisions equal to the single handle's and iterates within 1e-9.  Usage: python tools/dist_shape_hunt.py [cases] [seed]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from tests.test_gpu_fake_rccl import _spawn

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 12
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
failures = 0
for c in range(cases):
    world = int(rng.choice([2, 3, 4]))
    ingest = str(rng.choice(["global", "rows"]))
    seed = int(rng.integers(1, 2 ** 31 - 1))
    mode = str(rng.choice(["traj0", "traj1", "agtraj"]))
    t0 = time.time()
    try:
        if mode == "agtraj":
            chunks = int(rng.choice([2, 3, 4]))
            mode = f"agtraj{chunks}"
            out = _spawn(world, "agtraj", ingest, f"rand:{seed}", PDHG_DEV="1", PDHG_DIST_AG_CHUNKS=str(chunks))
        else:
            out = _spawn(world, "traj", ingest, mode[-1], f"rand:{seed}")
        line = [l for l in out.splitlines() if "fake worker ok" in l][-1]
        print(f"[{c}] world {world} {ingest} {mode} seed {seed}: {line[:150]} ({time.time() - t0:.1f}s)", flush=True)
    except AssertionError as e:
        failures += 1
        keep = [l for l in str(e).splitlines() if any(w in l for w in ("FAILED:", "diverged", "differs", "Error:", "dist_fake_worker.py\", line", "Mismatch", "Max "))]
        print(f"[{c}] world {world} {ingest} {mode} seed {seed}: FAILED\n   " + "\n   ".join(keep[-12:]), flush=True)
print(f"{cases} cases, {failures} failures")
sys.exit(1 if failures else 0)
