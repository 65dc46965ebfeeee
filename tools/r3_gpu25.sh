#!/bin/bash
cd "$GRAFT_REPO_ROOT"
python - <<'PY' 2>&1 | grep -v amdgpu.ids
import os, sys, time
sys.path.insert(0, os.getcwd())
import folp_loader; pkg = folp_loader.load()
from firstorderlp_jl_amd.generators import pagerank_lp, l1_svm_rcv1_like_lp, random_lp
for name, p in (("pagerank", pagerank_lp(1_000_000)), ("l1svm", l1_svm_rcv1_like_lp()), ("random1M", random_lp(1_000_000, 1_000_000, 10, 12345))):
    for rep in range(3):
        os.environ["PDHG_VERBOSE"] = "1" if rep == 2 else ""
        if rep < 2: os.environ.pop("PDHG_VERBOSE")
        t0 = time.perf_counter(); eng = pkg.HipPdhgEngine.from_problem(p); t1 = time.perf_counter()
        eng.close(); t2 = time.perf_counter()
        print(f"{name}: create {1e3*(t1-t0):.1f} ms, close {1e3*(t2-t1):.1f} ms", flush=True)
PY
