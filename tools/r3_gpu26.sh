#!/bin/bash
cd "$GRAFT_REPO_ROOT"
timeout 900 python -m pytest tests/test_gpu_device_layout.py -x -q 2>&1 | tail -4
bash tools/r3_gpu25.sh 2>&1 | grep "pagerank\|pdhg_create"
timeout 600 python tools/solve_demo.py --workload pagerank --verbosity 0 2>/dev/null | tail -1
