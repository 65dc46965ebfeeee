#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/g23
timeout 900 python -m pytest tests/test_gpu_native_take_step.py -x -q 2>&1 | tail -15
