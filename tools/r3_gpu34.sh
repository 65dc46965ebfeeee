#!/bin/bash
cd "$GRAFT_REPO_ROOT"
timeout 900 python -m pytest tests/test_gpu_device_loop.py -x -q 2>&1 | grep -v "^[A-Z][A-Za-z]* \(version\|path\) *:\|Hostname" | tail -25
