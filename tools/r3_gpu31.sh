#!/bin/bash
cd "$GRAFT_REPO_ROOT"
timeout 900 python -m pytest tests -m gpu -x -q -k "native_take_step or lazy or kat or step_parity or exact or big_nnz or row_order" 2>&1 | grep -v "^[A-Z][A-Za-z]* \(version\|path\) *:\|Hostname" | tail -3
for hw in 0 1 0 1; do
  for n in 1000000 4000000; do
    PDHG_TRIAL_HOST_WORD=$hw timeout 600 python bench.py --m $n --n $n --steps 1500 --warmup 100 --no-cpu-baseline --no-other-configs 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('hostword $hw n=$n', d['value'], d['ms_per_step'])"
  done
done
for hw in 0 1; do
PDHG_TRIAL_HOST_WORD=$hw timeout 600 python bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-other-configs 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('hostword $hw S', d['value'], d['ms_per_step'])"
done
