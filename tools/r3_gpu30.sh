#!/bin/bash
cd "$GRAFT_REPO_ROOT"
for e in 1 2 4 8 16; do
echo "== PDHG_EV_ELEMS=$e"
PDHG_EV_ELEMS=$e timeout 300 python tools/eval_cost.py l1svm 2>&1 | grep "point=0, range=0\|eval_point(AVERAGE) (fresh\|dist"
PDHG_EV_ELEMS=$e timeout 300 python tools/eval_cost.py 1000000 2>&1 | grep "point=0, range=0\|eval_point(AVERAGE) (fresh\|dist"
done
