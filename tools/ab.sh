#!/bin/bash
export PDHG_DEV=1   # development variables on (csrc/common.hpp: dev_env)
# usage: tools/ab.sh "ENV1=.. ENV2=.." ...   -> one bench line per configuration
for cfg in "$@"; do
  env $cfg python bench.py --steps 100 --warmup 10 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$cfg', '| it/s', d['value'], '| ms', d['ms_per_step'], '|', {k.split('<')[1][:-1] if '<' in k else k:(v['avg_ms'],v['achieved_GBps']) for k,v in d['kernels'].items() if 'spmv' in k})"
done
