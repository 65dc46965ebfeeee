mkdir -p gpurun_out/r3full
export PDHG_DEV=1   # development variables on (csrc/common.hpp: dev_env)
timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | tail -15 > gpurun_out/r3full/tests.log
cat gpurun_out/r3full/tests.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
timeout 900 python bench.py > gpurun_out/r3full/bench_default.json 2> gpurun_out/r3full/bench_default.err
python -c "
import json
d=json.load(open('gpurun_out/r3full/bench_default.json'))
print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['setup_sec'], d['cpu_baseline']['value'], d.get('cpu_baseline_socket',{}).get('value'))
for o in d.get('other_configs', []): print(o['config']['workload'][:40], o.get('value'), o.get('roofline', {}).get('frac'), o.get('launch_path'))
"
PDHG_FORCE_DIST=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29555 bench.py --gpus 1 --steps 100 --warmup 10 --no-cpu-baseline > gpurun_out/r3full/bench_dist_world1.json 2> gpurun_out/r3full/bench_dist_world1.err
python -c "
import json
d=json.load(open('gpurun_out/r3full/bench_dist_world1.json')); print('dist world 1:', d['value'], d['setup_sec'], d.get('host_us_per_trial'))"
