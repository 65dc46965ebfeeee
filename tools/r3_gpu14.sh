mkdir -p gpurun_out/r3p
timeout 1200 python -m pytest tests/test_gpu_tiled.py tests/test_gpu_lazy_accept.py tests/test_gpu_kat.py tests/test_gpu_native_take_step.py tests/test_gpu_full_size.py -x -q -m gpu 2>&1 | tail -6
for g in 1 0; do
PDHG_GRAPH=$g timeout 600 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-other-configs --profile-steps 0 > gpurun_out/r3p/configS_graph$g.json 2>/dev/null
PDHG_GRAPH=$g timeout 600 python bench.py --m 1000000 --n 1000000 --steps 2000 --warmup 100 --no-cpu-baseline --no-other-configs --profile-steps 0 > gpurun_out/r3p/random1m_graph$g.json 2>/dev/null
PDHG_GRAPH=$g timeout 600 python bench.py --m 4000000 --n 4000000 --steps 1000 --warmup 100 --no-cpu-baseline --no-other-configs --profile-steps 0 > gpurun_out/r3p/random4m_graph$g.json 2>/dev/null
for f in configS random1m random4m; do python -c "
import json; d=json.load(open('gpurun_out/r3p/${f}_graph$g.json')); print('$f graph=$g', d['value'], d['ms_per_step'], d['layout']['trial_graph'])"; done
done
