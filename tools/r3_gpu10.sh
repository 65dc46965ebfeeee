mkdir -p gpurun_out/r3k
PDHG_VERBOSE=1 timeout 600 python bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-other-configs --profile-steps 0 > gpurun_out/r3k/create_configS.json 2> gpurun_out/r3k/create_configS.err
grep -i "pdhg_create" gpurun_out/r3k/create_configS.err
python -c "
import json; d=json.load(open('gpurun_out/r3k/create_configS.json')); print(d['value'], d['setup_sec'])"
for mb in 4 2 1.34; do
PDHG_SLAB_MB=$mb timeout 300 python bench.py --workload pagerank --steps 1000 --warmup 100 --no-cpu-baseline --no-other-configs > gpurun_out/r3k/pagerank_slab$mb.json 2> gpurun_out/r3k/pagerank_slab$mb.err
python -c "
import json; d=json.load(open('gpurun_out/r3k/pagerank_slab$mb.json')); print('slab MB $mb', d['value'], d['ms_per_step'], d['layout']['A_slabs'], d['roofline']['avg_launch_ms'], d['roofline']['avg_launch_ms_net'], d['roofline']['frac_net'])"
done
timeout 600 python -m pytest tests/test_gpu_tiled.py tests/test_gpu_edge_shapes.py tests/test_gpu_abi_errors.py -x -q -m gpu 2>&1 | tail -3
