mkdir -p gpurun_out/r3o
for cap in 2.0 1.0; do for fill in 110 180 260; do
name=cap${cap}_fill${fill}
PDHG_SPMV=tiled PDHG_TW_NNZ_CAP=$cap PDHG_TILE_FILL=$fill PDHG_VERBOSE=1 timeout 300 python bench.py --workload pagerank --steps 600 --warmup 60 --no-cpu-baseline --no-other-configs > gpurun_out/r3o/pagerank_$name.json 2> gpurun_out/r3o/pagerank_$name.err
python -c "
import json; d=json.load(open('gpurun_out/r3o/pagerank_$name.json')); print('$name', d['value'], d['ms_per_step'], {k:v['avg_ms'] for k,v in d['kernels'].items() if 'spmv' in k})"
grep "tiles (widest" gpurun_out/r3o/pagerank_$name.err | cut -c1-230
done; done
