mkdir -p gpurun_out/r3m
timeout 1200 python -m pytest tests/test_gpu_device_layout.py -x -q -m gpu 2>&1 | tail -30 > gpurun_out/r3m/tests.log
cat gpurun_out/r3m/tests.log
for dl in 1 0; do
PDHG_DEVICE_LAYOUT=$dl PDHG_VERBOSE=1 timeout 600 python bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-other-configs --profile-steps 0 > gpurun_out/r3m/create_configS_dl$dl.json 2> gpurun_out/r3m/create_configS_dl$dl.err
grep -i "pdhg_create" gpurun_out/r3m/create_configS_dl$dl.err
python -c "
import json; d=json.load(open('gpurun_out/r3m/create_configS_dl$dl.json')); print('device layout $dl', d['value'], d['setup_sec'], d['layout'])"
done
