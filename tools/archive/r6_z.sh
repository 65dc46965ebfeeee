# round 6: a second, deeper round of the random-case hunts with other seeds
export PDHG_DEV=1
PDHG_FUZZ_SCALE=15 timeout 1500 python -m pytest tests/test_gpu_property.py -x -q --hypothesis-show-statistics --hypothesis-seed=31337 > gpurun_out/prop_deep2.log 2>&1
grep -n "passing examples\|failing\|passed\|failed\|Error" gpurun_out/prop_deep2.log | head -30
timeout 900 python tools/big_shape_hunt.py 300 7 > gpurun_out/hunt3.log 2>&1; grep -c " ok " gpurun_out/hunt3.log; grep -v " ok " gpurun_out/hunt3.log | tail -8
