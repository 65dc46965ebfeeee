mkdir -p gpurun_out/r3full
timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | tail -25 > gpurun_out/r3full/tests.log
cat gpurun_out/r3full/tests.log
