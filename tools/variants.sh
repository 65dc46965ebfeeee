#!/bin/bash
export PDHG_DEV=1   # development variables on (csrc/common.hpp: dev_env)
# usage: tools/variants.sh NAME "-DFLAG ..." [NAME "-D..."]...   (run in the build container)
# builds firstorderlp.jl_amd/csrc/variants/libpdhg_NAME.so for tools/trial_time.py
cd "$(dirname "$0")/.."
mkdir -p firstorderlp.jl_amd/csrc/variants
while [ $# -ge 2 ]; do
  hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -std=c++17 -shared -fPIC -pthread $2 -I include \
    -o firstorderlp.jl_amd/csrc/variants/libpdhg_$1.so firstorderlp.jl_amd/csrc/pdhg_hip.hip -L/opt/rocm/lib -lrccl -Wl,-rpath,/opt/rocm/lib &
  shift 2
done
wait
ls -la firstorderlp.jl_amd/csrc/variants/
