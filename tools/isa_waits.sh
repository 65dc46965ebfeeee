#!/bin/bash
# The s_waitcnt vmcnt(...) histogram of the product kernels' ISA (device-only compile, ~30 s; no GPU needed).  Round 6 found
# two accidents this way: `vmcnt(1)` in front of every gather of the sliced jagged loop (two gathers in flight per wave), and
# one behind every (col, val) load pair of the CSR stream kernel (eight dependent round trips per row block since round 1).
#   tools/isa_waits.sh ["-DFLAG ..."]
cd "$(dirname "$0")/.."
hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -std=c++17 -I include --cuda-device-only -S $1 -o /tmp/pdhg_isa.s firstorderlp.jl_amd/csrc/pdhg_hip.hip || exit 1
python3 - <<'PY'
import collections, re
txt = open("/tmp/pdhg_isa.s").read()
for m in re.finditer(r"\n(_ZN[^\n:]*(spmv_|steps_kernel|trial_kernel|small_lp)[^\n:]*):[^\n]*\n(.*?)\n\s*s_endpgm", txt, re.S):
    body = m.group(3)
    c = collections.Counter(re.findall(r"s_waitcnt ([^\n]*)", body))
    vm = {k: v for k, v in sorted(c.items()) if "vmcnt" in k}
    name = re.sub(r"^_ZN12_GLOBAL__N_1\d+", "", m.group(1))[:70]
    print(f"{name:72s} lines {body.count(chr(10)):5d}  {vm}")
PY
