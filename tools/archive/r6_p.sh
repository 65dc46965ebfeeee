export HSA_ENABLE_IPC_MODE_LEGACY=0
mkdir -p gpurun_out/r6prof
for cfg in "4 " "2 --dist-overlap"; do set -- $cfg; N=$1; shift
PDHG_RCCL_LIB=$PWD/tests/fake_rccl/libfake_rccl.so timeout 1200 python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port 2981$N bench.py --gpus $N --steps 20 --warmup 5 "$@" > gpurun_out/r6prof/fake_${N}.json 2> gpurun_out/r6prof/fake_${N}.err
echo "rc=$? N=$N $@"; tail -1 gpurun_out/r6prof/fake_${N}.json | cut -c1-900
python - <<PY
import json
d=json.loads(open("gpurun_out/r6prof/fake_${N}.json").read().strip().split("\n")[-1]); print(len(json.dumps(d)), d["n_gpus"], d["value"], d.get("transport"), d["scaling_model"])
f=json.load(open("bench_details.json")); print(list(f["layout_choices"].keys()), f["layout_choices"].get("all_gather",{}).get("mode"))
PY
done
