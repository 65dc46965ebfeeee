# failing cases of tools/dist_shape_hunt.py with the workers' own output
python - <<'PY'
import sys
sys.path.insert(0, ".")
from tests.test_gpu_fake_rccl import _spawn
for args, env in (((3, "traj", "global", "1", "rand:1180834645"), {}), ((3, "traj", "rows", "0", "rand:1247479809"), {})):
    try:
        print(_spawn(*args, **env)[-300:])
    except AssertionError as e:
        keep = [l for l in str(e).splitlines() if any(w in l for w in ("FAILED:", "diverged", "differs", "Error", "dist_fake_worker.py\", line", "dist_rank_worker.py\", line", "Mismatch", "Max ", "assert"))]
        print(args, "\n   ".join(keep[-25:]))
PY
