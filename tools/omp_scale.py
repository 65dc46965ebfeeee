import sys, time, os
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import numpy as np, folp_loader
folp_loader.load()
import bench
from firstorderlp_jl_amd.generators import random_lp
from oracle.oracle import OmpCpuState
print("nproc", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)))
for f in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpuset.cpus.effective"):
    try: print(f, open(f).read().strip())
    except Exception as e: print(f, "n/a")
os.system("lscpu | grep -E 'Model name|Socket|Core|Thread|NUMA'")
p = random_lp(4_000_000, 4_000_000, 10, 1)
A = p.constraint_matrix
step = 1.0/np.abs(A.data).max(); pw = float(np.linalg.norm(p.objective_vector)/np.linalg.norm(p.right_hand_side))
cores = bench.socket0_cores()
print("socket0 cores", len(cores), cores[:8], "...")
for nt in (1, 4, 16, 32, 64):
    om = OmpCpuState(A.shape[0],A.shape[1],A.indptr,A.indices,A.data,p.objective_vector,p.right_hand_side,p.variable_lower_bound,p.variable_upper_bound,p.num_equalities,cpus=cores[:nt])
    om.set_scalars(step,pw)
    for _ in range(2): om.take_step_adaptive()
    t0=time.perf_counter(); it=0
    while time.perf_counter()-t0 < 3.0: om.take_step_adaptive(); it+=1
    dt=time.perf_counter()-t0
    print(nt, "threads", round(it/dt,2), "it/s")
    om.close()
