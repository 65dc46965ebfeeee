#!/usr/bin/env python3
"""PDHG inner-loop benchmark (BASELINE.json metric: PDHG iterations/s and
achieved GB/s of the CSR SpMV vs the HBM roofline).

    python bench.py --gpus N --steps K --warmup W

A "step" is one ``take_step`` of the adaptive policy (pdhg.jl:653-731): one
accepted PDHG iteration including any rejected trials, from the zero start,
no restarts/rescaling (the reference's "basic algorithm" timer,
pdhg.jl:1025-1047).  Workload at N=1: BASELINE configs[4], synthetic random LP
m = n = 10M, nnz = 100M, fp64.  With N > 1 the same LP is row-partitioned
(strong scaling), one process per GPU over RCCL.

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md)


def socket0_cores():
    """One logical CPU per physical core of socket 0 (from /proc/cpuinfo)."""
    seen, cpus = set(), []
    cpu = phys = core = None
    try:
        with open("/proc/cpuinfo") as fh:
            for line in fh.read().split("\n") + [""]:
                if line.startswith("processor"):
                    cpu = int(line.split(":")[1])
                elif line.startswith("physical id"):
                    phys = int(line.split(":")[1])
                elif line.startswith("core id"):
                    core = int(line.split(":")[1])
                elif not line.strip() and cpu is not None:
                    if (phys or 0) == 0 and (phys, core) not in seen:
                        seen.add((phys, core))
                        cpus.append(cpu)
                    cpu = phys = core = None
    except OSError:
        pass
    return cpus or list(range(max(1, (os.cpu_count() or 2) // 2)))


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--m", type=int, default=10_000_000)
    ap.add_argument("--n", type=int, default=10_000_000)
    ap.add_argument("--nnz-per-row", type=int, default=10)
    ap.add_argument("--seed", type=int, default=12345)
    ap.add_argument("--workload", choices=["random", "pagerank", "l1svm"], default="random",
                    help="random: BASELINE configs[4] (default); pagerank: configs[2] (--n nodes)")
    ap.add_argument("--profile-steps", type=int, default=30)
    ap.add_argument("--cpu-baseline-seconds", type=float, default=12.0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    return ap.parse_args()


def main():
    args = parse()
    # Native libraries print to stdout too (RCCL's version banner on communicator
    # creation, for one).  The contract is ONE JSON line on rank 0's stdout: send
    # file descriptor 1 to stderr until that line is written.
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    import numpy as np
    import torch
    import folp_loader
    pkg = folp_loader.load()
    from firstorderlp_jl_amd.generators import l1_svm_rcv1_like_lp, pagerank_lp, random_lp
    from firstorderlp_jl_amd.primal_dual_hybrid_gradient import (
        AdaptiveStepsizeParams, PdhgSolverState, take_step)
    from firstorderlp_jl_amd import _lib
    if not os.path.exists(_lib.LIB_PATH) and int(os.environ.get("WORLD_SIZE", "1")) == 1:
        _lib.build()       # fresh clone on a 1-GPU box; normally the in-tree library is already there

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (no CPU fallback)")
    # Dev aid for 1-GPU boxes: PDHG_BENCH_SHARE_GPU=1 puts every rank on cuda:0 and
    # uses gloo on the device exchange tensor (RCCL refuses two ranks per device),
    # so the world_size > 1 code of this script can be exercised; timings of such
    # a run mean nothing.
    share_gpu = os.environ.get("PDHG_BENCH_SHARE_GPU", "0") == "1"
    if share_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dist = None
    # PDHG_FORCE_DIST=1: run the row-partitioned engine + RCCL even with one rank
    # (how the N > 1 code path is exercised on a 1-GPU box).
    force_dist = os.environ.get("PDHG_FORCE_DIST", "0") == "1"
    if world > 1 or force_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        if share_gpu:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world,
                                    device_id=torch.device("cuda", local_rank))

    t0 = time.time()
    if args.workload == "pagerank":
        problem = pagerank_lp(args.n, 4 * args.n, 0.99, seed=0)
        wl = (f"PageRank LP (generate_pagerank_lp.jl model) nodes={args.n} approx_edges={4 * args.n} "
              "damping=0.99 (BASELINE configs[2])")
    elif args.workload == "l1svm":
        problem = l1_svm_rcv1_like_lp(seed=0)
        wl = ("L1-SVM LP (generate_l1_svm_lp.jl model) on synthetic rcv1.binary-shaped data "
              "20242 x 47236, ~74 nnz/row, lambda=1 (BASELINE configs[3]; rcv1 itself is not available offline)")
    else:
        problem = random_lp(args.m, args.n, args.nnz_per_row, args.seed)
        wl = None
    A = problem.constraint_matrix
    nnz = int(A.nnz)
    t_gen = time.time() - t0

    t0 = time.time()
    if dist is not None:
        from firstorderlp_jl_amd.distributed import make_row_partitioned_hip_engine
        eng = make_row_partitioned_hip_engine(problem, device_id=local_rank)
        local = eng.local
    else:
        eng = pkg.HipPdhgEngine.from_problem(problem, device_id=local_rank)
        local = eng
    t_create = time.time() - t0

    step0 = 1.0 / float(np.abs(A.data).max())                      # pdhg.jl:823
    cn = float(np.sqrt(np.sum(problem.objective_vector ** 2)))
    bn = float(np.sqrt(np.sum(problem.right_hand_side ** 2)))
    pw0 = cn / bn if cn > 0 and bn > 0 else 1.0                    # saddle_point.jl:1049
    state = PdhgSolverState(eng, step_size=step0, primal_weight=pw0)
    policy = AdaptiveStepsizeParams(0.3, 0.6)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        take_step(policy, state)
    barrier()
    trials0 = state.total_number_iterations
    t0 = time.perf_counter()
    for _ in range(args.steps):
        take_step(policy, state)
    barrier()
    elapsed = time.perf_counter() - t0
    trials = state.total_number_iterations - trials0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    ms_per_step = 1e3 * elapsed / args.steps
    value = args.steps / elapsed

    # ---- roofline of the dominant kernel: HIP events on the engine's stream
    roofline = None
    kernels = {}
    local.profile_enable(True)
    trials_before = state.total_number_iterations
    for _ in range(args.profile_steps):
        take_step(policy, state)
    prof_trials = state.total_number_iterations - trials_before
    for kid in range(_lib.K_COUNT):
        cnt, ms = local.profile_read(kid)
        if cnt:
            byts = local.kernel_algorithmic_bytes(kid)
            # the row-partitioned form may issue A_p'y'_p in parts (several launches
            # per trial): price the whole product, not one part, against its bytes
            per_trial = max(1, round(cnt / max(prof_trials, 1)))
            avg_ms = ms / cnt * per_trial
            kernels[local.kernel_name(kid)] = {
                "launches": cnt, "avg_ms": round(avg_ms, 5),
                "algorithmic_bytes": byts,
                "achieved_GBps": round(byts / (avg_ms * 1e-3) / 1e9, 1)}
            if per_trial > 1:
                kernels[local.kernel_name(kid)]["launches_per_trial"] = per_trial
    local.profile_enable(False)
    dom = max((_lib.K_SPMV_DUAL, _lib.K_SPMV_ATY),
              key=lambda k: local.profile_read(k)[1])
    dk = kernels[local.kernel_name(dom)]
    traffic = None
    try:   # PMC-derived bytes/launch, measured offline with rocprofv3 on this exact workload (1 GPU)
        if dist is not None:
            raise OSError("PMC traffic was collected for the single-GPU launch only")
        with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as fh:
            traffic = json.load(fh).get(
                f"{local.kernel_name(dom)}@m={A.shape[0]},n={A.shape[1]},nnz={nnz}")
    except OSError:
        pass
    try:     # the box's own streaming ceiling next to the spec figure
        triad = round(local.measure_triad(1 << 26, 5), 1)
    except Exception:   # measurement extra
        triad = None
    roofline = {"bound": "hbm", "kernel": local.kernel_name(dom),
                "achieved": dk["achieved_GBps"], "peak": HBM_PEAK_GBS,
                "unit": "GB/s", "frac": round(dk["achieved_GBps"] / HBM_PEAK_GBS, 4),
                "peak_measured_triad": triad,
                "frac_of_triad": round(dk["achieved_GBps"] / triad, 4) if triad else None,
                "note": "a random 8-byte gather per nonzero bounds this kernel (L2 request path), not HBM "
                        "streaming: DESIGN.md section 4, profiles/r01_sweep_probe.txt",
                "traffic": traffic, "avg_launch_ms": dk["avg_ms"],
                "algorithmic_bytes_per_launch": dk["algorithmic_bytes"]}

    # ---- CPU baseline: the literal single-thread restatement, bounded sample
    cpu_baseline = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle.oracle import OracleState
        Q = problem.objective_matrix
        st = OracleState(A.shape[0], A.shape[1], A.indptr, A.indices, A.data,
                         problem.objective_vector, problem.right_hand_side,
                         problem.variable_lower_bound, problem.variable_upper_bound,
                         problem.num_equalities)
        st.step_size, st.primal_weight = step0, pw0
        t0 = time.perf_counter()
        its = 0
        while its < 3 or (time.perf_counter() - t0 < args.cpu_baseline_seconds and its < 1000):
            st.take_step_adaptive(0.3, 0.6)
            its += 1
        dt = time.perf_counter() - t0
        cpu_baseline = {"value": round(its / dt, 4), "unit": "iterations/s",
                        "cores": 1, "kind": "port",
                        "sample": f"first {its} adaptive take_step calls on the same LP "
                                  f"({st.total_number_iterations} trials), oracle/pdhg_oracle.c, "
                                  f"1 thread of {os.cpu_count()} host cores"}
        st.close()

    # ---- second CPU comparator: the same step with OpenMP on ONE socket's cores
    cpu_socket = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        try:
            cores = socket0_cores()
            from oracle.oracle import OmpCpuState
            om = OmpCpuState(A.shape[0], A.shape[1], A.indptr, A.indices, A.data,
                             problem.objective_vector, problem.right_hand_side,
                             problem.variable_lower_bound, problem.variable_upper_bound,
                             problem.num_equalities, cpus=cores)
            om.set_scalars(step0, pw0)
            for _ in range(2):
                om.take_step_adaptive(0.3, 0.6)
            t0 = time.perf_counter()
            its = 0
            while its < 5 or (time.perf_counter() - t0 < 0.6 * args.cpu_baseline_seconds and its < 5000):
                om.take_step_adaptive(0.3, 0.6)
                its += 1
            dt = time.perf_counter() - t0
            cpu_socket = {"value": round(its / dt, 4), "unit": "iterations/s", "cores": om.threads(),
                          "kind": "port-openmp",
                          "sample": f"{its} adaptive take_step calls on the same LP, oracle/pdhg_cpu_omp.c, "
                                    f"one thread per physical core of socket 0 ({len(cores)} cores)"}
            om.close()
        except Exception as exc:   # measurement extra: never fail the bench line for it
            cpu_socket = {"error": repr(exc)}

    if rank == 0:
        m, n = A.shape
        b_pair = 24 * nnz + 16 * (m + n) + 4 * (m + n + 2)
        b_iter = b_pair + 8 * (13 * n + 6 * m)
        out = {
            "metric": "pdhg_iterations_per_sec", "value": round(value, 3),
            "unit": "iterations/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(ms_per_step, 5),
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": (wl or f"random LP m={m} n={n} nnz={nnz} seed={args.seed} "
                                    "(BASELINE configs[4])") + ", adaptive step, zero start, "
                                   "no restarts/rescaling",
                       "m": m, "n": n, "nnz": nnz,
                       "parallelism": "single GPU" if dist is None else f"row-partition x{world} + RCCL all-reduce"},
            "trials_per_step": round(trials / args.steps, 4),
            "whole_iteration_GBps": round(b_iter * (trials / args.steps) / (ms_per_step * 1e-3) / 1e9, 1),
            "roofline": roofline, "cpu_baseline": cpu_baseline, "cpu_baseline_socket": cpu_socket,
            "kernels": kernels,
            "setup_sec": {"generate": round(t_gen, 1), "create_upload": round(t_create, 1)},
        }
        if cpu_baseline:
            out["speedup_vs_cpu_port"] = round(value / cpu_baseline["value"], 1)
        if cpu_socket and "value" in cpu_socket:
            out["speedup_vs_cpu_socket"] = round(value / cpu_socket["value"], 1)
        sys.stdout.flush()
        os.dup2(real_stdout, 1)
        print(json.dumps(out), flush=True)
        os.dup2(2, 1)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
