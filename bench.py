#!/usr/bin/env python3
"""PDHG inner-loop benchmark (BASELINE.json metric: PDHG iterations/s and
achieved GB/s of the CSR SpMV vs the HBM roofline).

    python bench.py --gpus N --steps K --warmup W

A "step" is one ``take_step`` of the adaptive policy (pdhg.jl:653-731): one
accepted PDHG iteration including any rejected trials, from the zero start,
no restarts/rescaling (the reference's "basic algorithm" timer,
pdhg.jl:1025-1047).  Workload at N=1: BASELINE configs[4], synthetic random LP
m = n = 10M, nnz = 100M, fp64.  With N > 1 the same LP is row-partitioned
(strong scaling), one process per GPU; the exchange (RCCL reduce-scatter /
all-gather over xGMI) is issued by the library itself (csrc/dist.hpp) --
torch.distributed (gloo) is used here only for the harness: handing out the
communicator id, the barriers around the timed region and the max over ranks.

Prints ONE compact JSON line (<= 4 KB: the contract's keys, `roofline`,
`cpu_baseline`, one short row per other config) as the LAST line of rank 0's
stdout; the whole record -- `kernels`, `scaling_model`, `ceiling`,
`self_profile`, the CPU comparator's variants, `trial_timeline`, the full
`other_configs` legs -- goes to bench_details.json next to this script and to
stderr.  At N=1 the record also carries, under "other_configs", the same
measurement for BASELINE configs[2] (PageRank LP, 1M nodes) and configs[3]
(L1-SVM LP on rcv1-shaped data) unless --no-other-configs is given;
--workload picks one of them as the headline.
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
PMC_TRAFFIC_FILE = os.path.join("profiles", "pmc_traffic.json")


def socket0_cores(all_threads=False):
    """One logical CPU per physical core of socket 0 (from /proc/cpuinfo); all_threads: every hardware thread of it."""
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
                    if (phys or 0) == 0 and (all_threads or (phys, core) not in seen):
                        seen.add((phys, core))
                        cpus.append(cpu)
                    cpu = phys = core = None
    except OSError:
        pass
    return cpus or list(range(max(1, (os.cpu_count() or 2) // 2)))


def cgroup_cpu_quota():
    """CPUs' worth of time the container may use per period (cgroup v2 cpu.max / v1 cfs quota); None: unlimited / unknown."""
    try:
        with open("/sys/fs/cgroup/cpu.max") as fh:
            q, per = fh.read().split()[:2]
            return None if q == "max" else float(q) / float(per)
    except (OSError, ValueError):
        pass
    try:
        with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as fq, open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as fp:
            q, per = float(fq.read()), float(fp.read())
            return None if q <= 0 else q / per
    except (OSError, ValueError):
        return None


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    # (--rows / --cols: the spellings that survive `python -m torch.distributed.run`, whose own parser claims "--m" as an
    #  abbreviation of --master-addr / --max-restarts / ... even behind the script name)
    ap.add_argument("--m", "--rows", dest="m", type=int, default=10_000_000)
    ap.add_argument("--n", "--cols", dest="n", type=int, default=10_000_000)
    ap.add_argument("--nnz-per-row", type=int, default=10)
    ap.add_argument("--seed", type=int, default=12345)
    ap.add_argument("--workload", choices=["random", "pagerank", "l1svm"], default="random",
                    help="random: BASELINE configs[4] (default); pagerank: configs[2]; l1svm: configs[3]")
    ap.add_argument("--pagerank-nodes", type=int, default=1_000_000)
    ap.add_argument("--no-other-configs", action="store_true",
                    help="skip the configs[2]/configs[3] measurements appended at N=1")
    ap.add_argument("--shards", type=int, default=0,
                    help="dev: run K row shards inside this one process on the one GPU (peer-kernel back end); "
                         "timings then show the sharded pipeline's total work, not a multi-GPU rate")
    ap.add_argument("--per-step-calls", action="store_true",
                    help="one host call per take_step in the timed region (pdhg_take_step_adaptive) instead of one "
                         "call for the K steps (pdhg_take_steps_adaptive, what optimize() issues between evaluations)")
    ap.add_argument("--plain-launches", action="store_true",
                    help="measurement aid: every kernel as its own launch, in stream order (PDHG_GRAPH=0), also in the timed "
                         "region -- the same kernels the one-launch paths run.  A rocprofv3 kernel trace of this command "
                         "attributes time to each product's kernels without the graph's concurrent branches or the "
                         "persistent trial kernel (profiles/r03_<workload>_rocprof_summary.json)")
    ap.add_argument("--profile-steps", type=int, default=30)
    ap.add_argument("--cpu-baseline-seconds", type=float, default=12.0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-self-profile", action="store_true",
                    help="do not run the short child runs under rocprofv3 (kernel trace + two --pmc passes) that fill "
                         "roofline.kernel_ms_rocprof / roofline.traffic from THIS run (tools/selfprof.py); the committed "
                         "files under profiles/ are quoted instead")
    ap.add_argument("--no-vendor", action="store_true", help="skip the rocSPARSE comparator (roofline.vendor_spmv_ms)")
    ap.add_argument("--dist-overlap", action="store_true",
                    help="N > 1: run the exchange overlapped with the products -- xbar's all-gather in column chunks beside A_p xbar "
                         "(PDHG_DIST_AG_OVERLAP=1) and per-slice reductions beside A_p'y' (PDHG_DIST_OVERLAP=1).  Off by default: "
                         "both have only ever run over the test transport")
    ap.add_argument("--full-line", action="store_true",
                    help="tools (tools/rocprof_summary.py, tools/pmc_traffic.sh): print the WHOLE record as the stdout line instead of "
                         "the compact one (never what the driver runs)")
    ap.add_argument("--no-details", action="store_true",
                    help="do not write bench_details.json (the short child runs of tools/selfprof.py: the parent's record is the one kept)")
    ap.add_argument("--replay", metavar="DETAILS_JSON", default=None,
                    help="dev / CPU test: no measurement -- read a whole record (a bench_details.json) and print the compact line "
                         "for it through the same final-print path")
    ap.add_argument("--no-ceiling", action="store_true",
                    help="skip the sweep-pattern probe behind roofline.ceiling_frac (pdhg_measure_sweep_ceiling)")
    return ap.parse_args()


def make_problem(args, workload):
    from firstorderlp_jl_amd.generators import l1_svm_rcv1_like_lp, pagerank_lp, random_lp
    if workload == "pagerank":
        n = args.pagerank_nodes
        return pagerank_lp(n, 4 * n, 0.99, seed=0), (
            f"PageRank LP (generate_pagerank_lp.jl model) nodes={n} approx_edges={4 * n} "
            "damping=0.99 (BASELINE configs[2])")
    if workload == "l1svm":
        return l1_svm_rcv1_like_lp(seed=0), (
            "SUBSTITUTE for LIBSVM rcv1 (not available offline): L1-SVM LP (generate_l1_svm_lp.jl model) on SYNTHETIC "
            "rcv1.binary-shaped data 20242 x 47236, ~74 nnz/row, lambda=1 (BASELINE configs[3])")
    p = random_lp(args.m, args.n, args.nnz_per_row, args.seed)
    A = p.constraint_matrix
    return p, f"random LP m={A.shape[0]} n={A.shape[1]} nnz={A.nnz} seed={args.seed} (BASELINE configs[4])"


def initial_scalars(problem):
    import numpy as np
    step0 = 1.0 / float(np.abs(problem.constraint_matrix.data).max())          # pdhg.jl:823
    cn = float(np.sqrt(np.sum(problem.objective_vector ** 2)))
    bn = float(np.sqrt(np.sum(problem.right_hand_side ** 2)))
    return step0, (cn / bn if cn > 0 and bn > 0 else 1.0)                      # saddle_point.jl:1049


def shard_for_rank(args, workload, ctx):
    """Rank 0 generates the LP, computes the row partition and writes one file per rank;
    every rank then loads only its own rows.  Returns (shard dict for
    make_row_shard_hip_engine, meta dict with the global sizes and start scalars)."""
    import pickle
    import tempfile
    import numpy as np
    from firstorderlp_jl_amd import HipPdhgEngine
    from firstorderlp_jl_amd.distributed import row_shard_of
    dist, rank, world = ctx["dist"], ctx["rank"], ctx["world"]
    # where the slices travel: shared memory if it is large enough (a container's /dev/shm may be 64 MB), else a
    # temporary directory on disk -- every rank evaluates the same rule on the same machine
    need = 20 * (args.m if workload == "random" else args.pagerank_nodes) * max(args.nnz_per_row, 12) + 40 * args.n * world + (64 << 20)
    base = tempfile.gettempdir()
    for cand in ("/dev/shm", tempfile.gettempdir(), ROOT):
        try:
            st = os.statvfs(cand)
            if os.access(cand, os.W_OK) and st.f_bavail * st.f_frsize > need:
                base = cand
                break
        except OSError:
            continue
    tag = os.environ.get("MASTER_PORT", "0")
    path = lambda r: os.path.join(base, f"pdhg_bench_{tag}_{workload}_rank{r}.pkl")   # noqa: E731
    if rank == 0:
        problem, wl = make_problem(args, workload)
        A = problem.constraint_matrix
        step0, pw0 = initial_scalars(problem)
        meta = {"wl": wl, "m": A.shape[0], "n": A.shape[1], "nnz": int(A.nnz), "step0": step0, "pw0": pw0}
        bounds = HipPdhgEngine.partition_rows(A, world)
        problem.constraint_matrix = A.tocsr()            # one conversion; row slices are then cheap
        for r in range(world):
            with open(path(r) + ".tmp", "wb") as fh:
                pickle.dump((row_shard_of(problem, bounds, r), meta), fh, protocol=pickle.HIGHEST_PROTOCOL)
            os.replace(path(r) + ".tmp", path(r))
        del problem
    dist.barrier()
    with open(path(rank), "rb") as fh:
        shard, meta = pickle.load(fh)
    dist.barrier()
    if rank == 0:
        for r in range(world):
            try:
                os.remove(path(r))
            except OSError:
                pass
    return shard, meta


def measure(args, workload, ctx, steps, warmup, cpu_seconds, with_socket=True):
    """Build the engine for `workload`, time `steps` adaptive take_steps, profile
    the kernels with HIP events, time the CPU restatement.  Returns a dict."""
    import numpy as np
    import torch
    from firstorderlp_jl_amd import _lib
    from firstorderlp_jl_amd.primal_dual_hybrid_gradient import (
        AdaptiveStepsizeParams, PdhgSolverState, take_step, take_steps)
    pkg, dist, rank, world, local_rank = ctx["pkg"], ctx["dist"], ctx["rank"], ctx["world"], ctx["local_rank"]

    t0 = time.time()
    problem = None
    if dist is not None:
        # N > 1: the LP is generated ONCE (rank 0), cut by the library's own row partition
        # (pdhg_partition_rows) and every rank ingests only ITS rows (pdhg_create_dist_rows);
        # the slices travel through files in shared memory
        shard, meta = shard_for_rank(args, workload, ctx)
        wl, m, n, nnz, step0, pw0 = (meta[k] for k in ("wl", "m", "n", "nnz", "step0", "pw0"))
    else:
        problem, wl = make_problem(args, workload)
        A = problem.constraint_matrix
        m, n, nnz = A.shape[0], A.shape[1], int(A.nnz)
        step0, pw0 = initial_scalars(problem)
    t_gen = time.time() - t0

    t0 = time.time()
    if dist is not None:
        from firstorderlp_jl_amd.distributed import make_row_shard_hip_engine
        eng = make_row_shard_hip_engine(shard, device_id=local_rank)
        del shard
        parallelism = (f"row-partition x{world}, rank-local ingest, RCCL reduce-scatter/all-gather inside the library "
                       f"({pkg.HipPdhgEngine.rccl_info()['path']})")
    elif args.shards > 0:
        eng = pkg.HipPdhgEngine.from_problem(problem, device_ids=[local_rank] * args.shards)
        parallelism = f"{args.shards} row shards inside one process on ONE GPU (peer-kernel back end; dev mode)"
    else:
        eng = pkg.HipPdhgEngine.from_problem(problem, device_id=local_rank)
        parallelism = "single GPU"
    t_create = time.time() - t0

    state = PdhgSolverState(eng, step_size=step0, primal_weight=pw0)
    policy = AdaptiveStepsizeParams(0.3, 0.6)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def run_steps(k):
        # optimize() runs the take_steps between two termination evaluations as one
        # library call (take_steps -> pdhg_take_steps_adaptive); so does the timed region.
        # --per-step-calls: one host call per take_step instead.
        if args.per_step_calls:
            for _ in range(k):
                take_step(policy, state)
            return
        done = 0
        while done < k:
            done += take_steps(policy, state, k - done)

    run_steps(warmup)
    barrier()
    trials0 = state.total_number_iterations
    t0 = time.perf_counter()
    run_steps(steps)
    barrier()
    elapsed = time.perf_counter() - t0
    trials = state.total_number_iterations - trials0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    ms_per_step = 1e3 * elapsed / steps
    value = steps / elapsed
    # the contract times EXACTLY the K steps asked for; when K is small (the driver's 20: 32 ms on config S) the line
    # also carries the rate over 200 further steps, timed the same way
    steady = None
    if steps < 100 and dist is None and not args.no_self_profile:      # (not in the short child runs of tools/selfprof.py)
        barrier()
        t0 = time.perf_counter()
        run_steps(200)
        barrier()
        steady = {"steps": 200, "value": round(200 / (time.perf_counter() - t0), 3), "unit": "iterations/s",
                  "note": "the next 200 take_steps after the K timed ones, same timing"}

    # host-side cost of a trial (groups: issuing threads; one-launch path: launch + wait), timed region + warm-up
    issue_stats = None
    try:
        tr, t_issue, t_wait = eng.host_issue_stats()
        if tr > 0:
            issue_stats = {"trials": tr, "issue": round(1e6 * t_issue / tr, 2), "wait_for_result": round(1e6 * t_wait / tr, 2),
                           "shard_threads": os.environ.get("PDHG_SHARD_THREADS", "1") != "0"}
    except Exception:      # measurement extra
        pass

    # ---- roofline of the dominant kernel: HIP events on the engine's stream
    # Three segments of profile_steps / 3 take_steps, each kernel's average launch duration per segment, the MEDIAN
    # segment reported: a bracket is (event, launch, event) issued by a host thread that the container's CPU quota may stall
    # between the first event and the launch -- one such stall (20 ms, seen once in round 5: 1.395 ms "average" where
    # rocprofv3 and every other run say 0.77) must not halve the roofline figure.  `avg_ms_segments` keeps all three.
    kernels = {}
    segments = 3 if args.profile_steps >= 9 else 1
    seg_rows, prof_trials, totals = [], 0, {}
    for seg in range(segments):
        eng.profile_enable(True)
        trials_before = state.total_number_iterations
        for _ in range(args.profile_steps // segments):
            take_step(policy, state)
        prof_trials += state.total_number_iterations - trials_before
        row = {}
        for kid in range(_lib.K_COUNT):
            cnt, ms = eng.profile_read(kid)
            if cnt:
                row[kid] = (cnt, ms)
                totals[kid] = totals.get(kid, 0.0) + ms
        seg_rows.append(row)
    for kid in sorted(set().union(*[set(r) for r in seg_rows]) if seg_rows else []):
        means = sorted(r[kid][1] / r[kid][0] for r in seg_rows if kid in r)
        cnt = sum(r[kid][0] for r in seg_rows if kid in r)
        byts = eng.kernel_algorithmic_bytes(kid)
        avg_ms = means[len(means) // 2]
        kernels[eng.kernel_name(kid)] = {
            "launches": cnt, "avg_ms": round(avg_ms, 5), "avg_ms_segments": [round(v, 5) for v in means],
            "algorithmic_bytes": byts, "achieved_GBps": round(byts / (avg_ms * 1e-3) / 1e9, 1),
            "launches_per_trial": round(cnt / max(prof_trials, 1), 2)}
    eng.profile_enable(False)
    def _cost(k):
        row = kernels.get(eng.kernel_name(k))
        return row["avg_ms"] * row["launches"] if row else 0.0
    dom = max((_lib.K_SPMV_DUAL, _lib.K_SPMV_ATY), key=_cost)
    # --profile-steps 0 (counter passes, timeline traces): no per-kernel events, no roofline object
    dk = kernels.get(eng.kernel_name(dom), {"achieved_GBps": 0.0, "avg_ms": None,
                                            "algorithmic_bytes": eng.kernel_algorithmic_bytes(dom)})
    # HBM bytes per launch from hardware counters.  PMC passes cannot run inside this
    # process; the figure is the committed result of `rocprofv3 --pmc` runs of THIS
    # command on the same workload and kernel (tools/pmc_traffic.sh), keyed by both.
    traffic, traffic_source = None, None
    if dist is None and args.shards == 0:
        try:
            with open(os.path.join(ROOT, PMC_TRAFFIC_FILE)) as fh:
                table = json.load(fh)
            traffic = table.get(f"{eng.kernel_name(dom)}@m={m},n={n},nnz={nnz}")
            if traffic is not None:
                traffic_source = f"{PMC_TRAFFIC_FILE} ({table.get('_round', 'offline')} rocprofv3 --pmc passes of this command)"
        except OSError:
            pass
    # What a HIP-event bracket holds besides kernel time: an EMPTY launch between two events reads ~9 us on this box,
    # real kernels more (kernel arguments, dispatch ramp) -- noise for a 0.75 ms sweep, a third to a half of a small LP's
    # 15-30 us products.  The bracket stays `avg_launch_ms` / `frac` (the contract); beside it the line quotes the kernel
    # durations rocprofv3 measured for the SAME command (committed summary, like the PMC traffic): `kernel_ms_rocprof`.
    overhead = None
    try:
        o1, o2 = eng.measure_launch_overhead(20)
        overhead = {"empty_launch_between_two_events_ms": round(o1, 5), "each_further_empty_launch_ms": round(o2, 5),
                    "launches_in_group": len(eng.kernel_name(dom).split(" + "))}
    except Exception:    # measurement extra
        pass
    rocprof_ms, rocprof_source = None, None
    if dist is None and args.shards == 0:
        fname = os.path.join("profiles", {"random": "r06_rocprof_summary.json", "pagerank": "r06_pagerank_rocprof_summary.json",
                                          "l1svm": "r06_l1svm_rocprof_summary.json"}[workload])
        try:
            with open(os.path.join(ROOT, fname)) as fh:
                for label, prod in json.load(fh).get("products", {}).items():
                    if label.split(" @ ")[0] == eng.kernel_name(dom) and prod.get("algorithmic_bytes") == dk["algorithmic_bytes"]:
                        rocprof_ms = round(prod["sum_avg_us_per_product"] * 1e-3, 5)
                        rocprof_source = f"{fname} (rocprofv3 --kernel-trace --stats of this command{' --plain-launches' if workload != 'random' else ''}, kernels of the product summed)"
        except (OSError, ValueError):
            pass
    try:     # the box's own streaming ceiling next to the spec figure
        triad = round(eng.measure_triad(1 << 26, 5), 1)
    except Exception:   # measurement extra
        triad = None
    roofline = {"bound": "hbm", "kernel": eng.kernel_name(dom),
                "kernel_is_group": " + " in eng.kernel_name(dom),
                "achieved": dk["achieved_GBps"], "peak": HBM_PEAK_GBS,
                "unit": "GB/s", "frac": round(dk["achieved_GBps"] / HBM_PEAK_GBS, 4),
                "peak_measured_triad": triad,
                "traffic": traffic, "traffic_source": traffic_source,
                "avg_launch_ms": dk["avg_ms"],
                "avg_launch_ms_segments": dk.get("avg_ms_segments"),
                "event_bracket_overhead": overhead,
                "kernel_ms_rocprof": rocprof_ms, "rocprof_source": rocprof_source,
                "frac_rocprof": round(dk["algorithmic_bytes"] / (rocprof_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if rocprof_ms else None,
                "algorithmic_bytes_per_launch": dk["algorithmic_bytes"],
                "note": "a random 8-byte gather per nonzero bounds this kernel (L2 request path), not HBM "
                        "streaming: DESIGN.md section 4.  `kernel` lists the launches of one fused product as rocprofv3 "
                        "prints them (column-slab passes, long-row pair: joined by ' + '); avg_launch_ms (the median of three segments' "
                        "average launch durations, avg_launch_ms_segments: a host stall inside one event bracket must not move it) brackets the whole "
                        "group; kernel_ms_rocprof / traffic are measured by child runs of this script under rocprofv3 (tools/selfprof.py)"}

    # ---- what the sweep's access pattern reaches on this box with nothing else in the kernel:
    # the product kernel's time against it is `ceiling_frac`, measured in this run beside the 8 TB/s figure
    if (dist is None and args.shards == 0 and not args.no_ceiling and "spmv_tiled_kernel" in eng.kernel_name(dom)
            and dk.get("avg_ms")):
        try:
            rows_, cols_ = (m, n) if dom == _lib.K_SPMV_DUAL else (n, m)
            probe = eng.measure_sweep_ceiling(rows_, cols_, nnz, 3)
            kernel_rate = nnz / (dk["avg_ms"] * 1e-3) / 1e9
            roofline["ceiling"] = dict(probe, kernel_Ggathers_per_s=round(kernel_rate, 1),
                                       what="pdhg_measure_sweep_ceiling: the product's geometry (waves, column tiles, pacing "
                                            "barriers, 12-byte entry streams, one 8-byte gather per entry) WITHOUT accumulators, "
                                            "row logic or epilogue; all_hit_window: the SAME geometry with every gather inside one resident "
                                            "window and no barriers (what tile switches and pacing cost; the chip's all-hit rate at full "
                                            "occupancy is higher: 220-250 G/s, NOTEBOOK.md section 4)")
            for k in ("pattern_Ggathers_per_s", "pattern_ms", "all_hit_window_Ggathers_per_s"):
                roofline["ceiling"][k] = round(roofline["ceiling"][k], 3)
            roofline["ceiling_frac"] = round(kernel_rate / probe["pattern_Ggathers_per_s"], 4)
            roofline["ceiling_frac_all_hit"] = round(kernel_rate / probe["all_hit_window_Ggathers_per_s"], 4)
        except Exception as exc:      # measurement extra
            roofline["ceiling"] = {"error": repr(exc)}

    # ---- the vendor's CSR SpMV on the same device and the same matrix, as the INDEPENDENT comparator of the product kernel
    # (rocSPARSE, best of its four CSR algorithms, preprocessing apart; tools/vendor_spmv.py -- a measurement aid the
    # package never loads).  The vendor kernel computes y = A x alone; the product kernel's time includes the fused epilogue.
    if dist is None and args.shards == 0 and not args.no_vendor and problem is not None:
        try:
            from tools import vendor_spmv
            M = problem.constraint_matrix.tocsr() if dom == _lib.K_SPMV_DUAL else problem.constraint_matrix.T.tocsr()
            vs = vendor_spmv.time_csr(M, reps=10, check=False)
            del M
            roofline["vendor_spmv_ms"] = vs.get("best_ms")
            roofline["vendor_spmv"] = {"library": "rocSPARSE rocsparse_spmv (CSR, 32-bit indices, fp64)", "best_algorithm": vs.get("best_alg"),
                                       "ms_by_algorithm": {k: v for k, v in vs.items() if k in vendor_spmv.ALGS},
                                       "product_over_vendor": round(dk["avg_ms"] / vs["best_ms"], 3) if dk.get("avg_ms") and vs.get("best_ms") else None,
                                       "note": "the product kernel's time includes the fused dual step / interaction sums; the vendor's is "
                                               "y = A x alone, kernel time over 10 back-to-back products (no event-bracket overhead)"}
        except Exception as exc:      # measurement extra
            roofline["vendor_spmv"] = {"error": repr(exc)}

    # ---- CPU baseline: the literal single-thread restatement, bounded sample
    cpu_baseline = cpu_socket = None
    if rank == 0 and dist is None and cpu_seconds > 0:
        from oracle.oracle import OracleState
        st = OracleState(m, n, A.indptr, A.indices, A.data,
                         problem.objective_vector, problem.right_hand_side,
                         problem.variable_lower_bound, problem.variable_upper_bound,
                         problem.num_equalities)
        st.step_size, st.primal_weight = step0, pw0
        t0 = time.perf_counter()
        its = 0
        while its < 3 or (time.perf_counter() - t0 < cpu_seconds and its < 1000):
            st.take_step_adaptive(0.3, 0.6)
            its += 1
        dt = time.perf_counter() - t0
        cpu_baseline = {"value": round(its / dt, 4), "unit": "iterations/s",
                        "cores": 1, "kind": "port",
                        "sample": f"first {its} adaptive take_step calls on the same LP "
                                  f"({st.total_number_iterations} trials), oracle/pdhg_oracle.c, "
                                  f"1 thread of {os.cpu_count()} host cores"}
        st.close()
        if with_socket:
            # second CPU comparator: the same step with OpenMP on ONE socket (oracle/pdhg_cpu_omp.c: 32-bit indices, one
            # contiguous row range per thread cut at equal nonzeros, first-touch placement).  Variants timed one after the
            # other on the same LP -- one thread per physical core / every hardware thread of the socket, gathers with and
            # without software prefetch -- and the FASTEST is the figure (a comparator worth the name)
            try:
                from oracle.oracle import OmpCpuState
                cores, smt = socket0_cores(), socket0_cores(all_threads=True)
                # a container's CPU quota (cgroup cpu.max) caps the CPU TIME whatever the thread count: 64 pinned threads
                # under a 16-CPU quota are throttled to a quarter of the wall clock and run SLOWER than 16 (measured on the
                # GPU box, tools/omp_scale.py: 1 / 4 / 16 / 32 / 64 threads = 3.8 / 15.6 / 49.1 / 44.5 / 22.5 it/s on a
                # 4M x 4M LP).  The comparator therefore uses as many of the socket's physical cores as the quota pays for,
                # says so, and never claims more cores than it had.
                quota = cgroup_cpu_quota()
                usable = cores if quota is None else cores[:max(1, min(len(cores), int(quota)))]
                variants = [(f"{len(usable)} physical cores", usable, 0), (f"{len(usable)} physical cores, prefetch 16", usable, 16)]
                if len(usable) == len(cores) and len(smt) > len(cores):
                    variants += [("all hardware threads", smt, 0), ("all hardware threads, prefetch 16", smt, 16)]
                elif len(usable) > 2:
                    variants += [(f"{len(usable) // 2} physical cores", usable[:len(usable) // 2], 0)]
                budget = max(1.2 * cpu_seconds, 6.0) / len(variants)
                tried, best = [], None
                for label, cpus, pf in variants:
                    om = OmpCpuState(m, n, A.indptr, A.indices, A.data,
                                     problem.objective_vector, problem.right_hand_side,
                                     problem.variable_lower_bound, problem.variable_upper_bound,
                                     problem.num_equalities, cpus=cpus)
                    om.set_scalars(step0, pw0)
                    om.set_prefetch(pf)
                    for _ in range(3):
                        om.take_step_adaptive(0.3, 0.6)
                    tr0 = om.total_number_iterations
                    t0 = time.perf_counter()
                    its = 0
                    min_steps = 50 if nnz >= 50_000_000 else 200
                    while its < 5 or ((time.perf_counter() - t0 < budget or its < min_steps) and its < 20000
                                      and time.perf_counter() - t0 < 3 * budget):
                        om.take_step_adaptive(0.3, 0.6)
                        its += 1
                    dt = time.perf_counter() - t0
                    trials_cpu = om.total_number_iterations - tr0
                    rec = {"variant": label, "threads": om.threads(), "value": round(its / dt, 4), "steps": its,
                           "index_bytes": om.index_bytes(),
                           "effective_GBps": round(om.bytes_per_trial() * trials_cpu / dt / 1e9, 1)}
                    om.close()
                    tried.append(rec)
                    if best is None or rec["value"] > best["value"]:
                        best = rec
                cpu_socket = {"value": best["value"], "unit": "iterations/s", "cores": best["threads"],
                              "kind": "port-openmp", "effective_GBps": best["effective_GBps"],
                              "effective_GBps_note": "bytes one trial must stream (both matrix copies with their index arrays + "
                                                     "8(13n+6m) of vectors) x trials / time: what the socket's memory system delivered, "
                                                     "random gathers included",
                              "variants": tried,
                              "host": {"cpu_count": os.cpu_count(), "socket0_physical_cores": len(cores),
                                       "cgroup_cpu_quota_cpus": quota,
                                       "note": None if quota is None or quota >= len(cores) else
                                       f"the container may use {quota:g} CPUs' worth of time (cgroup cpu.max): the comparator ran on "
                                       f"{len(usable)} cores, NOT on the {len(cores)}-core socket the north star names; the socket would "
                                       "be faster by at most the ratio of the core counts (memory bandwidth permitting)"},
                              "sample": f"{best['steps']} adaptive take_step calls on the same LP, oracle/pdhg_cpu_omp.c "
                                        f"({best['variant']}; socket 0 has {len(cores)} physical cores, {len(smt)} hardware threads, "
                                        f"CPU quota {quota if quota is not None else 'none'}), "
                                        "the fastest of the variants listed"}
            except Exception as exc:   # measurement extra: never fail the bench line for it
                cpu_socket = {"error": repr(exc)}

    # The north star's ">= 10x the single-socket CPU" is per SOCKET: when the OpenMP run succeeded it is the line's
    # cpu_baseline (cores = the threads used), with the single-thread figure -- the reference itself is single-threaded
    # Julia -- nested inside; both are the C restatement ("port"), never the reference (no Julia on the box).
    single_thread = cpu_baseline
    if cpu_socket and "value" in cpu_socket and cpu_baseline:
        cpu_baseline = dict(cpu_socket, kind="port", single_thread=single_thread)
    b_pair = 24 * nnz + 16 * (m + n) + 4 * (m + n + 2)
    b_iter = b_pair + 8 * (13 * n + 6 * m)
    out = {
        "value": round(value, 3), "unit": "iterations/s", "steps": steps, "warmup": warmup,
        "ms_per_step": round(ms_per_step, 5),
        "config": {"id": {"random": "configs[4]", "pagerank": "configs[2]", "l1svm": "configs[3]"}[workload],
                   "workload": wl + ", adaptive step, zero start, no restarts/rescaling",
                   "m": m, "n": n, "nnz": nnz, "parallelism": parallelism},
        "trials_per_step": round(trials / steps, 4),
        "steady_rate": steady,
        "whole_iteration_GBps": round(b_iter * (trials / steps) / (ms_per_step * 1e-3) / 1e9, 1),
        "roofline": roofline, "cpu_baseline": cpu_baseline, "cpu_baseline_socket": cpu_socket,
        "kernels": kernels,
        "layout": {k: v for k, v in eng.layout_info().items() if "tiled" in k or "tile_cols" in k or "slab" in k or "graph" in k},
        "layout_products": [eng.kernel_name(_lib.K_SPMV_DUAL), eng.kernel_name(_lib.K_SPMV_ATY)],
        "layout_choices": eng.layout_describe(),      # incl. what pdhg_create settled by timing (pdhg_layout_describe)
        "launch_path": ("one workgroup for the whole batch of steps, vectors in LDS (small_lp_steps_kernel)"
                        if eng.layout_info().get("small_lp") and not args.per_step_calls else
                        "several take_steps per launch of one persistent kernel (steps_kernel)"
                        if eng.layout_info().get("device_loop") and not args.per_step_calls else
                        {2: "one persistent kernel per trial (trial_kernel)", 1: "one HIP-graph launch per trial",
                         0: "separate launches"}.get(eng.layout_info().get("trial_graph"), "?")),
        "host_calls": "one per take_step (pdhg_take_step_adaptive)" if args.per_step_calls
                      else "one for the K steps (pdhg_take_steps_adaptive, as optimize() issues them between evaluations)",
        "setup_sec": {"generate": round(t_gen, 1), "create_upload": round(t_create, 1)},
    }
    if issue_stats is not None:
        out["host_us_per_trial"] = issue_stats
    if dist is not None or args.shards == 0:
        out["scaling_model"] = scaling_model(m, n, nnz, world, trials / steps)
        if dist is None:
            out["scaling_model"]["at_2_4_8_gpus"] = {str(p): {k: scaling_model(m, n, nnz, p, trials / steps)[k]
                                                             for k in ("predicted_it_per_s", "predicted_speedup", "link_floor_ms_per_trial")}
                                                    for p in (2, 4, 8)}
    # where a trial's time goes INSIDE the persistent kernel (stream-layout LPs on the one-launch paths): a second, traced
    # engine on the same LP (PDHG_COOP_TRACE=1: phase-boundary stamps of the 100 MHz wall clock per workgroup), 300 steps
    if dist is None and args.shards == 0 and out["layout"].get("trial_graph") == 2 and problem is not None and not args.per_step_calls:
        try:
            os.environ["PDHG_COOP_TRACE"] = "1"
            eng2 = pkg.HipPdhgEngine.from_problem(problem, device_id=local_rank)
            st2 = PdhgSolverState(eng2, step_size=step0, primal_weight=pw0)
            done = 0
            while done < 300:
                done += take_steps(policy, st2, 300 - done)
            tl = eng2.trial_timeline()
            eng2.close()
            if tl:
                tl["note"] = ("last trial of a traced run of 300 steps (pdhg_trial_timeline): per-workgroup durations of the phases "
                              "between the grid barriers; the stamps themselves cost ~1 us per trial")
                out["trial_timeline"] = tl
        except Exception as exc:      # measurement extra
            out["trial_timeline"] = {"error": repr(exc)}
        finally:
            os.environ.pop("PDHG_COOP_TRACE", None)
    if out["layout"].get("trial_graph") == 2:
        out["kernels_note"] = ("the timed region runs one trial as ONE persistent kernel (trial_kernel: the same device "
                               "functions as the separate kernels between two grid barriers); the per-kernel figures are "
                               "HIP-event brackets around the separate launches of a profiling pass (launch latency included), "
                               "so their sum exceeds ms_per_step")
    elif out["layout"].get("trial_graph"):
        out["kernels_note"] = ("the timed region launches one trial as ONE HIP graph (long-row kernels on a parallel "
                               "branch, no result copy); the per-kernel figures are HIP-event brackets around the plain "
                               "launches of a separate profiling pass (launch latency included, branches serialised), so "
                               "their sum can exceed ms_per_step")
    if single_thread:
        out["speedup_vs_cpu_port"] = round(value / single_thread["value"], 1)
    if cpu_socket and "value" in cpu_socket:
        out["speedup_vs_cpu_socket"] = round(value / cpu_socket["value"], 1)
    out["_selfprof"] = {"workload": workload, "product": eng.kernel_name(dom), "algorithmic_bytes": dk["algorithmic_bytes"]}
    eng.close()
    return out


# ---- what N GPUs should deliver, stated BEFORE the first multi-GPU run so that a SCALE record can be checked against it.
XGMI_LINKS = 7                    # per GPU, one to each peer of an 8-GPU node (point-to-point, no switch)
XGMI_LINK_GBPS_PEAK = 76.8        # per link and direction (153.6 GB/s bidirectional: 7 links = 1075 GB/s per GPU)
XGMI_LINK_GBPS_ASSUMED = 55.0     # what a collective is assumed to sustain per link and direction (NOTEBOOK section 5)
GATHER_RATE_G = 130.0             # nonzeros per ns the product kernels reach on one GPU (config S, measured rounds 1-4)
STREAM_TBPS = 5.5                 # the box's streaming triad


def scaling_model(m, n, nnz, world, trials_per_step=1.0, ag_chunks=4):
    """Strong scaling of one adaptive take_step on the 1-D row partition: per trial every GPU runs
    1/P of the kernels, receives (P-1) slices of xbar (all-gather) and sends (P-1) slices of A_p'y_p (reduce-scatter);
    on the fully connected xGMI node every slice has a link of its own, so a collective costs one slice over one link,
    and all 7 links are only busy at P = 8.  Four forms of the same trial: nothing overlapped (the library's default on
    RCCL), the all-gather cut into `ag_chunks` column chunks beside A_p xbar (PDHG_DIST_AG_OVERLAP=1, round 6), that and
    the per-slice reductions beside A_p'y' (PDHG_DIST_OVERLAP=1), and the ideal (every byte hidden)."""
    S = max(16, ((n + world - 1) // world + 15) // 16 * 16)       # slice stride (init_group_geometry)
    slice_bytes = 8 * S
    links = min(world - 1, XGMI_LINKS)
    prod_ms = lambda p: 1e3 * (nnz / p) / (GATHER_RATE_G * 1e9)                       # noqa: E731  one product's gathers
    vec_n_ms = lambda p: 1e3 * 8 * 13 * n / p / (STREAM_TBPS * 1e12)                  # noqa: E731  primal step + interaction sums
    vec_m_ms = lambda p: 1e3 * 8 * 6 * m / p / (STREAM_TBPS * 1e12)                   # noqa: E731  the dual step fused into A_p xbar
    kernel_ms = lambda p: 2 * prod_ms(p) + vec_n_ms(p) + vec_m_ms(p)                  # noqa: E731
    one = kernel_ms(1)
    out = {"partition": f"1-D rows x{world}, owned column slices of {S}", "xgmi_bytes_per_trial_per_gpu": {
               "all_gather_xbar_received": (world - 1) * slice_bytes, "reduce_scatter_sent": (world - 1) * slice_bytes,
               "scalars": 8 * 32 * world},
           "links_used": links, "bytes_per_link_per_collective": slice_bytes if world > 1 else 0,
           "link_GBps": {"peak_per_direction": XGMI_LINK_GBPS_PEAK, "assumed": XGMI_LINK_GBPS_ASSUMED},
           "kernel_ms_per_trial": round(kernel_ms(world), 4), "single_gpu_kernel_ms_per_trial": round(one, 4),
           "kernel_model": f"2 x nnz/P at {GATHER_RATE_G} G nonzeros/s (measured, one GPU) + 8(13n+6m)/P bytes at {STREAM_TBPS} TB/s"}
    if world == 1:
        out.update(predicted_ms_per_step=round(one * trials_per_step, 4), predicted_it_per_s=round(1e3 / (one * trials_per_step), 1),
                   predicted_speedup=1.0)
        return out
    floor = 2 * 1e3 * slice_bytes / (XGMI_LINK_GBPS_PEAK * 1e9)            # both collectives at the link's peak
    one_coll = 1e3 * slice_bytes / (XGMI_LINK_GBPS_ASSUMED * 1e9) + 0.015  # one collective: a slice over a link + launch / synchronisation
    coll = 2 * one_coll
    P = world
    pipe = lambda a, b, k: (a + b) / k + (k - 1) / k * max(a, b)           # noqa: E731  two stages cut into k pieces, piece i+1 of one beside piece i of the other
    dual = prod_ms(P) + vec_m_ms(P)
    carry = 1e3 * 16 * (m / P) * (ag_chunks - 1) / (STREAM_TBPS * 1e12)    # the row sums written and read back between the passes
    serial = kernel_ms(P) + coll                                           # csrc/dist.hpp's default on RCCL
    ag_ov = vec_n_ms(P) + pipe(one_coll + 0.01 * (ag_chunks - 1), dual + carry, ag_chunks) + prod_ms(P) + one_coll
    both_ov = vec_n_ms(P) + pipe(one_coll + 0.01 * (ag_chunks - 1), dual + carry, ag_chunks) + pipe(prod_ms(P), one_coll + 0.01 * (P - 1), P)
    overlapped = max(kernel_ms(P), coll) + 0.02                            # every byte hidden behind a kernel or vice versa
    forms = {"no_overlap": serial, "ag_overlap": ag_ov, "ag_and_rs_overlap": both_ov, "full_overlap": overlapped}
    ag_on = os.environ.get("PDHG_DIST_AG_OVERLAP", "0") == "1"
    rs_on = os.environ.get("PDHG_DIST_OVERLAP", "0") == "1"
    as_built = "ag_and_rs_overlap" if ag_on and rs_on else "ag_overlap" if ag_on else "no_overlap"
    out.update(link_floor_ms_per_trial=round(floor, 4), collectives_ms_per_trial=round(coll, 4),
               ag_chunks=ag_chunks, as_built=as_built,
               as_built_note="the form THIS run's environment selects (PDHG_DIST_AG_OVERLAP / PDHG_DIST_OVERLAP; both default off on "
                             "RCCL: the overlapped forms have only ever run over the test transport, bench.py --dist-overlap turns them on)",
               predicted_ms_per_step={k: round(v * trials_per_step, 4) for k, v in forms.items()},
               predicted_it_per_s={k: round(1e3 / (v * trials_per_step), 1) for k, v in forms.items()},
               predicted_speedup=dict({k: round(one / v, 2) for k, v in forms.items()},
                                      at_link_peak_full_overlap=round(one / (max(kernel_ms(P), floor) + 0.02), 2)),
               efficiency={k: round(one / v / P, 3) for k, v in forms.items()},
               note="a 10-per-row LP moves as many bytes over xGMI per trial (16 n (P-1)/P) as through one GPU's HBM share "
                    "(24 nnz / P): the >= 6x target at P = 8 is above what the links allow for this partition (DESIGN.md section 5 "
                    "prices the 2-D alternative: same bytes per link, more phases)")
    return out


# ---- the line the driver parses: compact (<= 4 KB), everything else in bench_details.json (round 5's
# line had grown to 22.9 KB and the driver could not read it)
COMPACT_LIMIT = 4096
DETAILS_FILE = "bench_details.json"
_ROOFLINE_KEYS = ("bound", "kernel", "achieved", "peak", "unit", "frac", "frac_rocprof", "kernel_ms_rocprof", "avg_launch_ms",
                  "algorithmic_bytes_per_launch", "traffic", "traffic_over_algorithmic", "vendor_spmv_ms", "ceiling_frac",
                  "peak_measured_triad", "rocprof_measured_in_this_run")


def _short(text, limit):
    text = str(text)
    return text if len(text) <= limit else text[:limit - 3] + "..."


def _compact_roofline(rf):
    if not isinstance(rf, dict):
        return rf
    out = {k: rf[k] for k in _ROOFLINE_KEYS if k in rf}
    if "kernel" in out:
        out["kernel"] = _short(out["kernel"], 160)
    return out


def _compact_cpu(cb):
    if not isinstance(cb, dict):
        return cb
    out = {k: cb[k] for k in ("value", "unit", "cores", "kind", "effective_GBps") if k in cb}
    if "sample" in cb:
        out["sample"] = _short(cb["sample"], 200)
    st = cb.get("single_thread")
    if isinstance(st, dict) and "value" in st:
        out["single_thread_value"] = st["value"]
    return out


def _compact_model(sm):
    """Five numbers of the scaling model: what N GPUs should deliver, as built / fully overlapped."""
    if not isinstance(sm, dict):
        return None
    out = {}
    for key in ("predicted_it_per_s", "predicted_speedup"):
        if key in sm:
            out[key] = sm[key]
    for key in ("kernel_ms_per_trial", "collectives_ms_per_trial", "link_floor_ms_per_trial", "as_built", "ag_chunks"):
        if key in sm:
            out[key] = sm[key]
    if "at_2_4_8_gpus" in sm:
        out["predicted_speedup_at_2_4_8"] = {p: v.get("predicted_speedup") for p, v in sm["at_2_4_8_gpus"].items()}
    return out


def compact_line(full, details_path=DETAILS_FILE):
    """The ONE stdout line: the contract's keys, `roofline`, `cpu_baseline`, the other configs as one short row each.
    `full` is the whole record (what goes to bench_details.json).  Never longer than COMPACT_LIMIT bytes: optional
    blocks are dropped in a fixed order until it fits."""
    out = {k: full.get(k) for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
                                    "scaling", "vs_baseline", "dtype", "data")}
    cfg = full.get("config") or {}
    out["config"] = {"workload": _short(cfg.get("workload", ""), 220)}
    for k in ("id", "m", "n", "nnz"):
        if k in cfg:
            out["config"][k] = cfg[k]
    if "parallelism" in cfg:
        out["config"]["parallelism"] = _short(cfg["parallelism"], 120)
    if full.get("error"):
        out["error"] = _short(full["error"], 400)
    out["roofline"] = _compact_roofline(full.get("roofline"))
    out["cpu_baseline"] = _compact_cpu(full.get("cpu_baseline"))
    for k in ("speedup_vs_cpu_port", "speedup_vs_cpu_socket", "trials_per_step", "whole_iteration_GBps", "launch_path", "transport"):
        if full.get(k) is not None:
            out[k] = _short(full[k], 100) if isinstance(full[k], str) else full[k]
    if isinstance(full.get("steady_rate"), dict):
        out["steady_rate"] = {k: full["steady_rate"].get(k) for k in ("steps", "value")}
    sm = _compact_model(full.get("scaling_model"))
    if sm:
        out["scaling_model"] = sm
    rows = []
    for leg in full.get("other_configs") or []:
        lcfg = leg.get("config") or {}
        row = {"id": lcfg.get("id"), "workload": _short(lcfg.get("workload", ""), 60)}
        if "error" in leg:
            row["error"] = _short(leg["error"], 120)
        else:
            lrf = leg.get("roofline") or {}
            row.update({k: lcfg.get(k) for k in ("m", "n", "nnz")})
            row.update(value=leg.get("value"), ms_per_step=leg.get("ms_per_step"), steps=leg.get("steps"),
                       kernel=_short(lrf.get("kernel", ""), 90), frac=lrf.get("frac"), frac_rocprof=lrf.get("frac_rocprof"),
                       kernel_ms_rocprof=lrf.get("kernel_ms_rocprof"), vendor_spmv_ms=lrf.get("vendor_spmv_ms"),
                       traffic_over_algorithmic=lrf.get("traffic_over_algorithmic"),
                       speedup_vs_cpu_port=leg.get("speedup_vs_cpu_port"))
        rows.append(row)
    if rows:
        out["other_configs"] = rows
    out["details"] = details_path
    for drop in (None, "scaling_model", "steady_rate", "launch_path", "other_configs", "whole_iteration_GBps", "trials_per_step"):
        if drop is not None:
            out.pop(drop, None)
        line = json.dumps(out, separators=(",", ":"))
        if len(line) <= COMPACT_LIMIT:
            return line
    # still too long: only a pathological kernel name or workload string can do that -- cut them
    out["config"] = {k: (_short(v, 60) if isinstance(v, str) else v) for k, v in out["config"].items()}
    if isinstance(out.get("roofline"), dict) and "kernel" in out["roofline"]:
        out["roofline"]["kernel"] = _short(out["roofline"]["kernel"], 60)
    if isinstance(out.get("cpu_baseline"), dict):
        out["cpu_baseline"].pop("sample", None)
    return json.dumps(out, separators=(",", ":"))


def write_details(full):
    """The whole record (other_configs, scaling_model, ceiling, self_profile, variants, kernels, notes, trial_timeline) next
    to the script; a read-only tree sends it to the temporary directory.  Returns the path written (or None)."""
    import tempfile
    text = json.dumps(full, indent=1)
    for base in (ROOT, tempfile.gettempdir()):
        try:
            path = os.path.join(base, DETAILS_FILE)
            with open(path + ".tmp", "w") as fh:
                fh.write(text + "\n")
            os.replace(path + ".tmp", path)
            return path
        except OSError:
            continue
    return None


def emit(full, real_stdout, details=True, full_line=False):
    """Details to the file and to stderr, then the compact line as the LAST line of stdout."""
    path = write_details(full) if details else None
    name = DETAILS_FILE if path == os.path.join(ROOT, DETAILS_FILE) else (path or "stderr")
    sys.stdout.flush()
    sys.stderr.write(json.dumps(full) + "\n")
    sys.stderr.flush()
    line = json.dumps(full) if full_line else compact_line(full, name)
    if real_stdout is not None:
        os.dup2(real_stdout, 1)
    print(line, flush=True)
    if real_stdout is not None:
        os.dup2(2, 1)


def multi_gpu_failure(args, world, rank, exc, real_stdout):
    """N > 1 has never run in the builder's environment (one GPU per box): whatever fails first -- a device that is not
    there, RCCL initialisation, peer access, a barrier another rank never reaches -- must leave ONE parseable line that
    says so, not a traceback alone or a hang.  The launcher stops every rank as soon as one exits, so the FIRST rank to
    fail prints the line (a marker file per rendezvous port elects it)."""
    import tempfile
    import traceback
    traceback.print_exc()
    first = True
    try:
        os.close(os.open(os.path.join(tempfile.gettempdir(), f"pdhg_bench_failed_{os.environ.get('MASTER_PORT', '0')}_{os.getppid()}"),
                         os.O_CREAT | os.O_EXCL | os.O_WRONLY))
    except OSError:
        first = False
    if first:
        emit({"metric": "pdhg_iterations_per_sec", "value": None, "unit": "iterations/s", "n_gpus": world,
              "steps": args.steps, "warmup": args.warmup, "ms_per_step": None, "higher_is_better": True,
              "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
              "config": {"workload": args.workload},
              "error": f"rank {rank}: {exc!r} (the multi-GPU path failed; nothing was measured)"}, real_stdout)
    os._exit(1)


def main():
    args = parse()
    if args.dist_overlap:
        os.environ["PDHG_DIST_AG_OVERLAP"] = "1"
        os.environ["PDHG_DIST_OVERLAP"] = "1"
    if args.replay:
        with open(args.replay) as fh:
            full = json.load(fh)
        sys.stderr.write(json.dumps(full) + "\n")
        print(compact_line(full, os.path.basename(args.replay)), flush=True)
        return
    if args.plain_launches:
        os.environ["PDHG_GRAPH"] = "0"
    # Native libraries print to stdout too (RCCL's version banner on communicator
    # creation, for one).  The contract is ONE JSON line on rank 0's stdout: send
    # file descriptor 1 to stderr until that line is written.
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    import torch
    import folp_loader
    pkg = folp_loader.load()
    from firstorderlp_jl_amd import _lib

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not os.path.exists(_lib.LIB_PATH) and world == 1:
        _lib.build()       # fresh clone on a 1-GPU box; normally the in-tree library is already there
    if world != args.gpus and world == 1 and args.gpus > 1:
        raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (no CPU fallback)")
    # Which device this rank drives, and what carries the exchange.  Normally rank r drives GPU r over RCCL.  On a box with
    # FEWER GPUs than ranks the run is only possible over the test-only stand-in (PDHG_RCCL_LIB = tests/fake_rccl: real RCCL
    # refuses two ranks on one device): every rank then shares the visible GPUs and the line says so -- a functional run
    # of the one-process-per-GPU route (rank-local ingest, the library's collectives, this harness), NOT a scaling figure.
    device, transport = local_rank, "rccl"
    rccl_lib = os.environ.get("PDHG_RCCL_LIB", "")
    if "fake_rccl" in os.path.basename(rccl_lib):
        ngpu = torch.cuda.device_count()
        device = local_rank % max(ngpu, 1)
        transport = f"fake ({ngpu} GPU)" if ngpu < world else "fake"
    try:
        torch.cuda.set_device(device)
    except Exception as exc:
        if world == 1:
            raise
        multi_gpu_failure(args, world, rank, exc, real_stdout)
    dist = None
    # PDHG_FORCE_DIST=1: take the N > 1 route of this script with ONE rank (gloo group of 1, id
    # broadcast, pdhg_create_dist, 1-rank RCCL communicator) -- how that route is exercised on a
    # 1-GPU box, e.g. under `python -m torch.distributed.run --nproc-per-node 1 ... bench.py`.
    if world > 1 or os.environ.get("PDHG_FORCE_DIST", "0") == "1":
        # harness only (id hand-out, barriers, max over ranks): CPU tensors over gloo
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        import datetime
        # (a rank that fails leaves the others in a barrier: ten minutes, not gloo's default thirty, then the error line below)
        dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(minutes=10))
    ctx = {"pkg": pkg, "dist": dist, "rank": rank, "world": world, "local_rank": device, "transport": transport}

    cpu_s = 0.0 if args.no_cpu_baseline else args.cpu_baseline_seconds
    try:
        head = measure(args, args.workload, ctx, args.steps, args.warmup, cpu_s)
    except Exception as exc:
        if dist is None:
            raise
        multi_gpu_failure(args, world, rank, exc, real_stdout)
    others = []
    if dist is None and not args.no_other_configs and args.workload == "random" and args.shards == 0:
        for wl in ("pagerank", "l1svm"):
            try:
                # an iteration of these LPs takes 0.05-0.2 ms: 2000 timed steps after 300 untimed ones (each leg states
                # its own steps / warmup), so that the figure is the steady rate and not the retry-heavy first steps
                r = measure(args, wl, ctx, max(args.steps, 2000), max(args.warmup, 300), min(cpu_s, 3.0),
                            with_socket=False)
                r["metric"] = "pdhg_iterations_per_sec"
                others.append(r)
            except Exception as exc:     # never lose the headline line
                others.append({"config": {"workload": wl}, "error": repr(exc)})

    # ---- the roofline's rocprofv3 figures from THIS run: short child runs of this script under rocprofv3
    # (tools/selfprof.py); the committed files stay as the fallback, marked as such
    t_prof0 = time.time()
    for leg in [head] + others:
        sp = leg.pop("_selfprof", None)
        if not sp or "roofline" not in leg:
            continue
        rf = leg["roofline"]
        rf["rocprof_measured_in_this_run"] = False
        if rf.get("traffic") is not None:
            rf["traffic_source"] = "committed: " + str(rf.get("traffic_source"))
        if rf.get("kernel_ms_rocprof") is not None:
            rf["rocprof_source"] = "committed: " + str(rf.get("rocprof_source"))
        if dist is not None or args.no_self_profile or args.shards or time.time() - t_prof0 > 240:
            continue
        from tools import selfprof
        extra = ["--m", str(args.m), "--n", str(args.n), "--nnz-per-row", str(args.nnz_per_row), "--seed", str(args.seed),
                 "--pagerank-nodes", str(args.pagerank_nodes)]
        res = selfprof.run(sp["workload"], sp["product"], extra, deadline=t_prof0 + 300)
        rf["self_profile"] = {k: res[k] for k in ("command", "seconds", "error", "failed_passes", "kernels") if k in res}
        if "kernel_ms" in res:
            rf["kernel_ms_rocprof"] = res["kernel_ms"]
            rf["frac_rocprof"] = round(sp["algorithmic_bytes"] / (res["kernel_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
            rf["rocprof_source"] = "measured in this run: rocprofv3 --kernel-trace --stats around a child run of this script (tools/selfprof.py)"
            rf["rocprof_measured_in_this_run"] = True
        if "l2_requests" in res and "kernel_ms" in res:
            # every random gather is one L2 request whether it hits or not: the rate at which the product's kernels make
            # them is what config S, PageRank and the L1-SVM LP have in common
            rf["l2_requests_per_launch"] = res["l2_requests"]
            rf["l2_request_rate_G_per_s"] = round(res["l2_requests"] / (res["kernel_ms"] * 1e-3) / 1e9, 1)
        if "traffic" in res:
            rf["traffic"] = res["traffic"]
            rf["traffic_over_algorithmic"] = round(res["traffic"] / sp["algorithmic_bytes"], 3)
            rf["traffic_source"] = ("measured in this run: two rocprofv3 --pmc passes (TCC_EA0_RDREQ{,_32B,_128B}_sum, TCC_HIT_sum | "
                                    "WRITE_SIZE, TCC_MISS_sum) around child runs of this script, reads by request size + writes, "
                                    "summed over the product's kernels (tools/selfprof.py)")

    if rank == 0:
        out = {"metric": "pdhg_iterations_per_sec", "value": head.pop("value"), "unit": head.pop("unit"),
               "n_gpus": world, "steps": head.pop("steps"), "warmup": head.pop("warmup"),
               "ms_per_step": head.pop("ms_per_step"), "higher_is_better": True, "scaling": "strong",
               "vs_baseline": None, "dtype": "f64", "data": "synthetic"}
        out.update(head)
        if dist is not None:
            out["transport"] = transport
            if transport.startswith("fake"):
                out["transport_note"] = ("test-only stand-in for RCCL (tests/fake_rccl: host-staged exchange between processes that "
                                         "share the visible GPUs): a functional run of the multi-rank route, not a scaling measurement")
        if others:
            out["other_configs"] = others
    if dist is not None:
        dist.destroy_process_group()
    if rank == 0:
        # the compact line is the LAST thing this script writes (the process then exits NORMALLY: rocprofv3 writes its
        # databases from exit handlers, so the child runs of tools/selfprof.py must not be cut short)
        emit(out, real_stdout, details=not args.no_details, full_line=args.full_line)


if __name__ == "__main__":
    main()
