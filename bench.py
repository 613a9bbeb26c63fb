#!/usr/bin/env python3
"""bench.py — headline benchmark: batched iLQR trajectory-iterations per second (BASELINE.json metric).

  python bench.py --gpus N --steps K --warmup W            (N=1 default)
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is one complete pass of the hot path over one batch of synthetic input: the batch's controls are
reset on the device to the initial guess (inputs stay resident in HBM) and the whole batch is solved with
iLQR (rollout -> [expansion -> backward Riccati -> forward line search]* until every trajectory converges).
The workload at N=1 is BASELINE.json configs[1]: Cartpole swing-up (n=4, m=1), N=101 knot points,
batch=1024 trajectories.  Every rank owns its own 1024-trajectory shard (weak scaling; inputs are drawn
from the global trajectory index so shards are disjoint); with N>1 each step ends with the north-star's
RCCL all-gather of the converged trajectories.

The K steps are timed twice: one after the other on one handle ("unpipelined" in the line), and — the line's value — PIPELINED over
four handles per rank (--pipeline): step i on handle i mod 4, every handle holding the same batch, the next solve admitted while the one
in flight drains (to_solve_progress / to_solve_wait_below).  A solve is batch-synchronous and its batch drains unevenly (C2: half of its
300 batch steps serve five trajectories), and every solve of the pipelined pass equals the unpipelined one bit for bit.

value = total trajectory-iterations of all ranks over the timed steps / max-over-ranks wall time between barriers.
roofline: algorithmic bytes (48 064 B per Cartpole trajectory-iteration, split per kernel as
DESIGN.md §4 states) / kernel time from hipEvents recorded on the library's own stream inside the timed
region.  cpu_baseline: the CPU oracle (a port; the Julia reference cannot run here) on the same workload.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))

import numpy as np

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec (MI355X_MICROARCH.md)
FP64_VALU_PEAK_TFLOPS = 78.6  # 256 CU x 4 SIMD x 16 lanes x 2 flop x 2.4 GHz (vector FP64; tools/fp64_issue_probe measures 65)

WORKLOADS = {
    # name: (builder kwargs, description)
    "cartpole": dict(batch=1024, N=101, solver="ilqr",
                     desc="C2 Cartpole swing-up iLQR (n=4,m=1), N=101, batch=1024 per GPU"),
    "quadrotor": dict(batch=4096, N=201, solver="ilqr",
                      desc="C3 Quadrotor point-to-point iLQR (n=13,m=4,ne=12), N=201, batch=4096 per GPU"),
    "quadrotor_altro": dict(batch=8192, N=201, solver="altro", opts={"n_steps": "C5_PN_STEPS"},
                         desc="C5 Quadrotor + GoalConstraint(xf, inds=[1,2,3,8..13]: position + velocities) + NormConstraint(SOC, |u|<=6), "
                              "solved as the reference's stack solves constrained problems: ALTRO = AL-iLQR to 1e-3 + projected-Newton "
                              "polish to constraint_tolerance 1e-6 (n_steps = 8); N=201, batch=8192 per GPU; the metric counts the "
                              "inner iLQR iterations, the time includes the polish"),
    "quadrotor_altro_defaults": dict(batch=8192, N=201, solver="altro", opts={"n_steps": 2, "rho_chol": 1e-2},
                         desc="C5 as quadrotor_altro but with Altro's own polish defaults (n_steps = 2, rho_chol = 1e-2): the fidelity line next to "
                              "the tuned one; N=201, batch=8192 per GPU"),
    "quadrotor_al": dict(batch=8192, N=201, solver="al",
                         desc="C5 without the polish (AL-iLQR run to 1e-6 on its own: the definition of the rounds-1..3 records under this key)"),
}


# pipelined solves: the next solve is admitted when the one in flight has this fraction of its batch still iterating (measured per
# workload: tools/ab/ab_pipeline.py, profiles/r06_ab/)
ADMIT_FRAC = {"quadrotor": 1.0, "quadrotor_altro": 1.0, "quadrotor_altro_defaults": 1.0, "quadrotor_al": 1.0}


def make_solver(T, configs, name, prob):
    W = WORKLOADS[name]
    kw = {k: (getattr(configs, v) if isinstance(v, str) else v) for k, v in W.get("opts", {}).items()}
    return {"ilqr": T.iLQRSolver, "al": T.ALSolver, "altro": T.ALTROSolver}[W["solver"]](prob, **kw)


def build_problem(T, configs, name, batch, b_offset, device, lib):
    if name == "cartpole":
        return configs.cartpole_problem(batch=batch, b_offset=b_offset, device=device, lib=lib)
    if name == "quadrotor":
        return configs.quadrotor_problem(batch=batch, b_offset=b_offset, device=device, lib=lib)
    if name in ("quadrotor_altro", "quadrotor_altro_defaults", "quadrotor_al"):
        return configs.quadrotor_problem(batch=batch, b_offset=b_offset, device=device, lib=lib, constrained=True,
                                         goal_inds=configs.C5_GOAL_INDS)
    raise ValueError(name)


def initial_controls_value(T, prob, name):
    return np.full(prob.m, 0.01) if name == "cartpole" else prob.model.hover_control()


def pmc_traffic(name, batch, phase, build_id):
    """HBM bytes per launch of `phase` from the committed rocprofv3 PMC passes of this same command
    (tools/run_profiles.sh: separate --pmc FETCH_SIZE / WRITE_SIZE runs, FETCH doubled per the gfx950 note of
    MI355X_MICROARCH.md).  PMC collection cannot run inside the timed bench, so the figure is read from profiles/ —
    and only from a profile taken with THIS binary: every *_pmc.json carries the build id of the library it measured
    (`__meta__.build_id`); a profile of another build is refused (traffic = null, the reason in traffic_source).
    Returns (bytes, source, flops)."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_%s_b%d_hbm_traffic_pmc.json" % (name, batch))))
    if not files:
        return None, "no profile committed for this workload", None
    with open(files[-1]) as f:
        tab = json.load(f)
    rel = os.path.relpath(files[-1], ROOT)
    meta = tab.pop("__meta__", {})
    if meta.get("build_id") != build_id:
        return None, "%s is stale: it measured build %s, the loaded library is %s" % (rel, meta.get("build_id"), build_id), None
    steps = sum(v["launches"] for k, v in tab.items() if "k_forward" in k)
    mine = [v for k, v in tab.items() if ("k_" + phase) in k
            or (phase == "forward" and any(s in k for s in ("k_select", "k_accept", "k_outer")))
            or (phase == "expand_backward" and ("k_expand" in k or "k_backward" in k))]
    total = sum(v["hbm_bytes_per_launch_fetch_x2"] * v["launches"] for v in mine)
    if steps == 0 or total == 0:
        return None, rel + " holds no launches of this phase", None
    flops = sum(v.get("fp64_flops_per_launch", 0.0) * v["launches"] for v in mine) / steps
    return total / steps, rel, (flops or None)


def kernel_split_bytes(n, m, ne, N, duals):
    """Per trajectory-iteration algorithmic bytes attributed to each kernel (sums to SURVEY §8d's W)."""
    xu = 8 * (N * n + (N - 1) * m)
    ab = 8 * (N - 1) * (ne * ne + ne * m)
    kd = 8 * (N - 1) * (m * ne + m)
    return {"expand": xu + ab + 8 * duals, "backward": ab + kd, "forward": kd + xu + 8 * duals}


def physical_cores():
    """(physical cores, logical CPUs) this process may run on: distinct (package, core) pairs of /proc/cpuinfo among the CPUs of
    the affinity mask."""
    allowed = os.sched_getaffinity(0) if hasattr(os, "sched_getaffinity") else set(range(os.cpu_count() or 1))
    cores, cpu, phys = set(), None, None
    try:
        for line in open("/proc/cpuinfo"):
            key, _, val = line.partition(":")
            key, val = key.strip(), val.strip()
            if key == "processor":
                cpu, phys = int(val), None
            elif key == "physical id":
                phys = val
            elif key == "core id" and cpu in allowed:
                cores.add((phys, val))
    except OSError:
        pass
    return (len(cores) or len(allowed)), len(allowed)


HOST_CORES = physical_cores()
START_AFFINITY = os.sched_getaffinity(0) if hasattr(os, "sched_getaffinity") else None


def cgroup_cpu_quota():
    """CPU time the container may use, in cores (cgroup v2 cpu.max / v1 cfs quota), or None when unlimited / unknown: a box can
    SHOW 128 cores and grant a dozen — which caps what any number of OpenMP threads delivers."""
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        return None if q == "max" else float(q) / float(per)
    except (OSError, ValueError):
        pass
    try:
        q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
        per = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        return None if q <= 0 else q / per
    except (OSError, ValueError):
        return None


def cpu_baseline(T, configs, name, batch, seconds_budget=20.0):
    """Oracle (port) on the host cores, bounded sample of the same workload.  One OpenMP thread per core the container may
    actually USE: min(physical cores, cgroup CPU quota) — a box can show 128 cores and grant 16, and 128 threads under a 16-core
    quota are throttled by the scheduler.  The one-thread-per-physical-core figure is reported beside it
    (`all_physical_cores`).  Threads pinned (OMP_PLACES=cores, OMP_PROC_BIND=close: set in main() before the OpenMP runtime starts)."""
    from oracle_binding import load_oracle_native, set_threads
    o, flags = load_oracle_native()
    phys, logical = HOST_CORES   # taken at start-up: OMP_PROC_BIND binds the main thread later, which shrinks its own affinity mask
    quota = cgroup_cpu_quota()
    all_threads = max(1, min(o.max_threads(), phys))
    threads = max(1, min(all_threads, int(quota + 0.5))) if quota else all_threads
    sample = min(batch, 1024 if name == "cartpole" else 256 if name == "quadrotor" else 128)
    Solver = lambda pr: make_solver(T, configs, name, pr)
    prob = build_problem(T, configs, name, sample, 0, 0, o)
    u0 = initial_controls_value(T, prob, name)
    solver = Solver(prob)

    def timed(nthreads):
        set_threads(prob, nthreads)
        T.initial_controls(prob, u0)
        solver.solve()                            # warm call: OpenMP team start-up, page faults
        T.initial_controls(prob, u0)
        t0 = time.perf_counter()
        solver.solve()
        return time.perf_counter() - t0

    dt = timed(threads)
    its = solver.total_iterations
    over = None
    if all_threads != threads:
        dt_all = timed(all_threads)
        over = {"threads": all_threads, "value": solver.total_iterations / dt_all,
                "note": "one thread per physical core the box SHOWS: oversubscribes the cgroup quota (the r01-r05 records used this)"}
    # one trajectory on one core: the figure comparable to published single-core solver timings;
    # best of 5 after a warm call (a single cold sample was 10x off in round 1)
    p1 = build_problem(T, configs, name, 1, 0, 0, o)
    set_threads(p1, 1)
    s1 = Solver(p1)
    s1.solve()
    d1 = float("inf")
    for _ in range(5):
        T.initial_controls(p1, u0)
        t1 = time.perf_counter()
        s1.solve()
        d1 = min(d1, time.perf_counter() - t1)
    single = {"value": s1.total_iterations / d1, "cores": 1,
              "sample": f"trajectory 0 alone, {s1.total_iterations} iterations in {d1 * 1e3:.1f} ms (best of 5 after a warm call)"}
    return {"value": its / dt, "unit": "trajectory-iterations/s", "cores": threads, "kind": "port", "build": flags, "single_thread": single,
            "physical_cores": phys, "logical_cpus": logical, "cgroup_cpu_quota_cores": quota,
            "pinning": "OMP_PLACES=cores OMP_PROC_BIND=close, one thread per core of the cgroup quota",
            "parallel_speedup_over_one_thread": (its / dt) / single["value"], "all_physical_cores": over,
            "sample": f"{WORKLOADS[name]['desc']}: first {sample} trajectories of the batch, 1 solve after a warm call, "
                      f"{its} iterations in {dt:.2f} s (oracle/, OpenMP over trajectories; {threads} threads = the cgroup CPU quota "
                      f"({quota} cores) on a box showing {phys} physical cores)"}


def c1_cpu_line(T, configs):
    """BASELINE config C1: examples/quickstart.jl:28-59 as written (2-D double integrator, N=21, tf=3, Goal +
    Circle + Norm-SOC + Bound) and the N=51 variant BASELINE.json's label names, AL-iLQR on ONE trajectory on ONE host core —
    the reference's own CPU-runnable case, timed on the oracle (best of 5 after a warm call).  Initial controls: the file
    draws U0 = randn (:63); U0 = 0 is the one start that cannot work — the straight line from x0 to xf runs through the
    centre of the circular obstacle and the problem is mirror-symmetric about it, so the iterates stay on the axis (a saddle:
    c_max stalls at 0.24) — hence the deterministic symmetry-breaking U0 = (0.1, 0) on every knot."""
    from oracle_binding import load_oracle_native, set_threads
    o, flags = load_oracle_native()
    out = {"kind": "port", "build": flags, "cores": 1, "unit": "trajectory-iterations/s"}
    for N in (21, 51):
        p = configs.quickstart_problem(N=N, lib=o)
        set_threads(p, 1)
        u0 = np.array([0.1, 0.0])
        T.initial_controls(p, u0)
        s = T.ALSolver(p)
        s.solve()
        best = float("inf")
        for _ in range(5):
            T.initial_controls(p, u0)
            p._call("reset_duals")
            t0 = time.perf_counter()
            s.solve()
            best = min(best, time.perf_counter() - t0)
        out["N%d" % N] = {"value": s.total_iterations / best, "iterations": int(s.total_iterations), "ms": 1e3 * best,
                          "outer": int(s.stats["iterations_outer"][0]), "c_max": float(s.stats["c_max"][0]),
                          "status": int(s.stats["status"][0])}
    out["value"] = out["N21"]["value"]
    out["sample"] = "quickstart.jl values (N=21, tf=3) -> value; N=51 variant alongside; one trajectory, one core; U0 = (0.1, 0)"
    return out


def roofline_block(configs, name, batch, dims, iters, value_per_gpu, kms, kln, build_id, path):
    """roofline object from the hipEvent phase timings of the PROFILED pass (same workload, run right after the timed one).
    path = to_solver_path: when the expansion is fused into the backward-pass kernel there is no expansion launch — the two
    phases are one kernel, "expand_backward", which owns the algorithmic bytes of both (the [A B] blocks it no longer moves
    through memory still count as work done: they are the SURVEY §8d figure)."""
    n, m, ne, N, duals = dims
    bytes_it = configs.algorithmic_bytes_per_iteration(n, m, ne, N, duals)
    split = kernel_split_bytes(n, m, ne, N, duals)
    kern = {}
    for i, kn in enumerate(["expand", "backward", "forward"]):
        if kln[i] > 0:
            kern[kn] = {"ms_total": kms[i], "launches": int(kln[i]), "avg_us": 1e3 * kms[i] / kln[i]}
    if path["fused_expansion"] and "expand" in kern and "backward" in kern:
        e, b = kern.pop("expand"), kern.pop("backward")  # the expand slot only holds the gap between two event records
        kern = {"expand_backward": {"ms_total": e["ms_total"] + b["ms_total"], "launches": b["launches"],
                                    "avg_us": 1e3 * (e["ms_total"] + b["ms_total"]) / b["launches"]}, **kern}
        split = {"expand_backward": split["expand"] + split["backward"], "forward": split["forward"]}
    if not kern:
        return None
    dom = max(kern, key=lambda k: kern[k]["ms_total"])
    # a launch processes, on average, (trajectory-iterations of this rank) / launches units
    units_per_launch = iters / kern[dom]["launches"]
    achieved = split[dom] * units_per_launch / (kern[dom]["avg_us"] * 1e-6) / 1e9
    traffic, traffic_src, flops = pmc_traffic(name, batch, dom, build_id)
    return {"bound": "hbm", "kernel": "k_" + dom, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
            "algorithmic_bytes_per_unit": split[dom], "units_per_launch": units_per_launch,
            "avg_launch_us": kern[dom]["avg_us"], "kernels": kern,
            "timing": "hipEvents on the library stream around each phase, in a separate profiled pass of the same "
                      "steps (the timed pass records no events)",
            # what actually bounds these kernels: FP64 vector issue / dependency latency
            "fp64_valu": None if not flops else {
                "flops_per_launch": flops, "achieved_tflops": flops / (kern[dom]["avg_us"] * 1e-6) / 1e12,
                "peak_tflops": FP64_VALU_PEAK_TFLOPS,
                "frac": flops / (kern[dom]["avg_us"] * 1e-6) / 1e12 / FP64_VALU_PEAK_TFLOPS},
            "whole_iteration": {"algorithmic_bytes_per_unit": bytes_it,
                                "achieved": bytes_it * value_per_gpu / 1e9,
                                "frac": bytes_it * value_per_gpu / 1e9 / HBM_PEAK_GBS}}


def overlap_run(T, configs, lib, name, batch, parts, device):
    """The same batch as `parts` contiguous sub-batches, each on its own handle (= its own stream and worker thread), all solves in
    flight at once.  A solve is batch-synchronous and its batch drains unevenly (C3: 141 batch steps for a mean of 52 iterations);
    the drained tail of one sub-batch leaves most of the chip to the others.  Results are those of the single handle (a
    trajectory's result does not depend on its batch).  Reported next to the headline, never as the headline."""
    sizes = [batch // parts + (1 if i < batch % parts else 0) for i in range(parts)]
    offs = [sum(sizes[:i]) for i in range(parts)]
    probs = [build_problem(T, configs, name, sz, off, device, lib) for sz, off in zip(sizes, offs)]
    solvers = [make_solver(T, configs, name, p) for p in probs]
    u0 = initial_controls_value(T, probs[0], name)
    best = None
    for rep in range(3):
        for p in probs:
            T.initial_controls(p, u0)
        t0 = time.perf_counter()
        for s in solvers:
            s.solve_async()
        for s in solvers:
            s.wait()
        dt = time.perf_counter() - t0
        its = sum(int(s.total_iterations) for s in solvers)
        if best is None or dt < best[0]:
            best = (dt, its)
    return {"handles": parts, "sub_batches": sizes, "value": best[1] / best[0], "unit": "trajectory-iterations/s", "ms": 1e3 * best[0],
            "trajectory_iterations": best[1], "note": "best of 3; sub-batches solved concurrently through to_*_solve_async on separate streams"}


def pipelined_run(T, configs, lib, name, batch, b_offset, depth, steps, warmup, admit_frac, device, prob0, solver0, u0, gather0, make_gather_fn, barrier):
    """`steps` solves of the workload's batch, pipelined over `depth` handles (to_solve_progress / to_solve_wait_below; api.SolvePipeline):
    step i runs on handle i % depth — every handle holds the SAME batch, so every step is the unpipelined step, bit for bit — and is
    admitted when the step in front has at most admit_frac * batch trajectories still iterating.  The drained tail of one solve (C3: 52 of
    141 batch steps on a handful of stragglers, chip empty; C2: half of its 300 batch steps serve five trajectories) runs under the dense
    first steps of the next.  Timed between barriers from the first submit to the last wait: the ramp-up and the final, un-overlapped
    drain are inside the figure.  With several ranks every handle has its own communicator and gathers when its solve is collected (the
    collectives come in submit order on every rank).  -> (seconds, trajectory-iterations, batch steps, admit_below, gather_note)"""
    probs = [prob0] + [build_problem(T, configs, name, batch, b_offset, device, lib) for _ in range(depth - 1)]
    solvers = [solver0] + [make_solver(T, configs, name, p) for p in probs[1:]]
    gathers, note = [gather0], None
    for i in range(1, depth):
        g, gn = make_gather_fn(probs[i])
        gathers.append(g)
        note = note or gn
        T.initial_controls(probs[i], u0)
        solvers[i].solve()               # warm every handle (allocations of the first solve: polish workspace, event pools)
    slot_of = {id(sv): i for i, sv in enumerate(solvers)}

    def on_done(job, sv):
        g = gathers[slot_of[id(sv)]]
        if g is not None:
            with Watchdog("all-gather of the converged trajectories and stats (%s, pipelined job %d)" % (name, job)):
                g()
                g.stats(sv)

    prep = lambda p: T.initial_controls(p, u0)
    best = None
    for rep in range(1 + warmup):        # the warm-up repetitions run the same pipelined sequence
        pipe = T.SolvePipeline(solvers, admit_below=int(admit_frac * batch), on_done=on_done)
        barrier()
        t0 = time.perf_counter()
        for _ in range(steps):
            pipe.submit(prep)
        pipe.drain()
        barrier()
        dt = time.perf_counter() - t0
        best = (dt, pipe.total_iterations, sum(r[2] for r in pipe.results))
    dt, its, bsteps = best
    for g in gathers[1:]:
        if g is not None:
            g.close()
    del pipe, solvers, probs, gathers      # the extra handles go NOW: idle streams take hardware-queue slots from whatever runs next
    import gc
    gc.collect()
    return dt, its, bsteps, int(admit_frac * batch), note


class Watchdog:
    """A collective that never returns (a peer that died inside ncclCommInitRank, a rank that took another branch) must end as a
    LABELLED failure of this rank, not as the driver's wall-clock limit: `with Watchdog("what", seconds)` exits the process with code 3
    and a rank-tagged line on stderr when the body is still running after `seconds` (TRAJOPT_BENCH_TIMEOUT overrides; 0 disables)."""

    def __init__(self, what, seconds=300.0):
        import threading
        env = os.environ.get("TRAJOPT_BENCH_TIMEOUT")
        self.seconds = float(env) if env else seconds
        self.what, self._timer, self._threading = what, None, threading

    def _fire(self):
        sys.stderr.write("[bench.py rank %s] TIMEOUT after %.0f s in: %s — giving up (exit 3)\n" % (os.environ.get("RANK", "0"), self.seconds, self.what))
        sys.stderr.flush()
        os._exit(3)

    def __enter__(self):
        if self.seconds > 0:
            self._timer = self._threading.Timer(self.seconds, self._fire)
            self._timer.daemon = True
            self._timer.start()
        return self

    def __exit__(self, *exc):
        if self._timer is not None:
            self._timer.cancel()
        return False


def make_gather(lib, prob, dist, torch, rank, local_rank, world, name):
    """RCCL all-gather of the converged trajectories of `prob`'s handle (device-to-device, the library's own communicator) -> (gather,
    note); (None, None) without torch.distributed.  Collective: every rank calls it for its handles in the same order."""
    gather, gather_note = None, None
    if dist is not None:  # RCCL all-gather of the converged trajectories, device-to-device
        from trajectoryoptimization_jl_amd.distributed import TrajectoryGather
        # The library's own communicator (to_comm_*).  Should it fail to come up on ANY rank (it has only ever run with one rank on
        # the builder's boxes), every rank falls back to gathering through torch.distributed — the same RCCL, staged through host
        # arrays — rather than losing the scaling run; the line says which path ran.
        err = None
        dev = torch.device("cuda", local_rank)
        # Step 1, symmetric by construction: can EVERY rank load RCCL through the library (dlopen + ncclGetUniqueId)?  A rank that
        # cannot must not leave its peers waiting inside ncclCommInitRank, so the answer is reduced before anyone enters it.
        try:
            if os.environ.get("TRAJOPT_BENCH_GATHER") == "torch":
                raise RuntimeError("forced by TRAJOPT_BENCH_GATHER=torch")
            lib.call("comm_unique_id", (C.c_char * 128)())
        except Exception as e:  # noqa: BLE001
            err = f"{type(e).__name__}: {e}"
        ok = torch.tensor([0 if err else 1], device=dev)
        with Watchdog("all_reduce of the RCCL availability flag (%s)" % name):
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        if int(ok.item()) == 1:  # Step 2: the communicator itself (a failure here — e.g. a duplicate device — hits every rank alike)
            try:
                with Watchdog("to_comm_init_rank with %d ranks (%s)" % (world, name)):
                    gather = TrajectoryGather(prob, dist, device=dev)
            except Exception as e:  # noqa: BLE001 - any failure of the native communicator takes the fallback
                err = f"{type(e).__name__}: {e}"
                sys.stderr.write("[bench.py rank %d] library communicator failed: %s\n" % (rank, err))
        ok = torch.tensor([0 if err else 1], device=dev)
        with Watchdog("all_reduce of the communicator status (%s)" % name):
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        if int(ok.item()) == 0:
            if gather is not None:
                gather.close()
            gather = TrajectoryGather(prob, dist, device=None)
            gather_note = "torch.distributed all_gather of host-staged arrays (the library's RCCL communicator did not come up: %s)" % (err or "on another rank")

    return gather, gather_note


def run_workload(T, configs, lib, name, batch, steps, warmup, rank, local_rank, world, dist, torch, profile=True, pipeline=0, pipeline_steps=0,
                 admit_frac=0.5):
    """Timed (event-free) pass of `steps` solves, then a profiled pass of the same steps for the per-phase timings.  pipeline = d > 1: a
    third pass of `pipeline_steps` solves pipelined over d handles becomes the line's value (with several ranks: every handle with its own
    communicator, gathering when its solve is collected); the unpipelined figure stays beside it under "unpipelined"."""
    W = WORKLOADS[name]
    # The cpu_baseline leg pins its OpenMP threads (OMP_PROC_BIND=close): once that runtime has started, THIS thread is bound to one core, and
    # the solve threads the library starts (one per handle in flight) would inherit a one-core mask — four of them driving 100 us batch steps
    # of the pipelined Cartpole solves from one core (5.4 instead of 8.8 M it/s).  Back to the mask the process started with.
    if hasattr(os, "sched_setaffinity"):
        try:
            os.sched_setaffinity(0, START_AFFINITY)
        except OSError:
            pass
    prob = build_problem(T, configs, name, batch, rank * batch, local_rank, lib)
    solver = make_solver(T, configs, name, prob)
    u0 = initial_controls_value(T, prob, name)
    n, m, N = prob.dims()
    dims = (n, m, prob.errstate_dim, N, sum(prob.constraints.p))
    info = (C.c_int32 * 8)()
    prob._call("solver_path", info)
    path = {"backward": ("coop", "mfma", "lane")[info[0]], "fused_expansion": bool(info[1]), "compaction": bool(info[2]),
            "first_round_step_sizes": int(info[3]), "forward_waves_per_workgroup": int(info[4]), "scan_backward": bool(info[5]),
            "accept_by_rollout": bool(info[6]), "line_search_repack": bool(info[7] & 1), "repacked_working_set": bool(info[7] & 2)}
    gather, gather_note = make_gather(lib, prob, dist, torch, rank, local_rank, world, name)

    def one_step():
        T.initial_controls(prob, u0)          # device-side reset of the batch to the initial guess
        solver.solve()
        if gather is not None:
            with Watchdog("all-gather of the converged trajectories and stats (%s)" % name):
                gather()
                gather.stats(solver)
        return solver.total_iterations, solver.batch_steps

    def barrier():
        if dist is not None:
            with Watchdog("barrier (%s)" % name, 900.0):
                dist.barrier()
        torch.cuda.synchronize()

    for _ in range(warmup):
        one_step()
    prob._call("set_profiling", 0)
    barrier()
    t0 = time.perf_counter()
    iters = bsteps = 0
    for _ in range(steps):
        it, bs = one_step()
        iters += it
        bsteps += bs
    barrier()
    dt = time.perf_counter() - t0
    status = solver.stats["status"].copy()
    it_pn = solver.stats["iterations_pn"].copy()
    kms, kln = (C.c_double * 4)(), (C.c_int64 * 4)()
    iters_prof = 0
    if profile:  # the per-phase launch counts AND the iteration count of the roofline block both come from this pass
        prob._call("reset_profile")
        prob._call("set_profiling", 1)
        for _ in range(steps):
            iters_prof += one_step()[0]
        prob._call("set_profiling", 0)
        prob._call("get_profile", kms, kln)
    if dist is not None:
        t = torch.tensor([dt, float(iters)], dtype=torch.float64, device="cuda")
        tmax = t.clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        tsum = t.clone()
        dist.all_reduce(tsum, op=dist.ReduceOp.SUM)
        dt_max, iters_all = float(tmax[0]), float(tsum[1])
    else:
        dt_max, iters_all = dt, float(iters)
    value = iters_all / dt_max
    res = {"value": value, "unit": "trajectory-iterations/s", "steps": steps, "warmup": warmup,
           "ms_per_step": 1e3 * dt_max / steps,
           "config": {"workload": W["desc"].replace(f"batch={W['batch']} per GPU", f"batch={batch} per GPU"), "batch_per_gpu": batch, "knot_points": N, "n": n, "m": m,
                      "trajectory_iterations_per_step": iters_all / steps, "batch_steps_per_solve": bsteps / steps,
                      "converged_fraction": float(np.mean(status == T.capi.SOLVE_SUCCEEDED)), "solver_path": path,
                      "projected_newton": None if W["solver"] != "altro" else {
                          "polished_fraction": float(np.mean(it_pn > 0)), "linearisations_histogram": np.bincount(it_pn).tolist(),
                          "projection_failed_fraction": float(np.mean(status == T.capi.PROJECTION_FAIL)),
                          "ms_per_solve": (kms[3] / kln[3]) if (profile and kln[3] > 0) else None},
                      "collective": ("RCCL all_gather of converged (X,U) once per solve + stats gather (to_allgather / to_allgather_stats); "
                                     "ranks the RCCL communicator saw: %d, shards %s" % (len(gather.counts), gather.counts))
                      if gather is not None and gather_note is None else (gather_note or "none")},
           "roofline": roofline_block(configs, name, batch, dims, iters_prof, value / world, kms, kln, lib.build_id(), path) if profile else None}
    if pipeline > 1:
        psteps = pipeline_steps or max(steps, 2 * pipeline)
        pdt, pits, pbsteps, admit_below, pnote = pipelined_run(
            T, configs, lib, name, batch, rank * batch, pipeline, psteps, warmup, admit_frac, local_rank, prob, solver, u0, gather,
            lambda p: make_gather(lib, p, dist, torch, rank, local_rank, world, name), barrier)
        if dist is not None:
            t = torch.tensor([pdt, float(pits)], dtype=torch.float64, device="cuda")
            tmax, tsum = t.clone(), t.clone()
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            dist.all_reduce(tsum, op=dist.ReduceOp.SUM)
            pdt, pits_all = float(tmax[0]), float(tsum[1])
        else:
            pits_all = float(pits)
        pr = {"depth": pipeline, "admit_below": admit_below, "steps": psteps, "value": pits_all / pdt, "ms_per_step": 1e3 * pdt / psteps,
              "trajectory_iterations": pits_all, "batch_steps_per_solve": pbsteps / psteps,
              "note": "value = trajectory-iterations of `steps` solves (all ranks) / max-over-ranks wall time between barriers, from the first submit to the "
                      "last wait (ramp-up and final drain included); every solve is the unpipelined solve bit for bit (tests/test_gpu_pipeline.py)"}
        res["unpipelined"] = {"value": res["value"], "ms_per_step": res["ms_per_step"], "steps": res["steps"]}
        res["value"], res["ms_per_step"], res["steps"] = pr["value"], pr["ms_per_step"], pr["steps"]
        res["config"]["workload"] += (f"; PIPELINED over {pipeline} handles: {pr['steps']} solves of the same batch, the next admitted when the one in flight "
                                      f"has <= {pr['admit_below']} trajectories still iterating")
        res["config"]["pipeline"] = pr
        if pnote and res["config"].get("collective"):
            res["config"]["collective"] += " | pipelined handles: " + pnote
        if res["roofline"] is not None:  # kernel timings: the unpipelined profiled pass; the whole-iteration figure: the pipelined value
            wi = res["roofline"]["whole_iteration"]
            wi["achieved"] = wi["algorithmic_bytes_per_unit"] * res["value"] / world / 1e9
            wi["frac"] = wi["achieved"] / HBM_PEAK_GBS
            wi["note"] = "from the pipelined value; the per-kernel figures above are the unpipelined profiled pass"
    return res, prob, u0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=0, help="timed solves (default 24 / 16 / 12 for cartpole / quadrotor / the constrained workloads): the unpipelined pass "
                                                          "and the pipelined pass both time exactly this many")
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="cartpole", choices=list(WORKLOADS))
    ap.add_argument("--batch", type=int, default=0, help="override the per-GPU batch (parity/scaling studies only)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--throughput-probe", type=int, default=-1,
                    help="also solve a batch of this size once (reported separately; default 32768 for the cartpole workload at N=1, 0 = off)")
    ap.add_argument("--no-probe-sweep", dest="probe_sweep", action="store_false",
                    help="skip the batch sweep (2x ... 64x the probe batch) that locates the throughput plateau")
    ap.add_argument("--no-profile", action="store_true", help="skip the separate hipEvent-profiled pass (no roofline object)")
    ap.add_argument("--no-extra", action="store_true", help="do not append the C3 / C5 lines (extra_workloads) to the default C2 run")
    ap.add_argument("--pipeline", type=int, default=-1,
                    help="pipeline the solves over this many handles (the next solve is admitted when the one in flight has drained below "
                         "--admit of its batch); default 4 for every workload (0 / 1: unpipelined only); the unpipelined figure of the same run "
                         "is always reported beside it")
    ap.add_argument("--pipeline-steps", type=int, default=0, help="solves of the pipelined pass (default: --steps)")
    ap.add_argument("--admit", type=float, default=-1.0, help="admit the next solve at this fraction of the batch still iterating (default per workload)")
    ap.add_argument("--overlap", type=int, default=0,
                    help="also solve the workload as this many sub-batches on as many handles / streams in flight at once (to_*_solve_async): "
                         "what overlapping the drained tails recovers; reported separately under \"overlap\", never the headline")
    args = ap.parse_args()

    # stdout carries exactly ONE line, the JSON record: everything native code prints while we run (RCCL's version banner
    # goes to stdout on communicator creation) is routed to stderr at the file-descriptor level
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    os.environ.setdefault("OMP_PLACES", "cores")       # the cpu_baseline leg: pinned threads (read when the OpenMP runtime starts)
    os.environ.setdefault("OMP_PROC_BIND", "close")
    import torch
    import trajopt_amd as T
    from trajectoryoptimization_jl_amd import configs

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N > 1")
    dist = None
    torch.cuda.set_device(local_rank)
    if "RANK" in os.environ:  # launched by torch.distributed.run (also with one rank: same code path, RCCL on one GPU)
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        import datetime
        with Watchdog("torch.distributed.init_process_group with %d ranks" % world):
            dist.init_process_group(backend="nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank),
                                    timeout=datetime.timedelta(seconds=900))

    lib = T.load_hip_library()
    if lib.device_count() < 1:
        raise SystemExit("bench.py needs a GPU: libtrajopt_hip.so has no CPU fallback")
    name = args.workload
    if args.steps <= 0:
        args.steps = {"cartpole": 24, "quadrotor": 16}.get(name, 12)
    batch = args.batch or WORKLOADS[name]["batch"]
    depth = args.pipeline if args.pipeline >= 0 else (0 if args.batch else 4)   # (--batch: scaling studies of one handle)
    res, prob, u0 = run_workload(T, configs, lib, name, batch, args.steps, args.warmup, rank, local_rank, world, dist, torch,
                                 profile=not args.no_profile, pipeline=depth,
                                 pipeline_steps=args.pipeline_steps or args.steps,
                                 admit_frac=args.admit if args.admit >= 0 else ADMIT_FRAC.get(name, 1.0))

    if rank == 0:
        out = {"metric": "iLQR iterations/sec (batched trajectories)", "value": res["value"], "unit": res["unit"],
               "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": res["ms_per_step"],
               "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
               "config": res["config"], "roofline": res["roofline"], "unpipelined": res.get("unpipelined"), "build_id": lib.build_id()}
        probe = args.throughput_probe
        if probe < 0:
            probe = 32768 if (name == "cartpole" and world == 1) else 0
        if probe > 0:  # outside the timed region: the same kernels on batches large enough to fill the chip
            n, m, N = prob.dims()
            bytes_it = configs.algorithmic_bytes_per_iteration(n, m, prob.errstate_dim, N, sum(prob.constraints.p))

            def probe_once(pbatch, reps=3):
                """One warm solve, then `reps` timed solves (each from the initial guess, reset on the device): the point is their mean."""
                pb = build_problem(T, configs, name, pbatch, 0, local_rank, lib)
                ps = make_solver(T, configs, name, pb)
                ps.solve()
                ms = []
                for _ in range(reps):
                    T.initial_controls(pb, u0)
                    t1 = time.perf_counter(); ps.solve(); ms.append(1e3 * (time.perf_counter() - t1))
                d1 = 1e-3 * float(np.mean(ms))
                r = {"batch": pbatch, "value": ps.total_iterations / d1, "unit": "trajectory-iterations/s", "ms": 1e3 * d1,
                     "solves_averaged": reps, "ms_each": [round(x, 2) for x in ms], "batch_steps": int(ps.batch_steps),
                     "whole_iteration_frac": bytes_it * ps.total_iterations / d1 / 1e9 / HBM_PEAK_GBS}
                del ps, pb
                return r

            out["throughput_probe"] = probe_once(probe)
            if args.probe_sweep and name == "cartpole":  # where does the throughput plateau?
                sweep = [out["throughput_probe"]]
                for pbatch in (2 * probe, 4 * probe, 8 * probe, 16 * probe, 32 * probe, 64 * probe):
                    try:
                        sweep.append(probe_once(pbatch))
                    except Exception as e:
                        sweep.append({"batch": pbatch, "error": repr(e)})
                        break
                best = max((r for r in sweep if "value" in r), key=lambda r: r["value"])
                out["throughput_sweep"] = {"points": sweep, "plateau": {k: best[k] for k in ("batch", "value", "whole_iteration_frac")}}
        if args.overlap > 1:
            out["overlap"] = overlap_run(T, configs, lib, name, batch, args.overlap, local_rank)
        if not args.no_cpu_baseline:
            try:
                out["cpu_baseline"] = cpu_baseline(T, configs, name, batch)
            except Exception as e:  # the oracle is a checker; its absence must not hide the GPU number
                out["cpu_baseline"] = {"error": repr(e)}
            if name == "cartpole" and world == 1:
                try:
                    out["c1_cpu"] = c1_cpu_line(T, configs)
                except Exception as e:
                    out["c1_cpu"] = {"error": repr(e)}
    del prob
    import gc
    gc.collect()
    # The other single-GPU BASELINE configurations, driver-visible in the same JSON line (2 steps each; C4 is C3 sharded)
    if world == 1 and name == "cartpole" and not args.batch and not args.no_extra:
        extra = {}
        for key, wname, pdepth, psteps in (("C3", "quadrotor", 4, 16), ("C5", "quadrotor_altro", 4, 12), ("C5_altro_defaults", "quadrotor_altro_defaults", 0, 0)):
            try:
                if args.pipeline >= 0:
                    pdepth = args.pipeline if pdepth else 0
                r, p2, _ = run_workload(T, configs, lib, wname, WORKLOADS[wname]["batch"], 2, 1, 0, local_rank, 1, None, torch,
                                        profile=(not args.no_profile) and key != "C5_altro_defaults",
                                        pipeline=pdepth, pipeline_steps=psteps,
                                        admit_frac=args.admit if args.admit >= 0 else ADMIT_FRAC.get(wname, 1.0))
                del p2
                import gc
                gc.collect()
                if not args.no_cpu_baseline and key != "C5_altro_defaults":
                    r["cpu_baseline"] = cpu_baseline(T, configs, wname, WORKLOADS[wname]["batch"])
                extra[key] = r
            except Exception as e:
                extra[key] = {"error": repr(e)}
        out["extra_workloads"] = extra
    # Launched by torch.distributed.run (the N > 1 case): BASELINE config C4 alongside the headline — the Quadrotor point-to-point
    # workload sharded 4096 trajectories per GPU (with 8 ranks: exactly C4's 32 768), same weak-scaling protocol, same RCCL gathers.
    if dist is not None and name == "cartpole" and not args.batch and not args.no_extra:
        r4, p4, _ = run_workload(T, configs, lib, "quadrotor", WORKLOADS["quadrotor"]["batch"], 2, 1, rank, local_rank, world, dist, torch,
                                 profile=False, pipeline=(args.pipeline if args.pipeline >= 0 else 4), pipeline_steps=16, admit_frac=1.0)
        del p4
        if rank == 0:
            r4["config"]["workload"] = "C4: " + r4["config"]["workload"] + f" x {world} ranks = {world * WORKLOADS['quadrotor']['batch']} trajectories, RCCL all-gather per solve"
            out.setdefault("extra_workloads", {})["C4"] = r4
    if rank == 0:
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
