"""cpu_baseline leg of bench.py as a process of its own (TEST INFRASTRUCTURE / measurement only, like everything under
oracle/; parity unpinned, see gat_oracle.py).

    python3 oracle/cpu_bench.py --scale 20 --draws 5000000 --fin 128 --heads 8 --fout 16 --steps 5 [--dx]

bench.py starts it BEFORE its own process touches the GPU or spawns torch's OpenMP pool and reads ONE JSON line from
stdout.  Why a process of its own (2.37 s/step in one round, 4.63 in the next, same source, same CPU
model): in-process the port ran on `omp_get_max_threads()` = every hardware thread of the host, unpinned, beside torch's
own libgomp pool (a second OpenMP runtime in the same process, its idle threads spinning) and whatever bench.py's other
legs had left running.  Here: no torch in the process, one OpenMP runtime, one thread per PHYSICAL core this process may
use (affinity mask and cgroup quota respected), OMP_PLACES=cores / OMP_PROC_BIND=close, median of >= 5 steps.

The graph is the headline's recipe (R-MAT, same scale / draws / quadrant probabilities, symmetrised, self loops) from a
numpy stream -- the SAME stream bench.py's GPU leg draws from on the host and uploads (pygat_amd/rmat.py rmat_csr_numpy; the
two copies are pinned to each other by tests/test_dist_cpu.py), so both legs time the identical graph (round 5).
value = E / median step time.
"""
import argparse
import json
import math
import os
import sys
import time


def physical_cores():
    """(cores, hw_threads, quota): distinct physical cores in this process's affinity mask, hardware threads in it, and the
    cgroup CPU quota in cores (None: unlimited)."""
    try:
        cpus = sorted(os.sched_getaffinity(0))
    except AttributeError:
        cpus = list(range(os.cpu_count() or 1))
    cores = set()
    for c in cpus:
        try:
            sib = open(f"/sys/devices/system/cpu/cpu{c}/topology/thread_siblings_list").read().strip()
        except OSError:
            sib = str(c)
        cores.add(sib)
    quota = None
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            quota = float(q) / float(p)
    except (OSError, ValueError):
        pass
    return len(cores), len(cpus), quota


def first_thread_of_each_core():
    try:
        cpus = sorted(os.sched_getaffinity(0))
    except AttributeError:
        return None
    seen, first = set(), []
    for c in cpus:
        try:
            sib = open(f"/sys/devices/system/cpu/cpu{c}/topology/thread_siblings_list").read().strip()
        except OSError:
            sib = str(c)
        if sib not in seen:
            seen.add(sib)
            first.append(c)
    return first


def rmat_csr_numpy(scale, n_draws, abcd=(0.57, 0.19, 0.19, 0.05), seed=1):
    import numpy as np
    n = 1 << scale
    rng = np.random.default_rng(seed)
    a, b, c, _ = abcd
    r = np.zeros(n_draws, np.int64)
    ci = np.zeros(n_draws, np.int64)
    for _ in range(scale):
        u = rng.random(n_draws, dtype=np.float32)
        rb = (u >= a + b).astype(np.int64)
        cb = (((u >= a) & (u < a + b)) | (u >= a + b + c)).astype(np.int64)
        r = (r << 1) | rb
        ci = (ci << 1) | cb
    ar = np.arange(n, dtype=np.int64)
    key = np.unique(np.concatenate([r, ci, ar]) * n + np.concatenate([ci, r, ar]))
    rr, cc = key // n, key % n
    rowptr = np.zeros(n + 1, np.int64)
    rowptr[1:] = np.cumsum(np.bincount(rr, minlength=n))
    return rowptr.astype(np.int32), cc.astype(np.int32)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scale", type=int, default=20)
    ap.add_argument("--draws", type=int, default=5_000_000)
    ap.add_argument("--fin", type=int, default=128)
    ap.add_argument("--heads", type=int, default=8)
    ap.add_argument("--fout", type=int, default=16)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--dx", action="store_true")
    args = ap.parse_args()

    cores, hw, quota = physical_cores()
    threads = cores if quota is None else max(1, min(cores, int(math.floor(quota + 1e-9))))
    if os.environ.get("PYGAT_CPU_BENCH_CHILD") != "1":
        # the OpenMP runtime reads its environment when it is loaded: re-run this script with it set (nothing here has
        # touched a GPU; this is the one exec of the process)
        env = dict(os.environ, PYGAT_CPU_BENCH_CHILD="1", OMP_NUM_THREADS=str(threads), OMP_PLACES="cores",
                   OMP_PROC_BIND="close", OMP_DYNAMIC="false", OMP_WAIT_POLICY="active")
        os.execve(sys.executable, [sys.executable] + sys.argv, env)

    import numpy as np
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from oracle import c_oracle
    lib = c_oracle.load()
    t0 = time.perf_counter()
    rowptr, col = rmat_csr_numpy(args.scale, args.draws)
    N, E = len(rowptr) - 1, len(col)
    rng = np.random.default_rng(2)
    H, Fo, Fin = args.heads, args.fout, args.fin
    X = rng.standard_normal((N, Fin), dtype=np.float32)
    W = rng.standard_normal((H, Fin, Fo), dtype=np.float32) * np.float32(1.414 * (2.0 / (Fin + Fo)) ** 0.5)
    a = rng.standard_normal((H, 2 * Fo), dtype=np.float32) * np.float32(1.414 * (2.0 / (1 + 2 * Fo)) ** 0.5)
    G = rng.standard_normal((N, H * Fo), dtype=np.float32)
    tp = c_oracle.transpose_pattern(rowptr, col)
    t_setup = time.perf_counter() - t0
    c_oracle.level(X, rowptr, col, W, a, 0.2, True, G, want_dx=args.dx, lib=lib, tp=tp)   # warm-up: page faults, thread start
    times = []
    for _ in range(max(1, args.steps)):
        t0 = time.perf_counter()
        c_oracle.level(X, rowptr, col, W, a, 0.2, True, G, want_dx=args.dx, lib=lib, tp=tp)
        times.append(time.perf_counter() - t0)
    med = float(np.median(times))
    try:
        cpu = next(ln.split(":", 1)[1].strip() for ln in open("/proc/cpuinfo") if ln.startswith("model name"))
    except (OSError, StopIteration):
        cpu = "unknown"
    print(json.dumps({
        "value": E / med, "unit": "edges/s", "cores": cores if quota is None else min(cores, threads), "threads": int(lib.gat_oracle_threads()),
        "hw_threads_visible": hw, "cgroup_cpu_quota": quota, "kind": "port", "cpu": cpu,
        "step_s": [round(t, 4) for t in times], "median_step_s": round(med, 4), "min_step_s": round(min(times), 4),
        "setup_s": round(t_setup, 2),
        "method": f"median of {len(times)} steps after one warm-up, one thread per physical core of the cgroup quota, child process",
        "graph_source": "numpy stream, seed 1 (pygat_amd.rmat.rmat_csr_numpy): the graph the GPU leg uploads", "nodes": N, "edges": E,
        "sample": f"the headline workload itself (N={N}, E={E}: the same numpy-drawn graph as the GPU leg), "
                  f"median of {len(times)} fwd+bwd steps after one warm-up, {med:.2f} s/step, OpenMP C port "
                  f"oracle/gat_oracle.c in its own process: {int(lib.gat_oracle_threads())} threads = one per physical core "
                  f"(OMP_PLACES=cores, OMP_PROC_BIND=close), no second OpenMP runtime in the process"}))


if __name__ == "__main__":
    main()
