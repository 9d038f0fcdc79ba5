#!/usr/bin/env python3
"""bench.py -- particle-steps/second of the bootstrap particle filter on MI355X.

Contract: ``python bench.py --gpus N --steps K --warmup W``.  A "step" is one stepFilter
(model/ParticleFilter.scala:116-132) over the whole cloud: propagate + weight + log-sum-exp + systematic
resampling for one observation.

N > 1: one process per GPU over RCCL.  Launched under ``torch.distributed.run`` (WORLD_SIZE set) this process is
one rank; launched bare (``python bench.py --gpus 8``) it starts the N ranks itself as child processes -- before
anything in this process has touched a GPU -- waits a bounded time, forwards rank 0's JSON line and exits
non-zero if any rank failed.

Workload: BASELINE.json configs[1] -- seasonal-Poisson with OU latents (d = 3), N = 2^20 particles per GPU (weak
scaling: the global filter has N_gpus * 2^20 particles), T = K observations, synthetic data (tests/cases.py
simulator, seed 20260101).  N = 1: the W warm-up steps start the filter and the timed legs CONTINUE it -- value =
N * K / wall seconds of K more steps (cssm_pf_ll_filter_more; median of ``repeats`` legs), the cloud resident in HBM, the
K observations' records built and uploaded inside the timed call.  N > 1: a K-step series of the sharded filter.

Extra objects on the JSON line (N = 1 only): ``roofline`` for the dominant kernel from HIP events on its launch
stream, ``roofline_16m`` the same kernel at N = 2^24 (the north-star size) for the bench model (d = 3) and for
Poisson-Brownian (d = 1), and ``cpu_baseline``: the CPU restatement of the reference path (oracle/, one thread as
the reference runs) on a bounded sample of the same workload.
"""
import argparse
import json
import os
import signal
import socket
import subprocess
import sys
import time

import numpy as np

# multi-process GPU work on this pool needs dmabuf IPC (the image exports this already; kept for launchers that do not pass it on)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy ceiling)
N_PER_GPU = 1 << 20
N_16M = 1 << 24
METRIC = "particle-steps/sec (N x T) bootstrap filter"


def build_workload(T, which="c2"):
    import cases
    model = cases.c2_model() if which == "c2" else cases.c1_model()
    t, y, has = cases.poisson_counts(T)
    return model, t, y, has


# ------------------------------------------------------------------------------------------------ CPU baseline
def _cpu_sample(job):
    """One oracle filter over the sample (worker of the all-cores leg; also the single-thread leg)."""
    n, Ts, seed = job
    from oracle import oracle
    model, t, y, has = build_workload(Ts)
    o = oracle.OraclePf(model.descriptor(), n, seed)
    t1 = time.perf_counter()
    ll = o.filter(t[:Ts], y[:Ts], has[:Ts])[0]
    return time.perf_counter() - t1, ll


def cpu_baseline(budget_s=12.0, all_cores=True):
    """Oracle (CPU restatement of the reference path) on one core -- the reference runs one thread per filter
    (sequential Vector.map / foldLeft, model/ParticleFilter.scala:118,123,139) -- on a bounded sample; plus, as a stronger
    baseline, as many independent filters as this process may use cores, side by side."""
    n = 65536
    dt0, _ = _cpu_sample((n, 8, 20260101))
    per = dt0 / (n * 8)
    Ts = int(max(8, min(400, budget_s / (per * n))))
    dt, ll = _cpu_sample((n, Ts, 20260101))
    out = {"value": n * Ts / dt, "unit": "particle-steps/s", "cores": 1, "kind": "port",
           "sample": f"oracle/cssm_oracle.c, same model/data/seed as the GPU run, N={n} particles x T={Ts} observations, "
                     f"{dt:.1f} s on 1 of {os.cpu_count()} host cores", "ll": ll}
    if all_cores:
        try:
            import multiprocessing as mp
            cores = max(1, min(16, len(os.sched_getaffinity(0))))   # (a GPU box gives one GPU's share of the host: 16 cores)
            Tm = max(8, Ts // 3)
            with mp.get_context("spawn").Pool(cores) as pool:
                t0 = time.perf_counter()
                pool.map(_cpu_sample, [(n, Tm, 20260101 + 7 * i) for i in range(cores)])
                wall = time.perf_counter() - t0
            out["all_cores"] = {"value": cores * n * Tm / wall, "unit": "particle-steps/s", "cores": cores, "kind": "port",
                                "sample": f"{cores} independent filters (one process each, as PMMH chains would run), "
                                          f"N={n} x T={Tm} each, {wall:.1f} s wall including process start"}
        except Exception as e:  # a baseline, never a reason to lose the GPU figure
            out["all_cores"] = {"error": repr(e)}
    return out


# ------------------------------------------------------------------------------------------------ single GPU
def _kernel_profile(pf, t, y, has, K, loop_ms_plain):
    """Per-kernel durations of one K-step series: HIP events on the launch stream around every kernel (a pass of its
    own, so that the event records do not perturb the throughput figure), minus what a bracketing event pair adds.
    The event packets sit on the queue and inflate every bracketed launch; calibrated in place: the bracketed durations
    of all kernels of the series add up to more than the same loop takes without brackets, and the excess, split evenly
    over the brackets, is what one bracket adds (idle gaps of the plain loop stay inside the kernels' figures: the
    estimate errs on the long side)."""
    pf.profile(True)
    pf.run(t[:K], y[:K], has[:K])
    prof = pf.profile_read()
    pf.profile(False)
    pairs = sum(v[1] for v in prof.values())
    bracket_sum_ms = sum(v[0] for v in prof.values())
    pair_s = max(bracket_sum_ms - loop_ms_plain, 0.0) * 1e-3 / max(pairs, 1)
    per = {k: (max(v[0] / v[1] * 1e-3 - pair_s, 1e-9), v[1], v[0] / v[1] * 1e-3) for k, v in prof.items() if v[1]}
    return per, pair_s, prof


def _roofline(name, d, n, avg_s, cnt, raw_s, pair_s, copy_gbs, note=None):
    alg_bytes = (16 * d + 8) * n  # read 8d + write 8d + write logw per particle-step
    achieved = alg_bytes / avg_s / 1e9
    r = {"bound": "hbm", "kernel": name, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
         "frac": achieved / HBM_PEAK_GBS, "algorithmic_bytes_per_launch": alg_bytes,
         "algorithmic_bytes_per_particle": 16 * d + 8, "avg_launch_us": avg_s * 1e6, "launches": cnt,
         "raw_event_us": raw_s * 1e6, "event_pair_us": pair_s * 1e6}
    if copy_gbs:
        r["copy_ceiling"] = copy_gbs
        r["frac_of_copy_ceiling"] = achieved / copy_gbs
    if note:
        r["note"] = note
    return r


def run_single(args, emit=print):
    import torch  # first: its import takes long enough for the GPU clocks to idle down
    from composablestatespacemodels_amd.filter import NativePf
    K, W = args.steps, args.warmup
    n = args.particles
    R = args.repeats if args.repeats > 0 else (3 if K >= 200 else 7)
    W = max(W, 1)                      # (the filter has to be running before K more steps can be timed)
    model, t, y, has = build_workload(W + R * K + 8, args.model)
    pf = NativePf(model, n, 20260101, device=0)
    if args.fused is not None:
        pf.set_option(3, args.fused)   # CSSM_OPT_FUSED_SUMS (experiment switch; the default is the library's)
    d = pf.d
    torch.cuda.synchronize()
    # W untimed warm-up steps start the filter (initial cloud, allocations, clocks) ...
    pf.run(t[:W], y[:W], has[:W])
    # ... and the timed leg CONTINUES it: R legs of exactly K more steps each (cssm_pf_ll_filter_more: no new cloud, the series
    # goes on), every leg bracketed by a device synchronisation on both sides; the figure is the MEDIAN leg (a short leg on a
    # fresh box is otherwise at the mercy of one hiccup).  A leg is the whole host call: records built and uploaded, 2 K
    # launches enqueued, ll / ess read back.
    walls, loops = [], []
    ll = ess_t = None
    for r in range(R):
        lo = W + r * K
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        ll, _, ess_t = pf.run_more(t[lo:lo + K], y[lo:lo + K], has[lo:lo + K])
        torch.cuda.synchronize()
        walls.append(time.perf_counter() - t0)
        loops.append(pf.last_loop_ms())
    wall = float(np.median(walls))
    loop_ms = float(np.median(loops))
    per, pair_s, prof_raw = _kernel_profile(pf, t, y, has, K, loop_ms)
    lib = pf.lib
    pf.close()
    # on-box streaming ceiling (plain 16-byte-per-lane copy, 1 GiB each way: far beyond the 256 MB Infinity Cache)
    import ctypes as C
    copy = C.c_double(0.0)
    copy_gbs = None
    if lib.cssm_diag_copy_ceiling(0, 1 << 30, 10, C.byref(copy)) == 0:
        copy_gbs = copy.value
    traffic, traffic_source = None, None
    tpath = os.path.join(ROOT, "profiles", "traffic_latest.json")
    if os.path.exists(tpath):
        try:
            tj = json.load(open(tpath))
            traffic = tj.get("k_propagate_bytes_per_launch", tj.get("k_propagate_self_bytes_per_launch"))
            traffic_source = ("NOT measured in this run: copied from profiles/traffic_latest.json (rocprofv3 --pmc passes of "
                              + str(tj.get("date", "an earlier session")) + ", " + str(tj.get("source", "profiles/")) + ")")
        except Exception:
            traffic = None
    avg_s, cnt, raw_s = per["k_propagate"]
    roof = _roofline(f"k_propagate<{d},...> (gather + propagate + weight + log-sum-exp sums), N={n}", d, n, avg_s, cnt, raw_s, pair_s, copy_gbs)
    roof["timing"] = ("HIP events on the launch stream around every k_propagate launch of a K-step series, minus what a bracketing "
                      "event pair adds: (sum of all bracketed kernel times - device time of the same loop without brackets) / brackets")
    kernels_us = {k: v[0] * 1e6 for k, v in per.items()}
    roof["traffic"] = traffic
    roof["traffic_source"] = traffic_source
    out = {
        "metric": METRIC, "value": n * K / wall, "unit": "particle-steps/s",
        "n_gpus": 1, "steps": K, "warmup": W, "ms_per_step": wall * 1e3 / K, "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": ("configs[1]: seasonal-Poisson, OU latent (poisson(ou(1)) |+| seasonal(24,1,ou(2)), d=3), " if args.model == "c2"
                                else "configs[0] model at bench size: poisson(brownianMotion(1)), d=1, ") +
                               f"N={n} particles, T={K} observations, systematic resampling every observation",
                   "particles_per_gpu": n, "observations": K, "latent_dim": d, "seed": 20260101},
        "repeats": R, "value_is": "median over `repeats` timed legs of K steps each, continuing the filter the warm-up steps started (cssm_pf_ll_filter_more)", "wall_ms_each": [w * 1e3 for w in walls],
        "roofline": roof,
        "kernels_us": kernels_us,
        "device_loop_ms": loop_ms, "ll": ll, "ess_last": int(ess_t[-1]),
    }
    if not args.no_16m:
        out["roofline_16m"] = roofline_16m(NativePf, copy_gbs, args.fused)
    if not args.no_cpu:
        out["cpu_baseline"] = cpu_baseline()
    emit(json.dumps(out))


def roofline_16m(NativePf, copy_gbs, fused=None, K=24):
    """The fused kernel at the north-star size N = 2^24, driver-timed: the bench model (d = 3, 56 B per particle) and
    Poisson-Brownian (d = 1, 24 B per particle: SURVEY.md 8d's reading of the target).  Two variants each:
    the DEFAULT kernel -- gather + propagate + weight + the log-sum-exp sums (SURVEY 8d's K_fused; CSSM_OPT_FUSED_SUMS = 1) --
    and, under ``*_lean``, the same kernel without the sums (CSSM_OPT_FUSED_SUMS = 0: they are then a pass of their own,
    k_tile_sums), which is the kernel round 1 quoted.  Same algorithmic bytes (the sums add none), more arithmetic."""
    res = {}
    for key, which in (("c2_d3", "c2"), ("c1_d1", "c1")):
        model, t, y, has = build_workload(K, which)
        for variant, fz in (("", 1 if fused is None else fused), ("_lean", 0)):
            pf = NativePf(model, N_16M, 20260101, device=0)
            pf.set_option(3, fz)
            pf.run(t[:8], y[:8], has[:8])
            loop_ms = 1e30
            for _ in range(3):            # (the first series on a 1 GB handle is not representative: best of three)
                pf.run(t[:K], y[:K], has[:K])
                loop_ms = min(loop_ms, pf.last_loop_ms())
            per, pair_s, _ = _kernel_profile(pf, t, y, has, K, loop_ms)
            avg_s, cnt, raw_s = per["k_propagate"]
            r = _roofline(f"k_propagate<{pf.d},...> {'with' if fz else 'without'} the fused sums, N={N_16M}", pf.d, N_16M, avg_s, cnt, raw_s, pair_s, copy_gbs)
            r["fused_sums"] = bool(fz)
            r["step_us"] = loop_ms * 1e3 / K
            r["particle_steps_per_s"] = N_16M * K / (loop_ms * 1e-3)
            r["kernels_us"] = {k: v[0] * 1e6 for k, v in per.items()}
            res[key + variant] = r
            pf.close()
    return res


# ------------------------------------------------------------------------------------------------ N ranks
def run_multi(args, emit=print):
    import torch
    import torch.distributed as dist
    from composablestatespacemodels_amd.sharded import DistComm, ShardedFilter
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", str(args.gpus)))
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29511")
    gpu = args.backend == "nccl"
    if gpu:
        torch.cuda.set_device(local)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
    else:   # rehearsal of the launcher and of the orchestration on CPU (tests): gloo + the test-only oracle shard
        dist.init_process_group("gloo", rank=rank, world_size=world)
    K, W = args.steps, args.warmup
    n_global = args.particles * world
    model, t, y, has = build_workload(max(K, W, 8))
    if gpu:
        from composablestatespacemodels_amd.sharded import GpuShard
        shard = GpuShard(model, n_global, rank, world, 20260101, local)
        f = ShardedFilter([shard], DistComm(device=torch.device("cuda", local)))
    else:
        from oracle_shard import OracleShard
        shard = OracleShard(model, n_global, rank, world, 20260101)
        f = ShardedFilter([shard], DistComm())
    sync = torch.cuda.synchronize if gpu else (lambda: None)
    # warm-up: allocations, RCCL channels, clocks.  At least 8 observations whatever W is: the library's own RCCL
    # communicator and the exchange buffers are created at the first observation of the first series -- that must not
    # happen inside the timed region
    Wn = max(W, 8) if gpu else max(W, 1)
    tw, yw, hw = build_workload(Wn)[1:]
    f.ll_filter(tw[:Wn], yw[:Wn], hw[:Wn])
    dist.barrier()
    sync()
    t0 = time.perf_counter()
    ll, ess = f.ll_filter(t[:K], y[:K], has[:K])    # the K timed observations: one series, read back once at its end
    sync()
    dist.barrier()
    wall = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device="cuda" if gpu else "cpu")
    dist.all_reduce(wall, op=dist.ReduceOp.MAX)
    if rank == 0:
        w = float(wall.item())
        native = bool(getattr(f, "last_native", False))
        emit(json.dumps({
            "metric": METRIC, "value": n_global * K / w, "unit": "particle-steps/s",
            "n_gpus": world, "steps": K, "warmup": W, "ms_per_step": w * 1e3 / K, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "configs[1] sharded: seasonal-Poisson, OU latent (d=3), "
                                   f"{args.particles} particles per GPU x {world} GPUs = {n_global}, T={K}, global systematic "
                                   "resampling every observation (one all-to-all per observation carrying every rank's sum words and, "
                                   f"between adjacent ranks, its boundary particles; capacity {f.last_cap} rows per pair, "
                                   f"{getattr(f, 'last_resumes', 0)} resumed capacity misses)",
                       "particles_per_gpu": args.particles, "observations": K, "latent_dim": shard.d, "seed": 20260101},
            "exchange": {"capacity_rows": f.last_cap, "attempts": f.last_attempts, "backend": args.backend,
                         "shard_backend": "libcssm_pf (HIP)" if gpu else "oracle (CPU rehearsal, not a measurement)",
                         "collectives_issued_by": "libcssm_pf (cssm_pf_shard_series_rccl)" if native else "torch.distributed",
                         "collectives_per_observation": 1 if getattr(f, "last_single", False) else 2,
                         "all_to_all": getattr(f, "last_all_to_all", "equal split"),
                         "rccl": shard.lib.cssm_rccl_library().decode() if gpu else None},
            "ll": ll, "ess_last": ess}))
    if gpu:
        shard.close()
        f.comm.close()
    dist.destroy_process_group()


def launch_ranks(args, argv):
    """``python bench.py --gpus N`` without a launcher: start the N ranks as children (torch.distributed.run, one process
    per GPU) BEFORE this process has made any GPU call -- it never makes one -- wait a bounded time, hand rank 0's JSON
    line on.  Non-zero exit if a rank fails, no line appears, or the wait expires (the children's process group is killed)."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + argv
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, text=True, start_new_session=True)
    try:
        out, _ = proc.communicate(timeout=args.launch_timeout)
    except subprocess.TimeoutExpired:
        try:
            os.killpg(proc.pid, signal.SIGKILL)
        except ProcessLookupError:
            pass
        proc.wait()
        print(f"bench.py: the {args.gpus} ranks did not finish within {args.launch_timeout} s; killed", file=sys.stderr)
        return 3
    line = None
    for ln in out.splitlines():
        ln = ln.strip()
        if ln.startswith("{") and '"metric"' in ln:
            line = ln
    if proc.returncode != 0:
        sys.stderr.write(out)
        print(f"bench.py: torch.distributed.run exited with {proc.returncode}", file=sys.stderr)
        return proc.returncode or 1
    if line is None:
        sys.stderr.write(out)
        print("bench.py: the ranks finished without a result line", file=sys.stderr)
        return 4
    print(line, flush=True)
    return 0


class _QuietStdout:
    """Everything written to file descriptor 1 while the benchmark runs (RCCL prints a version banner there when a
    communicator is created) goes to stderr; the descriptor is restored for the one JSON line."""

    def __enter__(self):
        sys.stdout.flush()
        self._saved = os.dup(1)
        os.dup2(2, 1)
        return self

    def emit(self, line: str):
        sys.stdout.flush()
        os.dup2(self._saved, 1)
        print(line, flush=True)
        os.dup2(2, 1)

    def __exit__(self, *exc):
        sys.stdout.flush()
        os.dup2(self._saved, 1)
        os.close(self._saved)
        return False


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=500)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--particles", type=int, default=N_PER_GPU, help="particles per GPU")
    ap.add_argument("--repeats", type=int, default=0, help="timed K-step series (0: 3 for K >= 200, else 7); the median is reported")
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--no-16m", action="store_true", help="skip the roofline_16m leg")
    ap.add_argument("--fused", type=int, default=None, help="CSSM_OPT_FUSED_SUMS override (single GPU)")
    ap.add_argument("--model", default="c2", choices=["c2", "c1"], help="c2: the bench workload (BASELINE configs[1], d = 3); c1: Poisson-Brownian (configs[0], d = 1) -- profiling runs only")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"], help="gloo: CPU rehearsal of the N-rank path with the test-only oracle shard")
    ap.add_argument("--launch-timeout", type=float, default=1500.0, help="seconds the self-launched ranks may take")
    args = ap.parse_args()
    in_launcher = "WORLD_SIZE" in os.environ and "RANK" in os.environ
    if args.gpus > 1 and not in_launcher:
        sys.exit(launch_ranks(args, sys.argv[1:]))
    with _QuietStdout() as out:
        if int(os.environ.get("WORLD_SIZE", "1")) > 1:
            run_multi(args, out.emit)
        elif args.backend == "gloo":
            run_multi(args, out.emit)
        else:
            run_single(args, out.emit)


if __name__ == "__main__":
    main()
