#!/usr/bin/env python3
"""bench.py -- particle-steps/second of the bootstrap particle filter on MI355X.

Contract: ``python bench.py --gpus N --steps K --warmup W``.  A "step" is one stepFilter
(model/ParticleFilter.scala:116-132) over the whole cloud: propagate + weight + log-sum-exp + systematic
resampling for one observation.

N > 1: one process per GPU over RCCL.  Launched under ``torch.distributed.run`` (WORLD_SIZE set) this process is
one rank; launched bare (``python bench.py --gpus 8``) it starts the N ranks itself as child processes -- before
anything in this process has touched a GPU -- waits a bounded time, forwards rank 0's JSON line and exits
non-zero if any rank failed.

Workload (default, ``--model c2``): BASELINE.json configs[1] -- seasonal-Poisson with OU latents (d = 3), N = 2^20 particles
per GPU (weak scaling: the global filter has N_gpus * 2^20 particles), T = K observations, synthetic data (tests/cases.py
simulator, seed 20260101).  ``--model c4``: configs[3] -- a log-Gaussian Cox process (FilterLgcp, precision 2, ~10 sub-steps per
event), N = 2^24 particles IN TOTAL (strong scaling: 2^24 / N_gpus per GPU; ``--particles`` overrides the total).
At every N the W warm-up steps start the filter and the timed legs CONTINUE it -- value = particles * K / wall seconds of K
more steps (cssm_pf_ll_filter_more / its sharded form; median of ``repeats`` legs, max over ranks per leg), the cloud resident
in HBM, the K observations' records built and uploaded inside the timed call.  N > 1: plus a profiled pass whose per-rank
kernel / collective times, plan and resumes go into the line (``per_rank``).

Extra objects on the JSON line (N = 1 only): ``roofline`` for the dominant kernel from HIP events on its launch
stream, ``roofline_16m`` the same kernel at N = 2^24 (the north-star size) for the bench model (d = 3) and for
Poisson-Brownian (d = 1), and ``cpu_baseline``: the CPU restatement of the reference path (oracle/, one thread as
the reference runs) on a bounded sample of the same workload.
"""
import argparse
import gc
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


LGCP_PRECISION = 2


def build_workload(T, which="c2"):
    """(model, t, y, has) of T observations.  c4: T events on [0, T / 10] (what thinning a log-Gaussian Cox process yields:
    irregular increments, ~10 sub-steps of 10^-2 each)."""
    import cases
    if which == "c4":
        t, y, has = cases.event_times(T, horizon=0.1 * T)
        return cases.c4_model(), t, y, has
    model = cases.c2_model() if which == "c2" else cases.c1_model()
    t, y, has = cases.poisson_counts(T)
    return model, t, y, has


def workload_text(which, n_total, K, per_gpu=None, world=1):
    if which == "c4":
        s = (f"configs[3]: log-Gaussian Cox process (lgcp(ouProcess(1)), precision {LGCP_PRECISION}, ~10 sub-steps per event, d=1), "
             f"N={n_total} particles in total")
        return s + (f" over {world} GPUs (strong scaling: {per_gpu} per GPU)" if world > 1 else "") + f", T={K} events, systematic resampling every event"
    head = ("configs[1]: seasonal-Poisson, OU latent (poisson(ou(1)) |+| seasonal(24,1,ou(2)), d=3), " if which == "c2"
            else "configs[0] model at bench size: poisson(brownianMotion(1)), d=1, ")
    size = f"N={n_total} particles" if world == 1 else f"{per_gpu} particles per GPU x {world} GPUs = {n_total} (weak scaling)"
    return head + size + f", T={K} observations, systematic resampling every observation"


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


def cpu_baseline(budget_s=10.0, all_cores=True):
    """Oracle (CPU restatement of the reference path) on one core -- the reference runs one thread per filter
    (sequential Vector.map / foldLeft, model/ParticleFilter.scala:118,123,139) -- on a bounded sample; plus, as a stronger
    baseline, as many independent filters as this process may use cores, side by side."""
    n = N_PER_GPU                      # the configuration's own cloud (configs[1]: 2^20 particles); the sample is bounded in OBSERVATIONS
    dt0, _ = _cpu_sample((n, 2, 20260101))
    per = dt0 / (n * 2)
    Ts = int(max(4, min(400, budget_s / (per * n))))
    dt, ll = _cpu_sample((n, Ts, 20260101))
    out = {"value": n * Ts / dt, "unit": "particle-steps/s", "cores": 1, "kind": "port",
           "sample": f"oracle/cssm_oracle.c, same model/data/seed as the GPU run at the configuration's own N={n} particles, the first T={Ts} "
                     f"observations of the series, {dt:.1f} s on 1 of {os.cpu_count()} host cores", "ll": ll}
    if all_cores:
        try:
            import multiprocessing as mp
            cores = max(1, min(16, len(os.sched_getaffinity(0))))   # (a GPU box gives one GPU's share of the host: 16 cores)
            Tm = max(3, Ts // 3)
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
def _kernel_profile(pf, t, y, has, K, loop_ms_plain, legs=None):
    """Per-kernel durations of one K-step series: HIP events on the launch stream around every kernel (a pass of its
    own, so that the event records do not perturb the throughput figure), minus what a bracketing event pair adds.
    The event packets sit on the queue and inflate every bracketed launch; calibrated in place: the bracketed durations
    of all kernels of the series add up to more than the same loop takes without brackets, and the excess, split evenly
    over the brackets, is what one bracket adds (idle gaps of the plain loop stay inside the kernels' figures: the
    estimate errs on the long side)."""
    # legs: [(lo, hi), ...] -- the pass CONTINUES the running filter leg by leg, the very shape of the timed region (as many legs
    # of as many steps: with the driver's --steps 20 that is 140 launches per kernel to average over instead of 20, and every
    # leg's first, slower launch weighs as it does in the timed legs); None: one fresh series of K steps
    pf.profile(True)
    if legs:
        for lo, hi in legs:
            pf.run_more(t[lo:hi], y[lo:hi], has[lo:hi])
        loop_ms_plain = loop_ms_plain * len(legs)
    else:
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


def _offspring_roofline(n, per):
    """The resampling kernel against the HBM roofline: SURVEY.md 8d's K_resample = read w1 (8 B) + write anc (4 B) = 12 B per particle."""
    if "k_offspring" not in per:
        return None
    t = per["k_offspring"][0]
    return {"kernel": "k_offspring_wave / k_offspring_self (unit prefix, end slots, ancestor runs, ll, sum of squares)", "algorithmic_bytes_per_particle": 12,
            "avg_launch_us": t * 1e6, "achieved": 12 * n / t / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": 12 * n / t / 1e9 / HBM_PEAK_GBS}


def run_single(args, emit=print):
    import torch  # first: its import takes long enough for the GPU clocks to idle down
    from composablestatespacemodels_amd.filter import NativePf
    K, W = args.steps, args.warmup
    lgcp = args.model == "c4"
    n = args.particles if args.particles > 0 else (N_16M if lgcp else N_PER_GPU)
    R = args.repeats if args.repeats > 0 else (3 if K >= 200 else 25)
    W = max(W, 1)                      # (the filter has to be running before K more steps can be timed)
    model, t, y, has = build_workload(W + 7 * R * K + 16, args.model)
    import ctypes as C

    def handle():
        h = NativePf(model, n, 20260101, device=0, lgcp_precision=LGCP_PRECISION if lgcp else 0)
        if args.fused is not None:
            h.set_option(3, args.fused)   # CSSM_OPT_FUSED_SUMS (experiment switch; the default is the library's)
        return h

    def timed_legs(timed_pf=None):
        """The figure of merit: W untimed warm-up steps start a filter (initial cloud, allocations) and the timed leg CONTINUES it: R legs
        of exactly K more steps each (cssm_pf_ll_filter_more: no new cloud, the series goes on), every leg bracketed by a device
        synchronisation on both sides; the figure is the MEDIAN leg.  A leg is the whole host call: records built and sent, 2 K launches
        enqueued, ll / ess read back.  Per leg also: the host time of the call alone (it returns on the completion word its closing kernel
        stores behind the K steps' results; the closing torch.cuda.synchronize() then waits for the runtime to see that kernel's completion
        signal -- measured in round 6: hipStreamQuery still says "not ready" when the call returns, the end-of-kernel release of the closing
        kernel is what the synchronise waits 4-6 us for) and the DEVICE time of the leg from the GPU's own clock."""
        pf = timed_pf if timed_pf is not None else handle()
        gc.disable()       # (as timeit does: a full collection of a process with torch imported takes ~50 ms -- seen inside one timed leg
        #                     of the sharded bench, always the same one; re-enabled behind the legs.  No gc.collect() here: it idles the
        #                     GPU for those 50 ms, and legs 2-5 ms behind an idle gap run up to 15 % slower -- tried, three runs of the
        #                     driver's command: 4.43 / 4.06 / 4.03e10 against 4.5-4.7e10 without)
        torch.cuda.synchronize()
        pf.run(t[:W], y[:W], has[:W])
        walls, devs, calls, idle = [], [], [], []
        ll = ess_t = None
        for r in range(R):
            lo = W + r * K
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            ll, _, ess_t = pf.run_more(t[lo:lo + K], y[lo:lo + K], has[lo:lo + K])
            t1 = time.perf_counter()
            torch.cuda.synchronize()
            t2 = time.perf_counter()
            walls.append(t2 - t0)
            calls.append(t1 - t0)
            idle.append(bool(pf.stream_idle()))          # (behind the synchronise: must be true)
            devs.append(pf.last_device_us() * 1e-6)      # (read behind the timed region: the GPU's own clock, no event packets)
        gc.enable()
        if not all(idle):
            raise SystemExit("bench.py: the filter's stream was not idle behind torch.cuda.synchronize()")
        d = pf.d
        pf.close()
        return walls, ll, ess_t, d, devs, calls

    def roofline_legs():
        """Kernel times of legs of the same shape on a handle of their own: the streaming ceiling of the box, the device time of R legs (HIP
        events around their per-observation kernels: CSSM_OPT_LOOP_EVENTS -- off in the timed legs, the two event packets cost a 20-step
        call ~5 us), R legs with every kernel bracketed, and the same with the structure specialisation switched off."""
        pf = handle()
        copy = C.c_double(0.0)
        copy_gbs = None
        # (plain 16-byte-per-lane copy, 1 GiB each way: far beyond the 256 MB Infinity Cache)
        if pf.lib.cssm_diag_copy_ceiling(0, 1 << 30, 10, C.byref(copy)) == 0:
            copy_gbs = copy.value
        pf.run(t[:W], y[:W], has[:W])
        pf.set_option(9, 1)
        # R legs nobody reads (the first legs of a process run 5-15 % slower than the ones behind them, and the bracket's cost is
        # calibrated as a DIFFERENCE of the two passes below: both have to be taken in the same state), then the two passes
        S = 3 * R if K < 200 else R      # (short legs: the per-leg times of a fresh process level off after ~15-20 legs)
        for r in range(S):
            lo = W + r * K
            pf.run_more(t[lo:lo + K], y[lo:lo + K], has[lo:lo + K])
        loops = []
        for r in range(R):
            lo = W + (S + r) * K
            pf.run_more(t[lo:lo + K], y[lo:lo + K], has[lo:lo + K])
            loops.append(pf.last_loop_ms())
        loop_ms = float(np.median(loops))
        per, pair_s, prof_raw = _kernel_profile(pf, t, y, has, K, loop_ms, legs=[(W + (S + R + r) * K, W + (S + R + r + 1) * K) for r in range(R)])
        # the same model and legs with the structure specialisation switched off (CSSM_OPT_SPECIALISE = 0: the kernel that reads the model's
        # structure as data, what every model outside BASELINE's configurations ran until round 4 gave each its own run-time-compiled kernel)
        roof_generic = None
        if not args.no_generic:
            pf.set_option(8, 0)
            lo = W + (S + 2 * R) * K
            pf.run_more(t[lo:lo + 8], y[lo:lo + 8], has[lo:lo + 8])
            lo += 8
            gl = []
            for r in range(R):
                pf.run_more(t[lo + r * K:lo + (r + 1) * K], y[lo + r * K:lo + (r + 1) * K], has[lo + r * K:lo + (r + 1) * K])
                gl.append(pf.last_loop_ms())
            gper, gpair, _ = _kernel_profile(pf, t, y, has, K, float(np.median(gl)), legs=[(lo + (R + r) * K, lo + (R + r + 1) * K) for r in range(R)])
            ga, gc, gr = gper["k_propagate"]
            roof_generic = _roofline(f"k_propagate<{pf.d},...> reading the model's structure as data (CSSM_OPT_SPECIALISE = 0), N={n}", pf.d, n, ga, gc, gr, gpair, None)
            roof_generic["step_us"] = float(np.median(gl)) * 1e3 / K
        keep.append(pf)                  # (closed behind the timed legs: a hipFree is a device synchronisation and an idle gap)
        return copy_gbs, loop_ms, per, pair_s, roof_generic

    # Order: the roofline legs FIRST.  W = 5 warm-up steps are 0.1 ms of device work, and the legs of a process that has kept the GPU busy
    # for a few ms in all run 5-15 % slower than every later one (leg after leg of the same handle: 522 493 461 465 468 ... 445 us, level
    # after ~15 legs): timed first, the legs of 20 steps measure that ramp, not the filter.  BENCH_TIMED_FIRST=1 restores that order.
    # ... and the timed filter's handle is CREATED ahead of them and the roofline handle closed behind the timed legs, so that no allocation, no
    # hipFree and no idle gap stands between the roofline legs and the W warm-up steps: a GPU that has idled for a millisecond or more runs the
    # legs 1.5-5 ms into new work up to 14 % slower before it settles (tools/leg_drift.py, device clock stamps: 434 434 445 456 467 496 491
    # 481 472 456 448 436 ... 425 us per 20-observation leg; uncorrelated with the data, back after 2 s of idling) -- the rising legs of
    # BENCH_r05.  The timed region is what the contract says: W warm-up steps, then R legs of exactly K.
    keep = []
    if os.environ.get("BENCH_TIMED_FIRST", "0") == "1":
        walls, ll, ess_t, d, devs, calls = timed_legs()
        copy_gbs, loop_ms, per, pair_s, roof_generic = roofline_legs()
    else:
        timed_pf = handle()
        copy_gbs, loop_ms, per, pair_s, roof_generic = roofline_legs()
        walls, ll, ess_t, d, devs, calls = timed_legs(timed_pf)
    for h in keep:
        h.close()
    wall = float(np.median(walls))
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
    if not args.no_pmc and not lgcp:
        live, why = live_traffic(n, args.model, fused=args.fused)
        if live is not None:
            traffic, traffic_source = live, why
        else:
            traffic_source = (traffic_source or "") + f" [live PMC passes unavailable: {why}]"
    avg_s, cnt, raw_s = per["k_propagate"]
    roof = _roofline(f"k_propagate<{d},...> (gather + propagate + weight + log-sum-exp sums), N={n}", d, n, avg_s, cnt, raw_s, pair_s, copy_gbs)
    roof["timing"] = ("HIP events on the launch stream around every k_propagate launch of `repeats` more legs of K steps each "
                      "(the timed region's shape, continued), minus what a bracketing event pair adds: (sum of all bracketed "
                      "kernel times - device time of the same legs without brackets) / brackets")
    kernels_us = {k: v[0] * 1e6 for k, v in per.items()}
    roof["traffic"] = traffic
    roof["traffic_source"] = traffic_source
    out = {
        "metric": METRIC, "value": n * K / wall, "unit": "particle-steps/s",
        "n_gpus": 1, "steps": K, "warmup": W, "ms_per_step": wall * 1e3 / K, "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": workload_text(args.model, n, K),
                   "particles_per_gpu": n, "observations": K, "latent_dim": d, "seed": 20260101},
        "repeats": R, "value_is": "median over `repeats` timed legs of K steps each (every leg counts, none is discarded: the legs 2-5 ms behind the idle gap in front of the filter run slower on the DEVICE -- device_ms_each), continuing the filter the warm-up steps started (cssm_pf_ll_filter_more); the roofline legs (a handle of their own) run before them, the timed filter's handle is created ahead of those and the roofline handle closed behind the timed legs (no allocation, hipFree or idle gap in front of the W warm-up steps)", "wall_ms_each": [w * 1e3 for w in walls],
        "roofline": roof,
        "roofline_generic": roof_generic,
        "kernels_us": kernels_us,
        "roofline_resampling": _offspring_roofline(n, per),
        "device_ms_each": [x * 1e3 for x in devs],
        "device_ms_each_is": "per timed leg, the GPU's constant 100 MHz clock (s_memrealtime) at the first instruction of the leg's first kernel and in its closing kernel (cssm_pf_last_device_us): device time of the very legs wall_ms_each times, no event packets on the queue",
        "call_ms_each": [x * 1e3 for x in calls],
        "call_ms_each_is": "host time of cssm_pf_ll_filter_more alone (the K steps' results are in host memory when it returns); wall_ms_each = torch.cuda.synchronize() -> call -> torch.cuda.synchronize()",
        "host_share": 1.0 - float(np.median(devs)) / wall,
        "device_loop_ms": loop_ms, "ll": ll, "ess_last": int(ess_t[-1]),
        "runtime": {"HSA_ENABLE_INTERRUPT": os.environ.get("HSA_ENABLE_INTERRUPT"), "note": "the ROCm runtime's completion wait as the caller left it (round 5 set 0 = polling; round 6's A/B on this very command, three interleaved pairs on one box: 22.55 / 23.15 / 22.87 us per step polling, 22.79 / 23.15 / 22.48 with interrupts -- no difference, setting removed: profiles/r06_ab_hsa_interrupt.txt)"},
    }
    if lgcp:
        roof["note"] = ("an LGCP step runs its ~10 sub-steps of Philox + Box-Muller + exp per particle in registers: the kernel is compute-bound by "
                        "design (SURVEY.md 8d); the HBM fraction is reported for completeness -- its bound is the vector-ALU issue rate: roofline_valu")
        out["roofline_valu"] = _roofline_valu(n, avg_s)
    if not args.no_16m and not lgcp:
        out["roofline_16m"] = roofline_16m(NativePf, copy_gbs, args.fused)
    if not args.no_cpu:
        out["cpu_baseline"] = cpu_baseline()
    emit(json.dumps(out))


def live_traffic(n, model, timeout_s=150.0, fused=None):
    """HBM bytes per launch of the fused kernel, MEASURED in this run: two rocprofv3 passes (FETCH_SIZE, then WRITE_SIZE: the two do not fit
    one pass on gfx950) around a short child `bench.py` of the same workload, each with `--pmc <counter> --kernel-trace` only, from /tmp.
    Bytes = 2 x FETCH_SIZE + WRITE_SIZE (both reported in KiB): the factor 2 is the gfx950 correction of MI355X_MICROARCH.md (FETCH_SIZE
    tallies its 128-byte requests at 64 bytes), WRITE_SIZE reads true for 16-byte-per-lane streaming stores.  Children of this process (it
    keeps its GPU context; nothing is exec'ed in place), each under a time limit; None -- and the reason -- if the profiler is missing, a
    pass fails or no k_propagate row comes back: the line then falls back to the committed figure and says so."""
    import csv
    import glob
    import shutil
    import tempfile
    exe = shutil.which("rocprofv3") or ("/opt/rocm/bin/rocprofv3" if os.path.exists("/opt/rocm/bin/rocprofv3") else None)
    if exe is None:
        return None, "rocprofv3 not found"
    got = {}
    work = tempfile.mkdtemp(prefix="cssm_pmc_", dir="/tmp")
    env = dict(os.environ, TMPDIR="/tmp")
    try:
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            out = os.path.join(work, counter)
            cmd = [exe, "--pmc", counter, "--kernel-trace", "--output-format", "csv", "-d", out, "--", sys.executable, os.path.abspath(__file__),
                   "--steps", "20", "--warmup", "2", "--repeats", "1", "--particles", str(n), "--model", model,
                   "--no-cpu", "--no-16m", "--no-generic", "--no-pmc"] + ([] if fused is None else ["--fused", str(fused)])   # (the kernel variant that is timed)
            try:
                r = subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, text=True, timeout=timeout_s)
            except subprocess.TimeoutExpired:
                return None, f"the {counter} pass did not finish within {timeout_s:.0f} s"
            if r.returncode != 0:
                return None, f"the {counter} pass exited with {r.returncode}: {r.stderr.strip().splitlines()[-1][:200] if r.stderr.strip() else ''}"
            # the launches of the child's TIMED leg only: its last 20 k_propagate_self launches in dispatch order (its warm-up and roofline
            # legs come first; every launch moves the same bytes, the filter makes the average mean what it says)
            vals = []
            for path in glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True):
                rows = [r for r in csv.DictReader(open(path)) if r["Counter_Name"] == counter and "k_propagate_self<" in r["Kernel_Name"]]
                rows.sort(key=lambda r: int(r.get("Dispatch_Id", 0) or 0))
                vals += [float(r["Counter_Value"]) for r in rows[-20:]]
            tot, cnt = sum(vals), len(vals)
            if cnt == 0:
                return None, f"the {counter} pass returned no k_propagate_self row"
            got[counter] = (tot / cnt, cnt)
    finally:
        shutil.rmtree(work, ignore_errors=True)
    bytes_per_launch = int(2 * got["FETCH_SIZE"][0] * 1024 + got["WRITE_SIZE"][0] * 1024)
    return bytes_per_launch, (f"measured in this run: rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes around a 20-observation child run of "
                              f"the same workload, {got['FETCH_SIZE'][1]} / {got['WRITE_SIZE'][1]} launches), bytes = 2 x FETCH_SIZE + WRITE_SIZE "
                              f"(KiB; gfx950 correction of the FETCH_SIZE tally)")


def _roofline_valu(n, avg_s):
    """configs[3]'s real bound: the LGCP kernel issues ~1000 vector-ALU instructions per particle and event (Philox-7, Box-Muller, exp and the
    OU transition of ~10 sub-steps, all in registers) against 24 bytes of HBM traffic.  achieved = VALU wave-instructions per second -- the
    count per particle and event from the committed PMC pass (SQ_INSTS_VALU; profiles/valu_latest.json: NOT measured in this run, the
    instruction count of a kernel does not depend on the box), the kernel time measured live with HIP events; peak = one fp64 instruction
    per 4.82 cycles and SIMD (profiles/r01s2_instr_rate.txt: v_fma_f64 at 8 waves per SIMD) on 256 CUs x 4 SIMDs at 2.4 GHz.  Part of the
    stream is 32-bit integer work (Philox), which issues in 2.4 cycles: `frac` can exceed 1 -- `avg_cycles_per_inst` says where between the
    two rates the kernel runs; both at once means the SIMDs issue back to back."""
    vpath = os.path.join(ROOT, "profiles", "valu_latest.json")
    if not os.path.exists(vpath):
        return None
    vj = json.load(open(vpath))
    simds, clock = 256 * 4, 2.4e9
    insts = vj["valu_wave_insts_per_particle_event"] * n
    achieved = insts / avg_s
    peak = simds * clock / 4.82
    return {"bound": "valu", "kernel": vj["kernel"] + f", N={n}", "achieved": achieved / 1e9, "peak": peak / 1e9, "unit": "G wave-instructions/s",
            "frac": achieved / peak, "avg_launch_us": avg_s * 1e6, "valu_lane_insts_per_particle_event": vj["valu_lane_insts_per_particle_event"],
            "avg_cycles_per_inst": avg_s * clock * simds / insts, "fp64_issue_cycles": 4.82, "int32_issue_cycles": 2.4,
            "frac_of_int32_rate": achieved / (simds * clock / 2.4),   # (the other end: every slot a 32-bit instruction)
            "insts_source": "NOT measured in this run: " + vj["source"]}


def roofline_16m(NativePf, copy_gbs, fused=None, K=24):
    """The fused kernel at the north-star size N = 2^24, driver-timed: the bench model (d = 3, 56 B per particle) and
    Poisson-Brownian (d = 1, 24 B per particle: SURVEY.md 8d's reading of the target).  Two variants each:
    the DEFAULT kernel -- gather + propagate + weight + the log-sum-exp sums (SURVEY 8d's K_fused; CSSM_OPT_FUSED_SUMS = 1) --
    and, under ``*_lean``, the same kernel without the sums (CSSM_OPT_FUSED_SUMS = 0: they are then a pass of their own,
    k_tile_sums), which is the kernel round 1 quoted.  Same algorithmic bytes (the sums add none), more arithmetic."""
    res = {}
    for key, which in (("c2_d3", "c2"), ("c1_d1", "c1")):
        model, t, y, has = build_workload(K, which)
        # "" = the default path; "_wave_ranges" / "_block_tiles" = the OTHER mapping of the propagate's blocks (round 6: wave ranges + k_offspring_wave
        # against block-wide tiles + k_offspring_self; the default takes wave ranges beyond 2^20 particles at d <= 2, where the step gains:
        # profiles/r06_ab_wave_sums.txt) -- both are in the line so that one run shows the trade; "_lean" = without the fused sums
        fzd = 1 if fused is None else fused
        other = ("_block_tiles", 0) if which == "c1" else ("_wave_ranges", 2)
        for variant, fz, ws in (("", fzd, 1), (other[0], fzd, other[1]), ("_lean", 0, 1)):
            pf = NativePf(model, N_16M, 20260101, device=0)
            pf.set_option(3, fz)
            pf.set_option(10, ws)         # CSSM_OPT_WAVE_SUMS
            pf.set_option(9, 1)            # CSSM_OPT_LOOP_EVENTS: these legs are timed on the device
            pf.run(t[:8], y[:8], has[:8])
            loop_ms = 1e30
            for _ in range(3):            # (the first series on a 1 GB handle is not representative: best of three)
                pf.run(t[:K], y[:K], has[:K])
                loop_ms = min(loop_ms, pf.last_loop_ms())
            per, pair_s, _ = _kernel_profile(pf, t, y, has, K, loop_ms)
            avg_s, cnt, raw_s = per["k_propagate"]
            r = _roofline(f"k_propagate<{pf.d},...> {'with' if fz else 'without'} the fused sums, N={N_16M}", pf.d, N_16M, avg_s, cnt, raw_s, pair_s, copy_gbs)
            r["fused_sums"] = bool(fz)
            r["wave_ranges"] = bool(fz and (ws == 2 or (ws == 1 and pf.d <= 2)))
            r["step_us"] = loop_ms * 1e3 / K
            r["particle_steps_per_s"] = N_16M * K / (loop_ms * 1e-3)
            r["kernels_us"] = {k: v[0] * 1e6 for k, v in per.items()}
            r["resampling_kernel"] = _offspring_roofline(N_16M, per)
            res[key + variant] = r
            pf.close()
    return res


# ------------------------------------------------------------------------------------------------ N ranks
class _Deadline:
    """A stage of a rank that must end: if it has not after `seconds`, the rank says where it hung and leaves with a non-zero status
    (os._exit from a watchdog thread: no cleanup can hang, nothing is exec'ed) -- the launcher (torch.distributed.run, or launch_ranks
    below) then ends the other ranks and reports failure.  Far above anything a healthy stage takes; the device-side waits of the
    peer-written exchange give up long before (CSSM_PEER_TIMEOUT_MS, 30 s).  ONE watchdog thread per process, started before anything is
    timed (a thread started inside the first timed leg cost that leg 40-60 ms); entering a stage only posts its name and deadline."""
    _stage = None          # (what, deadline on time.monotonic()) of the stage this process is in
    _thread = None

    def __init__(self, seconds, what):
        self.seconds, self.what = seconds, what

    @classmethod
    def _watch(cls):
        while True:
            time.sleep(0.25)
            st = cls._stage
            if st is not None and time.monotonic() > st[1]:
                sys.stderr.write(f"bench.py: rank {os.environ.get('RANK', '0')}: '{st[0]}' did not finish within {st[2]:g} s; leaving\n")
                sys.stderr.flush()
                os._exit(7)

    def __enter__(self):
        import threading
        cls = _Deadline
        if self.seconds > 0:
            if cls._thread is None:
                cls._thread = threading.Thread(target=cls._watch, daemon=True)
                cls._thread.start()
            cls._stage = (self.what, time.monotonic() + self.seconds, self.seconds)
        return self

    def __exit__(self, *exc):
        _Deadline._stage = None
        return False


def preflight(f, make_filter, t, y, has, lgcp, gpu, rank, world, stage_s):
    """Which exchange protocol the timed legs run on, decided by every rank alike BEFORE anything is timed, with the reasons:
    peer-written windows -> collectives issued by libcssm_pf over RCCL -> collectives through torch.distributed.  A protocol is taken
    only if (a) every rank could set it up (the library's own agreement: windows mapped, two handshake rounds with probe words read
    back) and (b) a short series on it returns the SAME BITS (ll, ess) on every rank as the same series on the next protocol down --
    the results of the sharded filter do not depend on how the segments travel, so any difference is a transport fault, and the walk
    moves on instead of timing wrong numbers.  Returns (filter to use, steps): steps = [{"protocol", "ok", "why"}, ...] in walk order."""
    import torch.distributed as dist
    steps = []
    Tp = min(6, len(t))

    def run_series(ff):
        ll, ess = ff.ll_filter(t[:Tp], y[:Tp], has[:Tp], lgcp=lgcp)
        return (float(ll).hex(), int(ess))

    def everyone(value):
        box = [None] * world
        dist.all_gather_object(box, value)
        return box

    results = {}          # protocol -> this rank's (ll bits, ess)
    saved = {k: os.environ.get(k) for k in ("CSSM_SHARD_PEER", "CSSM_SHARD_NATIVE")}
    order = ["peer", "rccl", "torch"]
    usable = {}
    for proto in order:
        os.environ["CSSM_SHARD_PEER"] = "1" if proto == "peer" else "0"
        os.environ["CSSM_SHARD_NATIVE"] = "0" if proto == "torch" else "1"
        if proto == "peer" and saved["CSSM_SHARD_PEER"] == "0":
            steps.append({"protocol": proto, "ok": False, "why": "switched off by CSSM_SHARD_PEER=0"})
            continue
        if proto == "rccl" and saved["CSSM_SHARD_NATIVE"] == "0":
            steps.append({"protocol": proto, "ok": False, "why": "switched off by CSSM_SHARD_NATIVE=0"})
            continue
        ff = make_filter()
        why, mine = "", None
        try:
            with _Deadline(stage_s, f"pre-flight series on the {proto} protocol"):
                mine = run_series(ff)
            if proto == "peer" and not getattr(ff, "last_peer", False):
                why = "refused: " + str(getattr(ff, "peer_refused", None) or ("not a GPU rank" if not gpu else "the series did not run on the windows"))
                mine = None
            elif proto == "rccl" and not getattr(ff, "last_native", False):
                why = "refused: " + ("backend is not RCCL" if not gpu else "no library-owned communicator")
                mine = None
        except Exception as e:        # noqa: BLE001 -- raised on every rank alike (e.g. libcssm_pf could not make its own RCCL communicator; a backend
            why, mine = "refused: " + (str(e).splitlines() or [repr(e)])[0][:300], None   # that cannot move device buffers): the walk moves on
        got = everyone(mine)
        if any(g is None for g in got):
            steps.append({"protocol": proto, "ok": False, "why": why or "refused on another rank"})
            continue
        if any(g != got[0] for g in got):
            steps.append({"protocol": proto, "ok": False, "why": f"ranks disagree on the {Tp}-observation series: {got}"})
            continue
        results[proto] = got[0]
        usable[proto] = ff
        extra = ""
        if proto == "peer":
            extra = f"; handshake probe words read stale by plain loads: {getattr(ff, 'peer_probe_stale', 'n/a')}"
        steps.append({"protocol": proto, "ok": True, "why": f"{Tp}-observation series ll = {got[0][0]}, ess = {got[0][1]} on every rank" + extra})
    # the first protocol of the walk whose bits equal those of EVERY usable protocol below it (torch.distributed is the yardstick: the
    # plainest path); a protocol that differs is marked and skipped
    chosen = None
    for i, proto in enumerate(order):
        if proto not in results:
            continue
        lower = [q for q in order[i + 1:] if q in results]
        bad = [q for q in lower if results[q] != results[proto]]
        if bad:
            for st in steps:
                if st["protocol"] == proto:
                    st["ok"], st["why"] = False, f"its bits {results[proto]} differ from the {bad[0]} protocol's {results[bad[0]]}: transport fault, not used"
            continue
        chosen = proto
        break
    if chosen is None:
        raise SystemExit("bench.py: no exchange protocol is usable on every rank: " + json.dumps(steps))
    os.environ["CSSM_SHARD_PEER"] = "1" if chosen == "peer" else "0"
    os.environ["CSSM_SHARD_NATIVE"] = "0" if chosen == "torch" else "1"
    if rank == 0:
        for st in steps:
            sys.stderr.write(f"bench.py pre-flight: {st['protocol']:5s} {'ok    ' if st['ok'] else 'NOT ok'} {st['why']}\n")
        sys.stderr.write(f"bench.py pre-flight: world {world}: the timed legs run on the '{chosen}' protocol\n")
        sys.stderr.flush()
    return usable[chosen], steps, chosen


def _rccl_transports(rank):
    """{peer: [transports RCCL chose for its channels to that peer]} from this process's NCCL_DEBUG_FILE ("... 0[0] -> 1[1] via P2P/IPC/read"),
    or a reason.  Written while the communicators of the pre-flight were set up (torch's and the library's own)."""
    import glob
    import re
    pat = os.environ.get("NCCL_DEBUG_FILE", "")
    if not pat:
        return None
    paths = glob.glob(pat.replace("%h", "*").replace("%p", str(os.getpid())))
    if not paths:
        return {"error": "RCCL wrote no debug file (NCCL_DEBUG overridden?)"}
    via = {}
    try:
        for path in paths:
            for ln in open(path, errors="replace"):
                m = re.search(r"(\d+)\[\w+\] -> (\d+)\[\w+\].*? via (\S+)", ln)
                if m and int(m.group(1)) == rank:
                    via.setdefault(m.group(2), set()).add(m.group(3))
    except OSError as e:
        return {"error": repr(e)}
    return {k: sorted(v) for k, v in sorted(via.items(), key=lambda kv: int(kv[0]))} or {"note": "no channel lines in RCCL's debug output"}


def _leg_device_ms(shard):
    """This rank's last leg on the GPU's own clock, or None where the protocol the walk ended on leaves no pair of stamps (a leg must not
    fail over a diagnostic)."""
    try:
        return round(shard.last_device_us() * 1e-3, 4)
    except Exception:   # noqa: BLE001
        return None


def run_multi(args, emit=print):
    import torch
    import torch.distributed as dist
    from composablestatespacemodels_amd.sharded import DistComm, ShardedFilter
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", str(args.gpus)))
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29511")
    gpu = args.backend in ("nccl", "gloo-gpu")
    # gloo-gpu: a REHEARSAL of the N-rank GPU path on fewer GPUs than ranks -- every rank a real GpuShard (all of them on the devices
    # there are, rank % device count), the peers' windows mapped over hipIpc, the control plane on gloo (RCCL refuses two ranks on one
    # device): the launcher, the pre-flight walk, the handshakes and the timed legs run as they will on a node, the figure means nothing
    shared_gpu = args.backend == "gloo-gpu"
    if shared_gpu:
        local = local % max(torch.cuda.device_count(), 1)
    if gpu and not shared_gpu:
        torch.cuda.set_device(local)
        # (RCCL's own account of how it reaches every peer -- xGMI P2P, PCIe, host shared memory -- is read from NCCL_DEBUG_FILE where the CALLER
        #  has switched it on: NCCL_DEBUG=INFO NCCL_DEBUG_SUBSYS=INIT,GRAPH,P2P NCCL_DEBUG_FILE=/tmp/rccl_%h_%p.log.  Not by default: with INFO
        #  logging on, one timed leg in seven took 60-100 ms on the test box -- measured in round 6)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
    else:   # rehearsals (tests): gloo -- with the test-only oracle shard on CPU, or (gloo-gpu) GPU shards
        dist.init_process_group("gloo", rank=rank, world_size=world)
        if shared_gpu:
            torch.cuda.set_device(local)
    K, W = args.steps, args.warmup
    if args.particles < 0:
        raise SystemExit("--particles must be positive (0: the configuration's default)")
    lgcp = args.model == "c4"
    if lgcp:      # configs[3]: a fixed cloud split over the GPUs (strong scaling)
        n_global = args.particles if args.particles > 0 else N_16M
        per_gpu = -(-n_global // world)
    else:         # configs[1]: a fixed cloud per GPU (weak scaling)
        per_gpu = args.particles if args.particles > 0 else N_PER_GPU
        n_global = per_gpu * world
    R = args.repeats if args.repeats > 0 else (3 if K >= 200 else 25)
    # warm-up: allocations, RCCL channels, clocks.  At least 8 observations whatever W is: the library's own RCCL communicator and
    # the exchange buffers are created at the first observation of the first series -- that must not happen inside a timed leg
    Wn = max(W, 8) if gpu else max(W, 1)
    model, t, y, has = build_workload(Wn + (R + 1) * K, args.model)
    prec = LGCP_PRECISION if lgcp else 0
    if gpu:
        from composablestatespacemodels_amd.sharded import GpuShard
        shard = GpuShard(model, n_global, rank, world, 20260101, local, lgcp_precision=prec)
        if shared_gpu:
            class GlooPeerComm(DistComm):          # control plane on the CPU (gloo), windows on the GPU (IPC)
                peer = property(lambda self: os.environ.get("CSSM_SHARD_PEER", "1") != "0")
            make_filter = lambda: ShardedFilter([shard], GlooPeerComm())   # noqa: E731
        else:
            make_filter = lambda: ShardedFilter([shard], DistComm(device=torch.device("cuda", local)))   # noqa: E731
    else:
        from oracle_shard import OracleShard
        shard = OracleShard(model, n_global, rank, world, 20260101, prec)
        make_filter = lambda: ShardedFilter([shard], DistComm())   # noqa: E731
    sync = torch.cuda.synchronize if gpu else (lambda: None)
    dev = "cuda" if (gpu and not shared_gpu) else "cpu"
    stage_s = float(args.stage_timeout)
    # pre-flight: the protocol of the timed legs, agreed and cross-checked (peer-written -> RCCL issued by the library -> torch.distributed)
    f, walk, protocol = preflight(make_filter(), make_filter, t, y, has, lgcp, gpu, rank, world, stage_s)
    fallbacks = [f"{st['protocol']}: {st['why']}" for st in walk if not st["ok"]]

    gc.disable()           # (see run_single: a full collection inside a timed leg costs it ~50 ms; re-enabled behind the legs)
    # the W warm-up observations START the sharded filter ...
    with _Deadline(stage_s, "warm-up series"):
        f.ll_filter(t[:Wn], y[:Wn], has[:Wn], lgcp=lgcp)
    # ... and every timed leg CONTINUES it with exactly K more (the sharded cssm_pf_ll_filter_more: records built and uploaded,
    # kernels and collectives enqueued by the library, one status read per stretch, ll / ess read back -- no new cloud, no k_init),
    # bracketed by a barrier + device synchronisation on both sides; a leg's time is the MAX over the ranks, the figure the
    # MEDIAN leg
    walls, plans = [], []
    ll = ess = None
    # the collectives that bracket a leg, once, untimed: a communicator sets its channels up at its first use
    dist.barrier()
    warm = torch.zeros(1, dtype=torch.float64, device=dev)
    dist.all_reduce(warm, op=dist.ReduceOp.MAX)
    sync()
    for r in range(R):
        lo = Wn + r * K
        dist.barrier()
        sync()
        with _Deadline(stage_s, f"timed leg {r}"):
            t0 = time.perf_counter()
            ll, ess = f.ll_filter_more(t[lo:lo + K], y[lo:lo + K], has[lo:lo + K], lgcp=lgcp)
            sync()
            t1 = time.perf_counter()     # this rank's K steps are done; the leg's time is the MAX over the ranks (below), behind the barrier
            dist.barrier()
        wall = torch.tensor([t1 - t0], dtype=torch.float64, device=dev)
        dist.all_reduce(wall, op=dist.ReduceOp.MAX)
        walls.append(float(wall.item()))
        plans.append({"plan": "max" if f.last_from_max else ("ref" if f.last_single else "exact"), "resumes": int(f.last_resumes),
                      "level_redos": int(getattr(f, "last_level_redos", 0)), "capacity_rows": f.last_cap, "wall_ms": round((t1 - t0) * 1e3, 4),
                      # this rank's leg on the GPU's own clock, first kernel of the call to k_finish (no event packets): wall_ms - device_ms is host
                      "device_ms": _leg_device_ms(shard) if gpu else None})
    gc.enable()
    # a pass of its own with HIP events around every kernel and every library-issued collective of this rank's stream (the
    # event records perturb the throughput figure): what a step is made of, per rank
    mine = {"rank": rank, "legs": plans}
    if gpu and not shared_gpu and os.environ.get("NCCL_DEBUG_FILE"):
        mine["rccl_transport"] = _rccl_transports(rank)       # (read at the end: RCCL sets channels up lazily, at a communicator's first use)
    if gpu:
        lo = Wn + R * K
        shard.profile(True)
        f.ll_filter_more(t[lo:lo + K], y[lo:lo + K], has[lo:lo + K], lgcp=lgcp)
        prof = shard.profile_read()
        shard.profile(False)
        mine["kernels_us"] = {k: round(v[0] / v[1] * 1e3, 2) for k, v in prof.items() if v[1] and k != "collective"}
        mine["collective_us"] = round(prof["collective"][0] / prof["collective"][1] * 1e3, 2) if prof["collective"][1] else None
        mine["collectives_per_observation"] = round(prof["collective"][1] / max(prof["k_propagate"][1], 1), 2)
        mine["note"] = "bracketed event times include ~2 us of event-record overhead per launch; a collective's time includes waiting for the slowest peer"
        if getattr(f, "last_peer", False):
            # what ONE block per exchange launch spent polling its peers (the GPU's own clock; warm-up and timed legs): at world 1 a block
            # waits for its own launch's header block, across GPUs the link and the peers' lateness show here and not in kernels_us
            nx, hw, xw, rw = shard.wait_stats()
            mine["peer_waits"] = {"exchanges": nx, "header_wait_us": hw and round(hw, 2), "expansion_block_header_wait_us": xw and round(xw, 2),
                                  "rows_wait_us": rw and round(rw, 2)}
            # what went over the links: rows this rank wrote into its neighbours' windows -- per neighbour segment and observation the
            # eager rows (written at once) or, where more were needed, the needed ones; pre-flight and warm-up included
            nrows, nseg, beyond = shard.peer_rows()
            rb = (shard.d + 2) * 8
            mine["rows_to_neighbours"] = {"rows": nrows, "segments": nseg, "segments_needing_rows_beyond_the_eager_ones": beyond,
                                          "mean_rows_per_segment": round(nrows / nseg, 1) if nseg else None,
                                          "mean_bytes_per_segment": round(nrows / nseg * rb) if nseg else None, "capacity_rows": f.last_cap,
                                          "capacity_bytes": f.last_cap * rb}
    every = [None] * world
    dist.all_gather_object(every, mine)
    if rank == 0:
        w = float(np.median(walls))
        native = bool(getattr(f, "last_native", False))
        peer = bool(getattr(f, "last_peer", False))
        emit(json.dumps({
            "metric": METRIC, "value": n_global * K / w, "unit": "particle-steps/s",
            "n_gpus": world, "steps": K, "warmup": W, "ms_per_step": w * 1e3 / K, "higher_is_better": True,
            "scaling": "strong" if lgcp else "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": workload_text(args.model, n_global, K, per_gpu, world) +
                                   " (global systematic resampling, one exchange per observation carrying every rank's sum words and, between adjacent "
                                   "ranks, its boundary particles -- " +
                                   ("written by every rank straight into its peers' receive windows, no collective" if peer else
                                    ("one RCCL all-to-all issued by the library" if native else "one all-to-all through torch.distributed")) +
                                   ("; LGCP: every event's level predicted from the max of the event before (numerics contract v8)" if lgcp else "") + ")",
                       "particles_per_gpu": per_gpu, "particles_total": n_global, "observations": K, "latent_dim": shard.d, "seed": 20260101},
            "repeats": R, "value_is": "median over `repeats` timed legs of K steps each (max over ranks per leg), continuing the sharded filter the warm-up "
                                      "steps started (cssm_pf_shard_continue)", "wall_ms_each": [x * 1e3 for x in walls],
            "exchange": {"backend": args.backend,
                         "shard_backend": ("libcssm_pf (HIP), ranks SHARING the GPUs there are (rehearsal, not a measurement)" if shared_gpu else "libcssm_pf (HIP)") if gpu else "oracle (CPU rehearsal, not a measurement)",
                         "protocol": ("peer-written: every rank writes its segments into the other ranks' receive windows (hipIpc-mapped device memory) and "
                                      "sets a flag; no collective per observation (cssm_pf_shard_series_peer)") if peer else "collective: one all-to-all per observation",
                         "collectives_issued_by": ("none per observation" if peer else
                                                   ("libcssm_pf (cssm_pf_shard_series_rccl)" if native else "torch.distributed")),
                         "all_to_all": getattr(f, "last_all_to_all", "equal split"),
                         "preflight": walk, "chosen": protocol,
                         "fallbacks": fallbacks,
                         "rccl": shard.lib.cssm_rccl_library().decode() if gpu else None},
            "per_rank": every,
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
    ap.add_argument("--particles", type=int, default=0, help="particles per GPU (c2 / c1: default 2^20) or in total (c4: default 2^24)")
    ap.add_argument("--repeats", type=int, default=0, help="timed K-step series (0: 3 for K >= 200, else 25); the median is reported")
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--no-16m", action="store_true", help="skip the roofline_16m leg")
    ap.add_argument("--no-pmc", action="store_true", help="skip the live HBM-traffic passes (two rocprofv3 --pmc child runs); roofline.traffic then comes from profiles/traffic_latest.json")
    ap.add_argument("--no-generic", action="store_true", help="skip the roofline_generic leg (the same kernel with the model's structure as data)")
    ap.add_argument("--fused", type=int, default=None, help="CSSM_OPT_FUSED_SUMS override (single GPU)")
    ap.add_argument("--model", default="c2", choices=["c2", "c1", "c4"], help="c2: the bench workload (BASELINE configs[1], d = 3, weak scaling); c4: BASELINE configs[3], the log-Gaussian Cox process at N = 2^24 in total (strong scaling); c1: Poisson-Brownian (configs[0], d = 1) -- profiling runs only")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo", "gloo-gpu"], help="gloo: CPU rehearsal of the N-rank path with the test-only oracle shard; gloo-gpu: rehearsal with real GPU shards sharing the devices there are (peer windows over hipIpc, control plane on gloo) -- not a measurement")
    ap.add_argument("--launch-timeout", type=float, default=1500.0, help="seconds the self-launched ranks may take")
    ap.add_argument("--stage-timeout", type=float, default=300.0, help="N ranks: seconds one stage of a rank (a pre-flight series, the warm-up, one timed leg) may take before the rank leaves with a non-zero status")
    ap.add_argument("--sharded", action="store_true", help="diagnostic: with --gpus 1, run the SHARDED code path at world = 1 over RCCL (its kernels, its collective) instead of the single-GPU filter")
    args = ap.parse_args()
    in_launcher = "WORLD_SIZE" in os.environ and "RANK" in os.environ
    if args.gpus > 1 and not in_launcher:
        sys.exit(launch_ranks(args, sys.argv[1:]))
    with _QuietStdout() as out:
        if int(os.environ.get("WORLD_SIZE", "1")) > 1 or args.sharded:
            run_multi(args, out.emit)
        elif args.backend in ("gloo", "gloo-gpu"):
            run_multi(args, out.emit)
        else:
            run_single(args, out.emit)


if __name__ == "__main__":
    main()
