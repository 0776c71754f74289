#!/usr/bin/env python3
"""bench.py -- BASELINE.json's headline metric on MI355X.

  python bench.py --gpus N --steps K --warmup W
  N > 1: one rank per GPU under torch.distributed.run.  Started WITHOUT the torchrun
  environment (WORLD_SIZE unset) the script starts the N ranks itself, as fresh child
  processes, before anything in this process touches the GPU, and exits with their code.
  WORLD_SIZE != --gpus is an error (never a silent 1-GPU run).

Workload (BASELINE.json configs[1]): findall of `[a-z]+\\d+` over 2^20 synthetic
1 KiB ASCII texts PER GPU (SURVEY.md 8(d) mix: 40 % full / 30 % tokens / 20 % noise
/ 10 % adversarial), already resident in HBM.  One step = one pass of the hot path
over the batch through the C ABI (mrx_findall_strided_dev): streaming scan kernel
+ CSR prefix sums + record decode into spans; outputs stay in HBM.  Texts are independent, so N GPUs each scan
their own batch (weak scaling, no data-path collective).

Prints ONE JSON line (rank 0) with the contract fields plus
  roofline     dominant kernel (k_stream_findall): algorithmic bytes per launch /
               its mean duration from HIP events on the launch stream; peak 8 TB/s
  cpu_baseline the oracle's C port (scalar, 1 thread) on a bounded sample of the
               same batch, timed on this box's host cores (rank 0, N == 1 only)
"""
from __future__ import annotations

import argparse
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PATTERN = b"[a-z]+\\d+"
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md


def _host_info():
    model, mem = "unknown", None
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                model = line.split(":", 1)[1].strip()
                break
        for line in open("/proc/meminfo"):
            if line.startswith("MemTotal"):
                mem = round(int(line.split()[1]) / (1 << 20), 1)
                break
    except OSError:
        pass
    return model, mem


def _median_time(fn, repeats: int):
    ts = []
    for _ in range(repeats):
        t0 = time.perf_counter()
        r = fn()
        ts.append(time.perf_counter() - t0)
    ts.sort()
    return ts[len(ts) // 2], r


def _reference_method_time(fn):
    """The reference's timing method (benchmarks/bench_engine.mojo:83-90, 205-240): 10 warm-up calls, the iteration count
    per sample calibrated so that a sample takes >= 10 ms, samples until their total reaches 500 ms, the MEDIAN sample's
    time per call.  Returns (seconds per call, samples, calls per sample, last result)."""
    r = None
    for _ in range(10):
        r = fn()
    iters = 1
    t0 = time.perf_counter()
    r = fn()
    cal = time.perf_counter() - t0
    if cal < 0.010:
        iters = int(0.010 // max(cal, 1e-9)) + 1
    times, total = [], 0.0
    while total < 0.5 and len(times) < 200000:
        t0 = time.perf_counter()
        for _ in range(iters):
            r = fn()
        dt = time.perf_counter() - t0
        total += dt
        times.append(dt / iters)
    times.sort()
    m = len(times)
    med = times[m // 2] if m % 2 else 0.5 * (times[m // 2 - 1] + times[m // 2])
    return med, m, iters, r


def cpu_baseline(host_rows, pattern: bytes, time_rows: int = 4096):
    """Oracle C port on a bounded sample (checker code, used here only as the reported CPU
    baseline -- never on the measured GPU path).  SURVEY.md 8(d): the same algorithm as the
    reference (scalar table walk, dfa.mojo:1996-2009; AVX2 range-compare skip scan,
    simd_ops.mojo:585-692), built -O3 -march=native on this box, one thread like the reference,
    timed after a warm-up run as the median of three."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import numpy as np
    from mrx_ref import cfast
    from mrx_ref.cfast import CDfa
    n, L = host_rows.shape
    native = bool(cfast.build_native())
    cd = CDfa(pattern, native=native)
    offsets = np.arange(0, (n + 1) * L, L, dtype=np.int64)
    data = host_rows.reshape(-1)
    # `value`: the first `time_rows` texts, timed with the reference's own method -- a call is one
    # findall pass over them; the counts of the whole sample (parity spot check, all-cores leg) come from one more pass
    m_t = min(n, time_rows)
    t_data, t_off = data[: m_t * L], offsets[: m_t + 1]
    dt, nsamples, iters, (_, _, t_total) = _reference_method_time(lambda: cd.findall_batch(t_data, t_off, want_spans=False))
    counts, _, total = cd.findall_batch(data, offsets, want_spans=False)
    model, mem_gib = _host_info()
    out = {
        "value": round(m_t * L / dt / 1e9, 4), "unit": "GB/s", "cores": 1, "kind": "port",
        "sample": "first %d texts (%d MiB) of the same batch, findall, oracle/c/mrx_oracle.c (%s), timed as the reference "
                  "times (bench_engine.mojo:83-90, 205-240): 10 warm-up calls, %d sample(s) of %d call(s) >= 10 ms until "
                  ">= 500 ms, median %.3f s per call, %d matches per call; parity and the all-cores leg on the first %d texts"
                  % (m_t, m_t * L >> 20, "-O3 -march=native" if native else "-O3 -march=x86-64-v3, no compiler on this box",
                     nsamples, iters, dt, t_total, n),
        "matches_per_s": round(t_total / dt, 1),
        "cpu_model": model, "host_mem_GiB": mem_gib,
    }
    # per text kind (the mix of SURVEY.md 8(d)): the reference's restart-per-position search is quadratic on
    # the adversarial rows, which dominate the figure above
    last = host_rows[:, L - 1]
    has_space = (host_rows == 32).any(axis=1)
    all_lower_but_last = ((host_rows[:, : L - 1] >= 97) & (host_rows[:, : L - 1] <= 122)).all(axis=1)
    alnum = (((host_rows >= 97) & (host_rows <= 122)) | ((host_rows >= 48) & (host_rows <= 57))).all(axis=1)
    alnum_sp = (((host_rows >= 97) & (host_rows <= 122)) | ((host_rows >= 48) & (host_rows <= 57)) | (host_rows == 32)).all(axis=1)
    kinds = {"adversarial": all_lower_but_last & (last == 33), "full": alnum, "tokens": has_space & alnum_sp}
    kinds["noise"] = ~(kinds["adversarial"] | kinds["full"] | kinds["tokens"])
    by_kind = {}
    for name, sel in kinds.items():
        rows = np.ascontiguousarray(host_rows[sel][:4096])
        m = rows.shape[0]
        if m == 0:
            continue
        offs = np.arange(0, (m + 1) * L, L, dtype=np.int64)
        kd, _ = _median_time(lambda: cd.findall_batch(rows.reshape(-1), offs, want_spans=False), 3)
        by_kind[name] = {"texts_timed": int(m), "share_of_sample": round(float(sel.mean()), 3),
                         "GBps": round(m * L / kd / 1e9, 4)}
    out["by_text_kind"] = by_kind
    # second leg: the same scan with the texts split over all host cores this
    # process may use; the reference itself is single threaded, so `value` stays the 1-thread figure
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    if cores > 1:
        cd.count_batch_mt(data, offsets, cores)   # warm-up (thread pool)
        dt, (counts_mt, total_mt) = _median_time(lambda: cd.count_batch_mt(data, offsets, cores), 3)
        out["all_cores"] = {"value": round(n * L / dt / 1e9, 4), "unit": "GB/s", "cores": cores,
                            "seconds": round(dt, 2), "same_counts": bool((counts_mt == counts).all())}
    return out, counts


def measured_traffic(n: int, L: int):
    """HBM bytes per launch of the scan kernel from the committed PMC passes
    (profiles/rNN_traffic.json, produced by tools/pmc_pass.sh + tools/summarize_profiles.py:
    separate rocprofv3 --pmc runs, FETCH_SIZE x2 + WRITE_SIZE as the MI355X guide prescribes).
    None unless a profile of exactly this workload exists."""
    import glob
    best = None
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_traffic.json"))):
        try:
            t = json.load(open(f))
        except Exception:
            continue
        if t.get("config") == {"texts_per_gpu": n, "text_bytes": L}:
            best = (int(t["traffic_bytes_per_launch"]), os.path.basename(f))
    return best


def launch_ranks(n: int, share: bool) -> int:
    """Start `python -m torch.distributed.run --nproc-per-node n bench.py <same args>` as a child
    process (one rank per GPU over RCCL) and return its exit code.  Called before this process has
    made any HIP call; the parent only waits."""
    import socket
    import subprocess
    import torch
    have = torch.cuda.device_count()   # counts devices without initialising the GPU
    if have < n and not share:
        print("bench.py: --gpus %d but this node has %d GPU(s) (MRX_BENCH_SHARE_GPU=1 rehearses the "
              "N-rank flow on fewer GPUs over gloo)" % (n, have), file=sys.stderr)
        return 2
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "4")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--settle", type=int, default=64, help="untimed steps before the warm-up (arena, clocks)")
    ap.add_argument("--texts", type=int, default=1 << 20, help="texts per GPU")
    ap.add_argument("--length", type=int, default=1024)
    ap.add_argument("--cpu-sample", type=int, default=1 << 15, help="texts of the batch the CPU baseline is timed on")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--strong", action="store_true",
                    help="strong scaling: --texts is the WHOLE job, split over the ranks by contiguous "
                         "index ranges (dist.shard_range); default is weak scaling (--texts per GPU)")
    ap.add_argument("--streams", type=int, default=1,
                    help="HIP streams the K timed steps are issued on.  1 (default) = strictly serial steps on "
                         "one stream: `value` and `ms_per_step` are always priced this way (round 1's method).  "
                         "The overlapped figure (steps round-robin on two streams, the decode of one step under "
                         "the scan of the next, as a caller feeding batches continuously runs) is reported as the "
                         "extra object `two_streams_overlapped`, never as `value`")
    ap.add_argument("--gather", action="store_true",
                    help="N > 1: time scan + results exchange (SURVEY.md 8(e): all-gatherv of the spans to every "
                         "rank) even with --no-extras; reported as an extra object, never as `value`")
    ap.add_argument("--no-overlap-leg", action="store_true",
                    help="skip the two-stream leg (`two_streams_overlapped`): a kernel trace of this command then holds "
                         "serial launches only, and its average scan duration is the one `roofline.kernel_ms` reports "
                         "(in the overlapped leg two scans share the device and each takes longer)")
    ap.add_argument("--no-extras", action="store_true",
                    help="N > 1: skip the extra legs (results exchange on the headline batch, config 3's per-GPU share)")
    ap.add_argument("--c3-texts", type=int, default=1 << 23, help="texts per GPU of the config-3 leg (256 B each)")
    args = ap.parse_args()

    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    share = bool(os.environ.get("MRX_BENCH_SHARE_GPU"))
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # Not under torchrun: start the ranks as fresh children.  Nothing in this process has
        # touched the GPU yet (device_count() does not initialise it on this image).
        sys.exit(launch_ranks(args.gpus, share))

    import torch
    import mojo_regex_amd as M
    from mojo_regex_amd import dist as D
    from mojo_regex_amd.workloads import make_c2_batch

    rank, local_rank, world = D.env_world()
    if world != args.gpus:
        raise SystemExit("WORLD_SIZE (%d) != --gpus (%d): refusing to report a %d-rank run as --gpus %d"
                         % (world, args.gpus, world, args.gpus))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    # MRX_BENCH_SHARE_GPU=1: rehearsal of the N>1 flow on a box with fewer GPUs than ranks (ranks
    # share the GPUs round-robin, rendezvous over gloo -- RCCL refuses two ranks on one device).
    # The driver's real runs use one GPU per rank and nccl (= RCCL).
    backend = os.environ.get("MRX_BENCH_BACKEND", "gloo" if share else "nccl")
    if not share and world > torch.cuda.device_count():
        raise SystemExit("%d ranks but %d GPU(s) (MRX_BENCH_SHARE_GPU=1 rehearses on fewer)"
                         % (world, torch.cuda.device_count()))
    dev_index = local_rank % torch.cuda.device_count() if share else local_rank
    torch.cuda.set_device(dev_index)
    dev = "cuda:%d" % dev_index
    if world > 1:
        D.init(backend)

    n, L = args.texts, args.length
    if args.strong:
        lo, hi = D.shard_range(args.texts, rank, world)
        n = hi - lo
    batch_t = make_c2_batch(n, L, seed=20260102 + rank, device=dev)
    batch = M.DeviceBatch.strided(batch_t.reshape(-1), L, length=L)
    rx = M.compile_regex(PATTERN)
    lib = M.load_library()

    # preallocated outputs: no allocation of result buffers inside the timed region.  Every step is
    # a complete findall pass into its stream's own (prefix, spans) buffers; with --streams 2 the
    # record decode of one step overlaps the scan of the next (two batches in flight, as a caller
    # feeding batches continuously would run it).
    span_cap = n * 32
    nstreams = max(1, args.streams)
    streams = [torch.cuda.Stream(device=dev) for _ in range(nstreams)]
    outs = [(torch.empty(n + 1, dtype=torch.int64, device=dev),
             torch.empty((span_cap, 2), dtype=torch.int32, device=dev)) for _ in range(nstreams)]
    prefix, spans = outs[0]
    out = outs[0]
    issued = [0]

    def step():
        # enqueue one full findall pass (scan kernel + prefix sums + decode); results and
        # the total stay on the device, nothing is read back inside the timed region
        k = issued[0] % nstreams
        issued[0] += 1
        with torch.cuda.stream(streams[k]):
            rx.findall_async(batch, outs[k])

    torch.cuda.synchronize()
    # Untimed set-up, not part of the W warm-up steps: the per-stream scratch arena settles within two
    # calls of a new batch shape, and the part needs some tens of ms of load to reach its steady
    # clocks (20 steps = 7 ms; measured 0.356 ms per step cold against 0.323 ms with 100 steps).
    for _ in range(args.settle):
        step()
    torch.cuda.synchronize()
    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    D.barrier(world, dev if backend == "nccl" else None)
    issued[0] = 0
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    D.barrier(world, dev if backend == "nccl" else None)
    elapsed = time.perf_counter() - t0
    total = int(prefix[n].item())
    if total > span_cap:
        raise SystemExit("span buffer too small (%d > %d): result incomplete, run invalid" % (total, span_cap))

    agg = D.combine(world, elapsed, {"bytes": float(n) * L * args.steps,
                                     "matches": float(total) * args.steps},
                    device=dev if backend == "nccl" else "cpu")

    # ---- N > 1, weak run: the same job at FIXED TOTAL size next to it (strong scaling) ----------
    # One 1-GPU batch (--texts texts in all) split over the ranks by contiguous index ranges; same
    # barrier + max-over-ranks timing.  Extra object `strong`, never `value`.
    strong_info = None
    if world > 1 and not args.strong:
        lo, hi = D.shard_range(args.texts, rank, world)
        ns = hi - lo
        sbatch = M.DeviceBatch.strided(batch_t[:ns].reshape(-1), L, length=L)
        sout = (prefix[:ns + 1], spans)
        for _ in range(args.warmup + 8):
            rx.findall_async(sbatch, sout)
        torch.cuda.synchronize()
        D.barrier(world, dev if backend == "nccl" else None)
        s0 = time.perf_counter()
        for _ in range(args.steps):
            rx.findall_async(sbatch, sout)
        torch.cuda.synchronize()
        D.barrier(world, dev if backend == "nccl" else None)
        sel = time.perf_counter() - s0
        sagg = D.combine(world, sel, {"bytes": float(ns) * L * args.steps},
                         device=dev if backend == "nccl" else "cpu")
        strong_info = {"value": round(sagg["bytes"] / sagg["elapsed_s"] / 1e9, 3), "unit": "GB/s",
                       "ms_per_step": round(sagg["elapsed_s"] / args.steps * 1e3, 4),
                       "total_texts": args.texts, "texts_per_gpu": ns,
                       "note": "fixed total job (one GPU's batch) split over the ranks; each rank's "
                               "share is the prefix of its weak-run batch"}
        for _ in range(2):   # put the arena back into the weak run's shape for the legs below
            step()
        torch.cuda.synchronize()

    # ---- roofline of the dominant kernel: HIP events on its own launch stream -------
    lib.mrx_timing_enable(1)
    lib.mrx_timing_reset()
    for _ in range(max(5, min(args.steps, 20))):
        step()
        torch.cuda.synchronize()
    launches = ctypes.c_int64(0)
    scan_ms = lib.mrx_timing_scan_ms(ctypes.byref(launches))
    lib.mrx_timing_enable(0)
    kernel = lib.mrx_last_kernel_name().decode()
    # the same K steps round-robin on TWO streams (decode of step i under the scan of step i + 1), next to
    # the serial headline: an extra object, never `value`
    overlap_ms = None
    if nstreams == 1 and rank == 0 and not args.no_overlap_leg:
        s2 = [streams[0], torch.cuda.Stream(device=dev)]
        o2 = [outs[0], (torch.empty(n + 1, dtype=torch.int64, device=dev),
                        torch.empty((span_cap, 2), dtype=torch.int32, device=dev))]
        def step2(i):
            with torch.cuda.stream(s2[i & 1]):
                rx.findall_async(batch, o2[i & 1])
        for i in range(8):
            step2(i)
        torch.cuda.synchronize()
        a0 = time.perf_counter()
        for i in range(args.steps):
            step2(i)
        torch.cuda.synchronize()
        overlap_ms = (time.perf_counter() - a0) / args.steps * 1e3
        del o2
    # algorithmic bytes per launch (DESIGN.md "Measurement"): every input byte once,
    # + 8 B per span written to its slot + 4 B per text for the count
    alg_bytes = float(n) * L + 8.0 * total + 4.0 * n
    achieved = alg_bytes / (scan_ms * 1e-3) / 1e9 if scan_ms > 0 else 0.0

    # ---- the other operations of the path on the same batch (rank 0's view; extra fields) -------
    def _time(fn, reps=10):
        fn()
        torch.cuda.synchronize()
        a = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - a) / reps

    other = None
    if rank == 0:
        nb = float(n) * L
        other = {"count_GBps": round(nb / _time(lambda: rx.count(batch)) / 1e9, 1),
                 "search_GBps": round(nb / _time(lambda: rx.match_next(batch)) / 1e9, 1),
                 "match_first_GBps_whole_batch": round(nb / _time(lambda: rx.match_first(batch)) / 1e9, 1),
                 "note": "same batch, device resident, per rank; count = scan without records/decode"}

    line = None
    if rank == 0:
        value = agg["bytes"] / agg["elapsed_s"] / 1e9
        line = {
            "metric": "GB/s input scanned + matches/sec, 1M x 1KiB batch, [a-z]+\\d+ DFA",
            "value": round(value, 3), "unit": "GB/s",
            "matches_per_s": round(agg["matches"] / agg["elapsed_s"], 1),
            "n_gpus": world, "steps": args.steps,
            # `warmup` = the W asked for; every untimed step before the timed region (the settle
            # phase that precedes them included) is `untimed_steps`
            "warmup": args.warmup, "untimed_steps": args.settle + args.warmup,
            "ms_per_step": round(agg["elapsed_s"] / args.steps * 1e3, 4),
            "higher_is_better": True, "scaling": "strong" if args.strong else "weak", "vs_baseline": None,
            "dtype": "u8", "data": "synthetic",
            "config": {"workload": "findall [a-z]+\\d+ over %d x %d B ASCII texts per GPU "
                                   "(40/30/20/10 full/tokens/noise/adversarial)" % (n, L),
                       "texts_per_gpu": n, "text_bytes": L, "pattern": PATTERN.decode(),
                       "op": "findall", "matches_per_batch": int(total),
                       "parallelism": "texts sharded, %d rank(s), no data-path collective" % world,
                       "streams": nstreams, "settle_steps": args.settle},
            "hbm_frac_of_peak_whole_step": round(value / world / HBM_PEAK_GBS, 4),
            "roofline": {"bound": "hbm", "kernel": kernel, "achieved": round(achieved, 2),
                         "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4),
                         "traffic": (measured_traffic(n, L) or (None, None))[0],
                         "traffic_source": (measured_traffic(n, L) or (None, None))[1],
                         "kernel_ms": round(scan_ms, 4), "launches_timed": int(launches.value),
                         # kernel_ms: each launch alone on the device (a synchronisation between the timed launches),
                         # which is what `achieved` is priced with.  In the timed region of a --streams 2 run the
                         # scans of the two streams run side by side, so a kernel trace of this command shows about
                         # twice this duration per launch (two kernels share the device); profiles/
                         # rNN_kernel_stats_streams1.csv is the trace of --streams 1, where the two agree.
                         "kernel_timing": "isolated launches",
                         # real HBM traffic rate of the kernel next to what a plain float4 copy reaches
                         # on this part (6.29 TB/s measured, MI355X_MICROARCH.md) -- informational
                         "traffic_GBps": (round(measured_traffic(n, L)[0] / (scan_ms * 1e-3) / 1e9, 1)
                                          if measured_traffic(n, L) and scan_ms > 0 else None),
                         "copy_GBps_measured_on_part": 6290.0,
                         "algorithmic_bytes_per_launch": int(alg_bytes)},
        }
        if overlap_ms is not None:
            line["two_streams_overlapped"] = {
                "ms_per_step": round(overlap_ms, 4),
                "value": round(float(n) * L / (overlap_ms * 1e-3) / 1e9, 3), "unit": "GB/s",
                "hbm_frac_of_peak_whole_step": round(float(n) * L / (overlap_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                "note": "rank 0, the same K steps issued round-robin on two streams with separate outputs; "
                        "round 2 printed this as `value`, round 1 and round 3 print the serial figure"}
        if other is not None:
            line["other_ops"] = other
        if strong_info is not None:
            line["strong"] = strong_info

    # ---- scan + results exchange (results stay sharded in the headline) -----------------------------
    # the one optional exchange step, "results only".  With the nccl backend it runs behind the
    # C ABI (include/mrx_comm.h: RCCL called by the library, all-gather of the sizes, then the spans into
    # prefix-sum offsets; padded form, nothing is read back inside the timed loop); gloo (shared-GPU rehearsal)
    # takes the torch.distributed path.  On by default for N > 1 (MRX_BENCH_EXTRAS=0 or --no-extras turns the
    # extra legs off); extra objects, never `value`.  A leg that raises is reported as {"error": ...} and the
    # headline line is still printed (the legs are symmetric: every rank raises or none does).
    gather_info = None
    config3_info = None
    extras = world > 1 and not args.no_extras and os.environ.get("MRX_BENCH_EXTRAS", "1") != "0"

    import threading as _threading
    progress = {"leg": "none", "call": "none"}     # what the watchdog reports if the legs hang
    _print_lock = _threading.Lock()
    _printed = [False]

    def _claim_print():
        with _print_lock:
            if _printed[0]:
                return False
            _printed[0] = True
            return True

    def exchange_leg(rx_, batch_, out_, n_, comm):
        """Times K steps of findall and of findall + exchange on this batch; returns the extra object."""
        gdev = dev if backend == "nccl" else None
        gsteps = max(3, min(args.steps, 10))
        pre_, sp_ = out_
        progress["call"] = "first findall of the leg"
        rx_.findall_async(batch_, out_)
        tot = torch.tensor([int(pre_[n_].item())], dtype=torch.int64, device=dev if backend == "nccl" else "cpu")
        import torch.distributed as tdist
        progress["call"] = "torch.distributed all_reduce of the span totals (%s)" % backend
        tdist.all_reduce(tot, op=tdist.ReduceOp.MAX)
        cap = int(tot.item()) + 1024          # padded form: slots every rank ships (the largest rank's spans)
        n_all = torch.tensor([n_], dtype=torch.int64, device=tot.device)
        tdist.all_reduce(n_all, op=tdist.ReduceOp.SUM)
        n_glob = int(n_all.item())
        if cap > sp_.shape[0]:
            raise RuntimeError("span buffer smaller than the exchange capacity")
        gout = None
        if comm is not None:
            progress["call"] = "mrx_comm_reserve (staging of the padded exchange)"
            comm.reserve_spans(n_glob, cap)
            gout = (torch.empty(n_glob + 1, dtype=torch.int64, device=dev),
                    torch.empty((world * cap, 2), dtype=torch.int32, device=dev),
                    torch.zeros(1, dtype=torch.int32, device=dev))

        def scan_only():
            rx_.findall_async(batch_, out_)

        def scan_gather():
            rx_.findall_async(batch_, out_)
            if comm is not None:
                return comm.gather_spans(pre_, sp_, n_global=n_glob, cap_spans_per_rank=cap, out=gout)
            return D.gather_spans(world, pre_, sp_, int(pre_[n_].item()))

        res = {}
        how = "mrx_allgatherv_spans over RCCL" if comm is not None else "torch.distributed gather (%s)" % backend
        for name, fn in (("scan_only", scan_only), ("scan_plus_gather", scan_gather)):
            what = "findall" if name == "scan_only" else "findall + " + how
            for k_ in range(2):
                progress["call"] = "%s, warm-up step %d" % (what, k_)
                r = fn()
            progress["call"] = "%s, device synchronisation after the warm-up" % what
            torch.cuda.synchronize()
            progress["call"] = "%s, barrier before the timed steps" % what
            D.barrier(world, gdev)
            g0 = time.perf_counter()
            for k_ in range(gsteps):
                progress["call"] = "%s, timed step %d of %d" % (what, k_, gsteps)
                r = fn()
            progress["call"] = "%s, device synchronisation after the timed steps" % what
            torch.cuda.synchronize()
            progress["call"] = "%s, barrier after the timed steps" % what
            D.barrier(world, gdev)
            gel = time.perf_counter() - g0
            ga = D.combine(world, gel, {"bytes": float(batch_.nbytes_text) * gsteps}, device=dev if backend == "nccl" else "cpu")
            res[name] = {"ms_per_step": round(ga["elapsed_s"] / gsteps * 1e3, 4),
                         "GBps_whole_job": round(ga["bytes"] / ga["elapsed_s"] / 1e9, 3)}
        if comm is not None:
            status = int(r[2].item())
            total_spans = int(r[0][n_glob].item())
            res["exchange"] = {"form": "padded, no host read-back (mrx_allgatherv_spans over RCCL)", "status": status,
                               "span_slots_per_rank": cap, "global_texts": n_glob, "global_spans": total_spans,
                               "gathered_span_bytes_per_rank": total_spans * 8}
        else:
            res["exchange"] = {"form": "torch.distributed (%s)" % backend, "global_texts": int(r[0].shape[0]) - 1,
                               "global_spans": int(r[1].shape[0]), "gathered_span_bytes_per_rank": int(r[1].shape[0]) * 8}
        res["steps"] = gsteps
        res["backend"] = backend
        return res

    watchdog = None
    if extras or (args.gather and world > 1):
        # The exchange legs are the one part of this file that no multi-GPU box has run yet.  If they hang (a rank
        # that died, an RCCL rendezvous that never completes), the headline measured above must not be lost with
        # them: after MRX_BENCH_EXTRAS_TIMEOUT seconds (default 150) rank 0 prints the line without the legs and
        # every rank leaves.
        import threading
        limit = float(os.environ.get("MRX_BENCH_EXTRAS_TIMEOUT", "150"))

        def _bail():
            # exactly one line, whoever gets there first (this timer or the main thread below); the exit status says
            # that the legs hung -- the diagnosis is in the error object: which leg, which call was in flight
            if rank == 0 and line is not None and _claim_print():
                where = "leg %r, in flight: %s" % (progress["leg"], progress["call"])
                line[progress["leg"] if progress["leg"] in ("scan_plus_gather", "config3") else "scan_plus_gather"] = {
                    "error": "the extra legs did not finish within %.0f s (watchdog); %s; headline unaffected" % (limit, where)}
                print(json.dumps(line), flush=True)
            os._exit(3)
        # (the first leg is named before the timer starts: a timer that fires at once must not report leg 'none')
        progress["leg"], progress["call"] = "scan_plus_gather", "starting"
        watchdog = threading.Timer(limit, _bail)
        watchdog.daemon = True
        watchdog.start()
        comm = None
        try:
            progress["leg"] = "scan_plus_gather"
            if backend == "nccl":
                progress["call"] = "Comm.create: mrx_comm_init (ncclCommInitRank, %d ranks)" % world
                comm = D.Comm.create(world, rank)
            batch.nbytes_text = n * L
            gather_info = exchange_leg(rx, batch, out, n, comm)
            gather_info["workload"] = "config 2 (the headline batch)"
        except Exception as e:   # noqa: BLE001 -- reported, the headline must still be printed
            gather_info = {"error": "%s: %s" % (type(e).__name__, str(e)[:300])}
        # config 3 in its defining form: `\\d+` findall over 64M x 256 B sharded over 8 GPUs = 8M x 256 B per GPU
        # (BASELINE.json configs[2]); the per-GPU share is fixed, so this is a weak-scaling figure as well
        if extras:
            try:
                from mojo_regex_amd.workloads import make_digits_batch
                progress["leg"], progress["call"] = "config3", "building the batch"
                n3, L3 = args.c3_texts, 256
                b3_t = make_digits_batch(n3, L3, seed=20260103 + rank, device=dev)
                b3 = M.DeviceBatch.strided(b3_t.reshape(-1), L3, length=L3)
                b3.nbytes_text = n3 * L3
                rx3 = M.compile_regex(b"\\d+")
                out3 = (torch.empty(n3 + 1, dtype=torch.int64, device=dev),
                        torch.empty((n3 * 6, 2), dtype=torch.int32, device=dev))
                config3_info = exchange_leg(rx3, b3, out3, n3, comm)
                config3_info["workload"] = "findall \\d+ over %d x %d B texts per GPU (config 3's per-GPU share)" % (n3, L3)
                del b3, b3_t, out3
            except Exception as e:   # noqa: BLE001
                config3_info = {"error": "%s: %s" % (type(e).__name__, str(e)[:300])}
        if comm is not None:
            progress["call"] = "Comm.close: mrx_comm_free"
            comm.close()
        watchdog.cancel()


    if rank == 0:
        if gather_info is not None:
            line["scan_plus_gather"] = gather_info
        if config3_info is not None:
            line["config3"] = config3_info
        if world == 1 and not args.no_cpu_baseline:
            m = min(args.cpu_sample, n)
            host = batch_t[:m].cpu().numpy()
            cb, counts = cpu_baseline(host, PATTERN)
            # the sample doubles as a parity spot check of the measured path: counts, CSR offsets AND every span of
            # the sample's texts against the oracle's (the oracle writes its spans where the device's offsets put them)
            pre = prefix[:m + 1].cpu().numpy()
            gpu_counts = pre[1:] - pre[:-1]
            from mrx_ref.cfast import CDfa
            import numpy as np
            ocnt, ospans, _ = CDfa(PATTERN).findall_at_mt(host.reshape(-1), np.arange(0, (m + 1) * L, L, dtype=np.int64), pre,
                                                          max(1, min(64, len(os.sched_getaffinity(0)))))
            same_counts = bool((gpu_counts == counts).all()) and bool((ocnt == counts).all())
            cb["parity_on_sample"] = same_counts and bool(np.array_equal(spans[:int(pre[-1])].cpu().numpy(), ospans))
            cb["parity_checks"] = "counts, offsets and all %d spans of the sample's texts" % int(pre[-1])
            # match_first in SURVEY.md 8(d)'s units: its algorithmic bytes are data dependent -- the sum over the texts of
            # min(len, bytes consumed before the dead transition + 1) -- so the fraction comes from the oracle (same sample)
            frac = CDfa(PATTERN).match_first_bytes(host.reshape(-1), np.arange(0, (m + 1) * L, L, dtype=np.int64)) / float(m * L)
            if other is not None:
                line["other_ops"]["match_first_alg_GBps"] = round(frac * other["match_first_GBps_whole_batch"], 1)
                line["other_ops"]["match_first_algorithmic_fraction"] = round(frac, 4)
                line["other_ops"]["note"] += ("; match_first_alg_GBps = the whole-batch rate x the fraction of the bytes the reference's "
                                              "match_first examines (sum of min(len, consumed + 1), oracle, the cpu_baseline sample)")
            line["cpu_baseline"] = cb
        if _claim_print():
            print(json.dumps(line), flush=True)

    if world > 1:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
