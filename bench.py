#!/usr/bin/env python3
"""bench.py -- throughput of the batched M17 receive chain on N MI355X.

A "step" is one pass of the hot path (m17gpu_rx_blocks) over one batch of synthetic IQ already
resident in HBM: C channels x NBLK 1920-sample blocks per GPU.

Default workload = the configuration BASELINE.json's metric is quoted on ("channels decoded in real
time ... at 1/2/4/8 MI355X", configs[3]/[4]): FULL chain (discriminator, polyphase RRC timing
recovery, sync correlator/framer, demap, de-randomise/de-interleave/de-puncture, soft Viterbi,
Golay, LICH/LSF bookkeeping) at 16,384 channels per GPU, SIXTEEN blocks (640 ms of signal) per step
since round 5 -- the launch size at which the library's wave-per-channel FIR stage applies (whole
sixteen-block tiles); rounds 1-4 ran twelve, and the line still carries that
step as `step_12_blocks`.  The FIR-stage figures ride along as nested objects -- `fir_stage`
(configs[1]: 1,024 channels x 50 blocks), `fir_stage_16384` (the headline's batch and launch size),
`fir_stage_16384x12` and `fir_stage_16384x48` -- so that north_star's 40 %-of-HBM target on that
stage stays visible in the same line, each with its launch size and buffering latency in `workload`.

`python bench.py --gpus N` starts the N ranks itself (one process per GPU, RCCL), unless it is
already running as a rank of an external launcher (WORLD_SIZE set, e.g. torch.distributed.run).
The parent process never touches the GPU.

Rank 0 prints ONE JSON line (contract in the task statement) carrying `roofline` (live HIP-event
kernel durations on the launch stream against algorithmic bytes), `cpu_baseline` (the CPU oracle
timed on this host on a bounded sample, N=1 only), `fir_stage` (N=1 only) and, for N>1, the
separately timed RCCL fan-out legs `fanout_ms` / `gather_ms` (compute-only `value` and
the with-fan-out figure are both reported).
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0                 # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
BYTES_FRONT = 7680 + 768              # IQ in + symbols out per channel-block
BYTES_FULL = 7680 + 64                # IQ in + 64-byte record out
KNAMES = ["k_frontend", "k_sync_frame", "k_worklist+k_decode", "k_bookkeeping"]
EXIT_FANOUT_ABANDONED = 3             # the watchdog ended hung transfer legs: the line was printed WITHOUT them (a crash is 1, a signal > 128)


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--channels", type=int, default=None, help="channels per GPU (default 16384 full / 1024 frontend)")
    ap.add_argument("--blocks", type=int, default=None, help="1920-sample blocks per channel per step (default 16 full / 50 frontend)")
    ap.add_argument("--workload", choices=["frontend", "full"], default="full")
    ap.add_argument("--ebn0", type=float, default=200.0, help="AWGN level of the synthetic IQ (>=100: none)")
    ap.add_argument("--noise-cutoff", type=float, default=0.0,
                    help="one-sided cutoff (Hz) of the channel filter on the AWGN of the device generator (0 = white over 48 kHz); "
                         "--ebn0 8 --noise-cutoff 6250 is the workload of the nested `noisy` leg")
    ap.add_argument("--unique", type=int, default=256, help="host generator: distinct generated channels (tiled to --channels)")
    ap.add_argument("--gen", choices=["gpu", "host"], default="gpu",
                    help="signal source: m17gpu_gen_batch on the device (every channel distinct) or the host generator")
    ap.add_argument("--signal-gb", type=float, default=40.0, help="distinct signal kept in HBM per GPU before the stream wraps")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-fir-stage", action="store_true", help="skip the nested FIR-stage measurements (configs[1] and 16,384 channels)")
    ap.add_argument("--no-step12", action="store_true", help="skip the nested full-chain step of twelve blocks (the headline of rounds 1-4)")
    ap.add_argument("--no-noisy", action="store_true", help="skip the nested measurement on the AWGN workload of configs[3]")
    ap.add_argument("--no-fanout", action="store_true", help="N>1: skip the separately timed RCCL scatter/gather legs")
    ap.add_argument("--fanout", choices=["torch", "capi"], default=None,
                    help="transport of the fan-out legs: m17_sdr_amd/shard.py over torch.distributed (default for N>1) or the C-ABI "
                         "entries of include/m17gpu.h on an ncclComm_t; given explicitly the legs also run at N=1 (degenerate: "
                         "device copies, every call made)")
    ap.add_argument("--no-syms", action="store_true", help="front end: do not write the symbol stream")
    ap.add_argument("--no-settle", action="store_true",
                    help="skip the untimed clock-settling calls in front of the warm-up steps (the timed region then sees the DVFS ramp)")
    ap.add_argument("--option", action="append", default=[], metavar="NAME=VALUE",
                    help="m17gpu_set_option on the receiver (A/B of bit-identical kernel variants; recorded in config)")
    args = ap.parse_args(argv)
    if args.channels is None:
        args.channels = 16384 if args.workload == "full" else 1024
    if args.blocks is None:
        args.blocks = 16 if args.workload == "full" else 50
    return args


# ------------------------------------------------------------------------------------------------
# parent: start one process per GPU (never touches HIP itself)
# ------------------------------------------------------------------------------------------------
def launch_ranks(args, timeout_s=1200.0):
    """Start one fresh child per rank (never a re-exec: this process has not touched the GPU and never does), poll
    them all, and on the first non-zero exit -- or after timeout_s -- terminate the others instead of leaving them in
    a rendezvous or a barrier until the process-group timeout."""
    n = args.gpus
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]            # released here: a rank that finds it taken fails fast and all are stopped
    import tempfile
    procs, out0 = [], tempfile.TemporaryFile(mode="w+")
    for r in range(n):
        env = dict(os.environ)
        env.update({"RANK": str(r), "LOCAL_RANK": str(r), "WORLD_SIZE": str(n), "MASTER_ADDR": "127.0.0.1",
                    "MASTER_PORT": str(port), "HSA_ENABLE_IPC_MODE_LEGACY": os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0")})
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=out0 if r == 0 else subprocess.DEVNULL))
    t_end = time.monotonic() + timeout_s
    bad = None
    while True:
        rcs = [p.poll() for p in procs]
        bad = next(((r, rc) for r, rc in enumerate(rcs) if rc not in (None, 0)), None)
        if bad or all(rc is not None for rc in rcs) or time.monotonic() > t_end:
            break
        time.sleep(0.2)
    if bad or any(p.poll() is None for p in procs):
        for p in procs:
            if p.poll() is None:
                p.terminate()
        for p in procs:
            try:
                p.wait(timeout=10)
            except subprocess.TimeoutExpired:
                p.kill()
        sys.stderr.write(f"bench.py: rank {bad[0]} exited with {bad[1]}; the other ranks were stopped\n" if bad
                         else f"bench.py: ranks still running after {timeout_s:.0f} s were stopped\n")
    out0.seek(0)
    text = out0.read()
    line = [ln for ln in text.splitlines() if ln.startswith("{")]
    rcs = [p.returncode for p in procs]
    if bad or any(rcs):
        # the line, if rank 0 got as far as printing it (the watchdog prints it before it ends the ranks), survives whatever
        # the ranks' exit codes are; the code says what happened: EXIT_FANOUT_ABANDONED when every failing rank ended
        # that way (the compute-only measurement is whole, the transfer legs are not), else the first failure's
        if line:
            print(line[-1], flush=True)
        else:
            sys.stdout.write(text)
        failing = [rc for rc in rcs if rc not in (0, None)]
        if failing and all(rc == EXIT_FANOUT_ABANDONED for rc in failing):
            return EXIT_FANOUT_ABANDONED
        return (bad[1] if bad else 1) or 1
    if not line:
        sys.stderr.write("bench.py: rank 0 printed no result line\n")
        return 1
    print(line[-1], flush=True)
    return 0


# ------------------------------------------------------------------------------------------------
# one rank
# ------------------------------------------------------------------------------------------------
def make_input(args, rank, torch, rx, C, nblk, T):
    """Synthetic IQ for this rank's channel shard, resident in HBM before the timed region.
    Default: ONE continuous signal per channel made on the device by m17gpu_gen_batch (SURVEY 8f-1,
    every channel distinct, seeded by global channel id: carrier, preambles, link setup frame, 40
    stream frames, EOT, repeating) and cut into per-step slabs -- each step receives the next
    nblk x 40 ms of every channel, as a live receiver would.  At most --signal-gb of distinct signal is
    kept; longer runs wrap around (one frame-phase break per pass, like a transmission cut short).
    --gen host: the host generator, tiled to C channels, the same slab every step.
    Returns (list of per-step device tensors, host copy of <= 256 channels of the first timed slab)."""
    import m17_sdr_amd as m
    if args.gen == "gpu":
        Tg = max(1, min(T, int(args.signal_gb * 1e9 // (C * nblk * 7680))))
        # the generator writes one stream per channel [C][nblk*Tg]; the C-ABI takes [C][nblk] per step, so the
        # stream is re-laid out once as [step][C][nblk] (peak 2 x the signal; 288 GB of HBM make that a non-issue)
        big = rx.gen_batch(nblk * Tg, n_stream_frames=40, ebn0_db=args.ebn0, first_channel=rank * C,
                           noise_cutoff_hz=args.noise_cutoff)["iq"]
        torch.cuda.synchronize(rx.device)
        slabs = torch.empty((Tg, C, nblk, 1920, 2), dtype=torch.int16, device=big.device)
        slabs.copy_(big.view(C, Tg, nblk, 1920, 2).permute(1, 0, 2, 3, 4))
        del big
        torch.cuda.synchronize(rx.device)
        torch.cuda.empty_cache()
        steps = [slabs[k % Tg] for k in range(T)]
        w = min(args.warmup, Tg - 1)
        nhost = min(C, max(256, 4 * host_cores()[0]))          # cpu_baseline: >= 4 channels per usable host thread
        return steps, {"iq": steps[w][:nhost].cpu().numpy()}, Tg
    uniq = min(args.unique, C)
    nthreads = max(1, min(16, (os.cpu_count() or 8) // max(1, args.gpus)))
    sig = m.generate_batch(uniq, nblk, n_stream_frames=40, ebn0_db=args.ebn0,
                           first_channel=rank * C, nthreads=nthreads)
    host = torch.from_numpy(sig["iq"])
    dev = torch.empty((C, nblk, 1920, 2), dtype=torch.int16, device=f"cuda:{rx.device}")
    for c0 in range(0, C, uniq):
        n = min(uniq, C - c0)
        dev[c0:c0 + n].copy_(host[:n], non_blocking=False)
    return [dev] * T, sig, 1


def _time_oracle(oracle, iq, mode, threads, budget_s):
    """Passes of the oracle over `iq` on `threads` OpenMP threads for about budget_s seconds: (symbols, seconds, passes)."""
    nch = iq.shape[0]
    ch = oracle.Channels(nch)
    ch.rx_blocks(iq, mode=mode, want_syms=False, nthreads=threads)      # warm (tables, page faults, thread pool)
    reps, t_used = 0, 0.0
    t0 = time.perf_counter()
    while t_used < budget_s and reps < 4000:
        ch.rx_blocks(iq, mode=mode, want_syms=False, nthreads=threads)
        reps += 1
        t_used = time.perf_counter() - t0
    return nch * iq.shape[1] * 192 * reps, t_used, reps


def cgroup_cpu_quota():
    """CPUs' worth of time the job's cgroup may use (cpu.max of cgroup v2 / cfs quota of v1), or None when unlimited."""
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        return None if q == "max" else round(int(q) / int(per), 2)
    except Exception:                                            # noqa: BLE001
        pass
    try:
        q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
        per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        return None if q <= 0 else round(q / per, 2)
    except Exception:                                            # noqa: BLE001
        return None


def cpu_baseline(mode, sig):
    """The CPU oracle (port of the reference path, m17_dsp.cpp:461-476 call tree) on a bounded sample of the same
    workload, same run, on the host cores this job can really use.  sched_getaffinity may grant every hardware thread
    of the box while the job's cgroup holds it to a share of them (the GPU pool: 256 threads visible, a 16-CPU share per
    GPU): the sample is therefore timed at 16, 32, 64, 128 threads and at every thread the affinity mask grants;
    `value` is the BEST of them, `cores` the thread count that reached it, `by_threads` all of them -- so that the
    line understates the host neither by using too few threads nor by oversubscribing a quota."""
    from tests import oracle
    avail, phys, smt = host_cores()
    nch = min(sig["iq"].shape[0], max(256, 4 * avail))
    iq = np.ascontiguousarray(sig["iq"][:nch])
    counts = sorted({t for t in (16, 32, 64, 128, avail) if t <= avail} | {min(avail, 16)})
    table, best = {}, None
    for t in counts:
        syms, t_used, reps = _time_oracle(oracle, iq, mode, t, 0.7)
        v = syms / t_used / 1e6
        table[str(t)] = round(v, 3)
        if best is None or v > best[0]:
            best = (v, t, t_used, reps)
    ch1 = oracle.Channels(1)                        # single-thread figure, to set beside SURVEY's 47.8 us/block
    best1 = None                                    # the fastest of eight passes: behind the 256-thread leg the first ones can
    for _ in range(8):                              # find the job's CPU quota spent (seen once: 749 us/block instead of 19)
        t1 = time.perf_counter()
        ch1.rx_blocks(iq[:1], mode=mode, want_syms=False, nthreads=1)
        dt = time.perf_counter() - t1
        best1 = dt if best1 is None or dt < best1 else best1
    us_blk = best1 / iq.shape[1] * 1e6
    v, cores, t_used, reps = best
    return {"value": round(v, 3), "unit": "Msym/s", "cores": cores, "kind": "port",
            "host": {"threads_usable_by_this_job": avail, "physical_cores": phys, "hardware_threads": smt,
                     "cgroup_cpu_quota": cgroup_cpu_quota(),
                     "note": "cores = OpenMP threads of the best run; threads_usable = the affinity mask, cgroup_cpu_quota = "
                             "CPUs' worth of time the job may actually use (null: unlimited); with SMT two threads share a core"},
            "by_threads": table,
            "sample": f"{nch} channels x {iq.shape[1]} blocks x {reps} passes of the same synthetic IQ, "
                      f"{'full chain' if mode == 1 else 'front end'}, OpenMP over channels ({t_used:.2f} s wall at {cores} threads; "
                      f"{len(counts)} thread counts, ~0.7 s each); 1 thread: {us_blk:.1f} us/block",
            "realtime_channels": int(v * 1e6 / 4800)}


def settle(torch, device, rx, slabs, mode, out, seconds=0.4):
    """Untimed conditioning before the W warm-up steps: the same step, back to back over whole passes of the resident
    stream, for about `seconds`.  The shader clock of an MI355X follows the load with a lag (scripts/exp_clock.py,
    profiles/r04_clock_under_load.txt: 1.8 GHz in the first calls behind the signal generator's kernels, 2.25-2.37 GHz
    after 200 back-to-back calls), so a region timed a few milliseconds after the generator would measure the ramp,
    not the receiver.  Nothing of this is timed; the W warm-up steps and the K timed steps follow unchanged.  Returns the
    number of calls made (a multiple of the number of distinct slabs: the stream stays continuous)."""
    n = len(slabs)
    t_end = time.perf_counter() + seconds
    calls = 0
    while time.perf_counter() < t_end and calls < 4000:
        for k in range(n):
            rx.rx_blocks(slabs[k], mode, out)
        torch.cuda.synchronize(device)
        calls += n
    return calls


def load_traffic(key):
    """HBM-side bytes per launch from the PMC passes kept under profiles/ (FETCH_SIZE doubled, WRITE_SIZE as is:
    MI355X_MICROARCH.md, HBM section).  Measured offline with rocprofv3 --pmc (it cannot run inside this process);
    null when no pass exists for this exact workload key."""
    path = os.path.join(ROOT, "profiles", "traffic.json")
    try:
        tj = json.load(open(path))
        ent = tj.get(key)
        if isinstance(ent, dict):
            return ent.get("bytes"), ent.get("source")
        return ent, "profiles/traffic.json"
    except Exception:
        return None, None


def roofline_obj(kms, ncalls, mode, cb_per_launch, key, path=None, wall_ms=None):
    """path = Receiver.last_path() of the timed calls: which kernels ran is what the library says it ran, not a guess from
    the durations.  wall_ms = the step by the wall clock of the same launches: `frac_wall` is what a caller gets,
    `frac` what the kernels take by HIP events between them."""
    per_unit = BYTES_FRONT if mode == 0 else BYTES_FULL
    kms = list(kms)
    names = list(KNAMES)
    fir = (path or {}).get("fir", 1)
    if fir in (4, 5):
        # one kernel for front end, timing loop and framer of a channel: k_rx_chan6 (a wave per channel) or, up to 1,024
        # channels, k_sync_frame_duo<1> (three waves per channel); there is no front-end launch, the first interval is
        # the gap between two events
        kms[1] += kms[0]
        kms[0] = 0.0
        names[1] = "k_rx_chan6" if fir == 4 else "k_sync_frame_duo<1>"
    if path:
        names[3] = {1: "k_book_chan", 2: "k_book_lanes"}.get(path.get("book"), names[3])
        if path.get("plain_slots"):
            names[2] = "k_worklist+k_decode_lists_p"
    used = [i for i in range(4) if kms[i] > 0.002]
    t_path_ms = sum(kms[i] for i in used)
    dom = max(used, key=lambda i: kms[i]) if used else 0
    achieved = per_unit * cb_per_launch / (t_path_ms * 1e-3) / 1e9 if t_path_ms > 0 else 0.0
    traffic, tsrc = load_traffic(key)
    frac_wall = round(per_unit * cb_per_launch / (wall_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5) if wall_ms else None
    return {"bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBS, 5), "frac_wall": frac_wall, "wall_ms": round(wall_ms, 4) if wall_ms else None,
            "traffic": traffic, "traffic_source": tsrc, "path": path,
            "kernel": "+".join(names[i] for i in used), "dominant": names[dom],
            "algorithmic_bytes_per_channel_block": per_unit, "channel_blocks_per_launch": cb_per_launch,
            "avg_ms": {names[i]: round(kms[i], 4) for i in used}, "kernel_sum_ms": round(t_path_ms, 4),
            "calls_timed": ncalls}


def fir_stage(args, torch, device, C=1024, nblk=50, steps=30, warm=3):
    """The FIR stage alone in the same run -- limiter / discriminator + polyphase RRC timing recovery + sync correlator,
    symbols out (mode 0), one continuous stream per channel -- at BASELINE configs[1]'s own size (1,024 channels x 50
    blocks) and, as `fir_stage_16384`, at the size north_star attaches its 40 %-of-HBM target to (>= 10,000 channels:
    16,384 x 12, the headline's per-GPU batch)."""
    import m17_sdr_amd as m
    rx = m.Receiver(C, nblk, device=device)
    for kv in args.option:
        name, value = kv.split("=")
        rx.set_option(name, int(value))
    T = steps + warm
    big = rx.gen_batch(nblk * T, n_stream_frames=40, ebn0_db=200.0)["iq"]
    torch.cuda.synchronize(device)
    slabs = torch.empty((T, C, nblk, 1920, 2), dtype=torch.int16, device=big.device)
    slabs.copy_(big.view(C, T, nblk, 1920, 2).permute(1, 0, 2, 3, 4))
    del big
    out = rx.alloc_outputs(nblk, want_syms=True)
    settled = settle(torch, device, rx, slabs, 0, out)
    for k in range(warm):
        rx.rx_blocks(slabs[k], 0, out)
    torch.cuda.synchronize(device)
    rx.set_profiling(True)
    t0 = time.perf_counter()
    for k in range(steps):
        rx.rx_blocks(slabs[warm + k], 0, out)
    torch.cuda.synchronize(device)
    dt = time.perf_counter() - t0
    rx.set_profiling(False)
    kms, ncalls = rx.kernel_ms()
    path = rx.last_path()
    rx.close()
    del slabs
    torch.cuda.empty_cache()
    ro = roofline_obj(kms, ncalls, 0, C * nblk, f"frontend:{C}x{nblk}", path, dt / steps * 1e3)
    what = "BASELINE configs[1]" if (C, nblk) == (1024, 50) else "the FIR stage at the headline's per-GPU batch"
    return {"workload": f"{what}: {C:,} channels x {nblk} blocks per launch ({nblk * 40} ms of signal buffered per call), "
                        "RRC FIR + timing recovery + sync correlator only",
            "blocks_per_launch": nblk,
            "ms": round(dt / steps * 1e3, 4), "value": round(C * nblk * 192 * steps / dt / 1e6, 3), "unit": "Msym/s",
            "frac": ro["frac"], "frac_wall": ro["frac_wall"], "achieved": ro["achieved"], "traffic": ro["traffic"], "avg_ms": ro["avg_ms"],
            "kernel_sum_ms": ro["kernel_sum_ms"], "path": path, "algorithmic_bytes_per_channel_block": BYTES_FRONT,
            "channel_blocks_per_launch": C * nblk, "steps": steps, "warmup": warm, "settle_calls": settled, "target_frac": 0.40}


def noisy_leg(args, torch, device, C, nblk, ebn0=8.0):
    """The same step on the workload BASELINE configs[3] names: band-limited AWGN (12.5 kHz channel noise, Eb/N0 8 dB),
    so that the hunt path of the framer and a decoder load that depends on lock state are on record next to the
    noiseless headline.  One continuous stream per channel, cut into steps like the headline."""
    import m17_sdr_amd as m
    steps, warm = 20, 3
    rx = m.Receiver(C, nblk, device=device)
    for kv in args.option:
        name, value = kv.split("=")
        rx.set_option(name, int(value))
    T = steps + warm
    big = rx.gen_batch(nblk * T, n_stream_frames=40, ebn0_db=ebn0, noise_cutoff_hz=6250.0)["iq"]
    torch.cuda.synchronize(device)
    slabs = torch.empty((T, C, nblk, 1920, 2), dtype=torch.int16, device=big.device)
    slabs.copy_(big.view(C, T, nblk, 1920, 2).permute(1, 0, 2, 3, 4))
    del big
    out = rx.alloc_outputs(nblk)
    settled = settle(torch, device, rx, slabs, 1, out)
    for k in range(warm):
        rx.rx_blocks(slabs[k], 1, out)
    torch.cuda.synchronize(device)
    rx.set_profiling(True)
    t0 = time.perf_counter()
    for k in range(steps):
        rx.rx_blocks(slabs[warm + k], 1, out)
    torch.cuda.synchronize(device)
    dt = time.perf_counter() - t0
    rx.set_profiling(False)
    kms, ncalls = rx.kernel_ms()
    path = rx.last_path()
    locked = int(rx.lock().sum())
    rx.close()
    del slabs
    torch.cuda.empty_cache()
    ro = roofline_obj(kms, ncalls, 1, C * nblk, f"full-noisy:{C}x{nblk}", path, dt / steps * 1e3)
    return {"workload": f"full chain, {C:,} channels x {nblk} blocks per step, band-limited AWGN at Eb/N0 {ebn0:g} dB (BASELINE configs[3])",
            "ebn0_db": ebn0, "ms": round(dt / steps * 1e3, 4), "value": round(C * nblk * 192 * steps / dt / 1e6, 3),
            "unit": "Msym/s", "frac": ro["frac"], "frac_wall": ro["frac_wall"], "achieved": ro["achieved"], "traffic": ro["traffic"], "avg_ms": ro["avg_ms"],
            "kernel_sum_ms": ro["kernel_sum_ms"], "path": path, "steps": steps, "warmup": warm, "settle_calls": settled,
            "channels_locked_at_end": locked}


def step12_leg(args, torch, device, C, nblk=12):
    """The headline step of rounds 1-4 -- full chain, 16,384 channels x TWELVE blocks, noiseless -- kept on the line so that
    the rounds stay comparable.  (Twelve blocks are no whole sixteen-block tile: since the end of round 5 the wave-per-channel
    stage packs the rows of a workgroup's four channels into three tiles here; until then front end + timing kernel ran as
    two launches.)"""
    import m17_sdr_amd as m
    steps, warm = 20, 3
    rx = m.Receiver(C, nblk, device=device)
    for kv in args.option:
        name, value = kv.split("=")
        rx.set_option(name, int(value))
    T = steps + warm
    big = rx.gen_batch(nblk * T, n_stream_frames=40, ebn0_db=200.0)["iq"]
    torch.cuda.synchronize(device)
    slabs = torch.empty((T, C, nblk, 1920, 2), dtype=torch.int16, device=big.device)
    slabs.copy_(big.view(C, T, nblk, 1920, 2).permute(1, 0, 2, 3, 4))
    del big
    out = rx.alloc_outputs(nblk)
    settled = settle(torch, device, rx, slabs, 1, out)
    for k in range(warm):
        rx.rx_blocks(slabs[k], 1, out)
    torch.cuda.synchronize(device)
    rx.set_profiling(True)
    t0 = time.perf_counter()
    for k in range(steps):
        rx.rx_blocks(slabs[warm + k], 1, out)
    torch.cuda.synchronize(device)
    dt = time.perf_counter() - t0
    rx.set_profiling(False)
    kms, ncalls = rx.kernel_ms()
    path = rx.last_path()
    rx.close()
    del slabs
    torch.cuda.empty_cache()
    ro = roofline_obj(kms, ncalls, 1, C * nblk, f"full:{C}x{nblk}", path, dt / steps * 1e3)
    return {"workload": f"full chain, {C:,} channels x {nblk} blocks per step ({nblk * 40} ms of signal), noiseless: the default step of rounds 1-4",
            "ms": round(dt / steps * 1e3, 4), "value": round(C * nblk * 192 * steps / dt / 1e6, 3), "unit": "Msym/s",
            "frac": ro["frac"], "frac_wall": ro["frac_wall"], "achieved": ro["achieved"], "traffic": ro["traffic"], "avg_ms": ro["avg_ms"],
            "kernel_sum_ms": ro["kernel_sum_ms"], "path": path, "steps": steps, "warmup": warm, "settle_calls": settled}


def pci_bus_id(torch, device):
    """PCI bus id of the HIP device (domain:bus:device.function), through the runtime torch runs on."""
    import ctypes as C
    try:
        # the copy of the HIP runtime this process already runs on (torch's), by its path: dlopen of a path that is
        # loaded returns that library -- never a second runtime beside it (INTEGRATION.md "One HIP runtime per process")
        path = next((ln.split()[-1] for ln in open("/proc/self/maps") if "libamdhip64" in ln), None)
        if path:
            hip = C.CDLL(path)
            buf = C.create_string_buffer(32)
            if hip.hipDeviceGetPCIBusId(buf, 32, int(device)) == 0:
                return buf.value.decode()
    except OSError:
        pass
    p = torch.cuda.get_device_properties(device)
    return "%04x:%02x:%02x.0" % (getattr(p, "pci_domain_id", 0), getattr(p, "pci_bus_id", 0), getattr(p, "pci_device_id", 0))


def two_contexts_leg(args, torch, device, C, nblk):
    """What a HOST can have that a call cannot: the same channels as TWO contexts of C / 2 each, free-running
    on two streams -- each context's steps back to back on its own stream, no join per step -- so that one context's decoder
    runs beside the other's FIR stage across step boundaries and fills the tail of its kernel.  Same kernels, same results per
    channel (contexts are independent); wall clock per step of all C channels.  Not the headline: `value` keeps one context on one
    stream, where a kernel's HIP-event duration is what it takes alone."""
    import m17_sdr_amd as m
    steps, warm, parts = 20, 3, 2
    n = C // parts
    rxs = [m.Receiver(n, nblk, device=device) for _ in range(parts)]
    for rx in rxs:
        rx.set_option("fir_impl", 4)        # the library picks its kernels by a context's own channel count; the GPU carries C channels here
        for kv in args.option:
            name, value = kv.split("=")
            rx.set_option(name, int(value))
    T = steps + warm
    slabs = []
    for p, rx in enumerate(rxs):
        big = rx.gen_batch(nblk * T, n_stream_frames=40, ebn0_db=200.0, first_channel=p * n)["iq"]
        torch.cuda.synchronize(device)
        sl = torch.empty((T, n, nblk, 1920, 2), dtype=torch.int16, device=big.device)
        sl.copy_(big.view(n, T, nblk, 1920, 2).permute(1, 0, 2, 3, 4))
        del big
        slabs.append(sl)
    outs = [rx.alloc_outputs(nblk) for rx in rxs]
    streams = [torch.cuda.Stream(device=device) for _ in range(parts)]

    def run(k0, k1):
        for k in range(k0, k1):
            for p in range(parts):
                with torch.cuda.stream(streams[p]):
                    rxs[p].rx_blocks(slabs[p][k % T], 1, outs[p])
    t_end = time.perf_counter() + 0.4
    while time.perf_counter() < t_end:                       # whole passes of the stream: it stays continuous per context
        run(0, T)
        torch.cuda.synchronize(device)
    run(0, warm)
    torch.cuda.synchronize(device)
    t0 = time.perf_counter()
    run(warm, warm + steps)
    torch.cuda.synchronize(device)
    dt = time.perf_counter() - t0
    path = rxs[0].last_path()
    for rx in rxs:
        rx.close()
    del slabs
    torch.cuda.empty_cache()
    ms = dt / steps * 1e3
    return {"workload": f"full chain, {C:,} channels as {parts} contexts of {n:,} on {parts} streams, free-running (no join per step), {nblk} blocks per step, noiseless",
            "ms": round(ms, 4), "value": round(C * nblk * 192 * steps / dt / 1e6, 3), "unit": "Msym/s",
            "frac_wall": round(BYTES_FULL * C * nblk / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5), "path": path, "steps": steps, "warmup": warm,
            "note": "a host-side choice the C-ABI allows (contexts are independent); kernels of the two contexts overlap, so there is "
                    "no per-kernel time here"}


def host_cores():
    """(threads this job may use, physical cores and SMT threads of the host as /proc/cpuinfo lists them)."""
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    phys, threads = set(), 0
    try:
        pid = cid = None
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("processor"):
                threads += 1
            elif ln.startswith("physical id"):
                pid = ln.split(":")[1].strip()
            elif ln.startswith("core id"):
                cid = ln.split(":")[1].strip()
                phys.add((pid, cid))
    except OSError:
        pass
    return avail, (len(phys) or None), (threads or None)


class CapiFanout:
    """The C-ABI fan-out entries north_star names (include/m17gpu.h: m17gpu_shard_scatter_iq, m17gpu_pack_records,
    m17gpu_shard_gather_packed) driven through ctypes with an ncclComm_t of this job's own: rank 0 makes the unique id
    with RCCL's C API, the process group carries its 128 bytes to the other ranks, every rank joins with
    ncclCommInitRank -- what a C++ host does with its own bootstrap."""

    def __init__(self, torch, dist, rx, world, rank, dev):
        """Phase 1, local to this rank: load RCCL, rank 0 makes the unique id.  Nothing here can block on a peer, so a
        failure (no librccl next to torch, ...) can be agreed on before any rank enters ncclCommInitRank (join)."""
        import ctypes as C
        import m17_sdr_amd as m
        self.C, self.torch, self.dist, self.rx, self.world, self.rank, self.dev = C, torch, dist, rx, world, rank, dev
        self.lib, self.comm = m.lib(), None
        self.rccl = C.CDLL(os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so"))

        class UID(C.Structure):
            _fields_ = [("internal", C.c_char * 128)]
        self.UID, self.uid = UID, UID()
        if rank == 0 and self.rccl.ncclGetUniqueId(C.byref(self.uid)) != 0:
            raise RuntimeError("ncclGetUniqueId failed")

    def join(self):
        """Phase 2, collective: the id travels over the process group, every rank joins the communicator."""
        C, torch = self.C, self.torch
        if self.world > 1:
            t = torch.frombuffer(bytearray(bytes(self.uid.internal) if self.rank == 0 else bytes(128)), dtype=torch.uint8).to(self.dev)
            self.dist.broadcast(t, src=0)
            C.memmove(C.byref(self.uid), bytes(t.cpu().numpy().tobytes()), 128)
        self.comm = C.c_void_p()
        self.rccl.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, self.UID, C.c_int]
        if self.rccl.ncclCommInitRank(C.byref(self.comm), self.world, self.uid, self.rank) != 0:
            raise RuntimeError("ncclCommInitRank failed")
        n, d = C.c_int(-1), C.c_int(-1)
        self.rccl.ncclCommCount.argtypes = [C.c_void_p, C.POINTER(C.c_int)]
        self.rccl.ncclCommCuDevice.argtypes = [C.c_void_p, C.POINTER(C.c_int)]
        if self.rccl.ncclCommCount(self.comm, C.byref(n)) != 0 or self.rccl.ncclCommCuDevice(self.comm, C.byref(d)) != 0:
            raise RuntimeError("ncclCommCount / ncclCommCuDevice failed")
        self.count, self.cu_device = n.value, d.value       # RCCL's own view: ranks in the communicator, this rank's device

    def _chk(self, rc, what):
        if rc != 0:
            raise RuntimeError(f"{what}: {self.lib.m17gpu_last_error().decode()}")

    def scatter(self, full, total, nblk, mine, stream):
        self._chk(self.lib.m17gpu_shard_scatter_iq(self.rx._ctx, self.comm, self.rank, self.world, 0,
                                                   full.data_ptr() if full is not None else None, total, nblk, mine.data_ptr(),
                                                   self.C.c_void_p(stream.cuda_stream)), "m17gpu_shard_scatter_iq")

    def gather(self, packed, offs, total, packed_all, offs_all, stream):
        totals = (self.C.c_int32 * self.world)()
        self._chk(self.lib.m17gpu_shard_gather_packed(self.rx._ctx, self.comm, self.rank, self.world, 0, packed.data_ptr(),
                                                      int(packed.shape[0]), offs.data_ptr(),
                                                      total, packed_all.data_ptr() if packed_all is not None else None,
                                                      int(packed_all.shape[0]) if packed_all is not None else 0,
                                                      offs_all.data_ptr() if offs_all is not None else None, totals,
                                                      self.C.c_void_p(stream.cuda_stream)), "m17gpu_shard_gather_packed")
        return list(totals)

    def close(self):
        if self.comm:
            self.rccl.ncclCommDestroy(self.comm)
            self.comm = None


def fanout_legs(args, torch, dist, rx, out, iq_step, world, rank, backend, C, nblk, mode):
    """SURVEY 8(e) "with fan-out": rank 0 holds the IQ of all world x C channels of one step and fans it out to the
    owning ranks point to point (one send per peer, xGMI is a full mesh); every rank runs the step; the step's records
    are packed on the device (valid rows only: m17gpu_pack_records) and come back to rank 0 with their offset tables.
    Two transports: `torch` = m17_sdr_amd/shard.py over torch.distributed (RCCL for device tensors; under the gloo
    rehearsal backend the same calls stage through host memory), `capi` = the C-ABI entries of include/m17gpu.h on an
    ncclComm_t of the job's own.  Timed apart from the compute-only region, leg by leg; then the same step
    double-buffered -- the scatter of step k+1 on a second stream beside the compute of step k -- as
    `with_fanout_overlapped`."""
    from m17_sdr_amd import shard
    dev = torch.device("cuda", rx.device)
    total = world * C
    full = None
    reps = 5
    t_sc, t_cp, t_ga = [], [], []
    flag_dev = dev if backend == "nccl" else "cpu"
    transport = args.fanout or "torch"
    if transport == "capi" and backend != "nccl":
        transport = "torch"                      # the C-ABI entries move device memory over RCCL only

    def all_ok(err):
        """A leg that failed on ANY rank ends the legs on EVERY rank: one rank leaving alone would strand its peers
        in a recv or a barrier until the process-group timeout."""
        if world == 1:
            return err is None
        f = torch.tensor([0 if err is None else 1], dtype=torch.int32, device=flag_dev)
        dist.all_reduce(f, op=dist.ReduceOp.MAX)
        return int(f.item()) == 0

    def barrier():
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()

    # what can fail on ONE rank alone (building the staging tensor, a bad argument) is agreed on before any rank enters
    # the transfer: a peer already blocked in a matching send / recv could not be told afterwards
    err, capi = None, None
    try:
        if os.environ.get("M17_BENCH_INJECT_FANOUT_FAILURE") == str(rank):           # test hook (tests/test_a_bench_ranks.py)
            raise RuntimeError("injected fan-out failure")
        if os.environ.get("M17_BENCH_INJECT_FANOUT_HANG") == str(rank):              # test hook: a transfer that never returns
            time.sleep(3600)
        if rank == 0:
            full = iq_step.repeat(world, 1, 1, 1) if world > 1 else iq_step         # content is irrelevant to the transfer
        if transport == "capi":
            capi = CapiFanout(torch, dist, rx, world, rank, dev)
    except Exception as e:                                       # noqa: BLE001 -- reported in the line
        err = f"fan-out set-up on rank {rank}: {type(e).__name__}: {e}"[:300]
    if not all_ok(err):
        return {"fanout_error": err or "fan-out set-up failed on another rank"}
    if capi:
        try:
            capi.join()                                          # collective: every rank got here (agreed above)
        except Exception as e:                                   # noqa: BLE001
            err = f"communicator on rank {rank}: {type(e).__name__}: {e}"[:300]
        if not all_ok(err):
            return {"fanout_error": err or "ncclCommInitRank failed on another rank"}

    main = torch.cuda.current_stream(dev)
    cap = int(out["rec_cap"])
    packed = torch.empty((C * cap, 64), dtype=torch.uint8, device=dev)
    offs = torch.empty((C + 1,), dtype=torch.int32, device=dev)
    packed_all = torch.empty((total * cap, 64), dtype=torch.uint8, device=dev) if (capi and rank == 0) else None
    offs_all = torch.empty((total + 1,), dtype=torch.int32, device=dev) if (capi and rank == 0) else None
    shards = [torch.empty((C, nblk, 1920, 2), dtype=torch.int16, device=dev) for _ in range(2)]
    rec_rows = [0]

    def scatter(stream, k):
        if capi:
            capi.scatter(full, total, nblk, shards[k % 2], stream)
            return shards[k % 2]
        with torch.cuda.stream(stream):
            mine = shard.scatter_iq(full, total, nblk, src=0, device=dev) if world > 1 else full
        mine.record_stream(main)
        return mine

    def gather():
        rx.pack_records(out, packed, offs)
        if capi:
            tot = capi.gather(packed, offs, total, packed_all, offs_all, main)
            rec_rows[0] = tot[rank] if rank == 0 else int(offs[-1].item())
        elif world > 1:
            _, _, tot = shard.gather_packed(packed, offs, dst=0)
            rec_rows[0] = tot[rank] if tot else int(offs[-1].item())
        else:
            rec_rows[0] = int(offs[-1].item())

    for _ in range(reps + 1):
        barrier(); t0 = time.perf_counter()
        err, mine = None, None
        try:
            mine = scatter(main, 0)
        except Exception as e:                                   # noqa: BLE001 -- reported in the line
            err = f"scatter: {type(e).__name__}: {e}"[:300]
        if not all_ok(err):
            return {"fanout_error": err or "the scatter failed on another rank"}
        barrier(); t1 = time.perf_counter()
        rx.rx_blocks(mine, mode, out)
        barrier(); t2 = time.perf_counter()
        try:
            gather()
        except Exception as e:                                   # noqa: BLE001
            err = f"gather: {type(e).__name__}: {e}"[:300]
        if not all_ok(err):
            return {"fanout_error": err or "the gather failed on another rank"}
        barrier(); t3 = time.perf_counter()
        t_sc.append(t1 - t0); t_cp.append(t2 - t1); t_ga.append(t3 - t2)

    # double-buffered: the scatter of step k+1 on a second stream beside the compute (and the gather) of step k
    t_ov, err = None, None
    try:
        side = torch.cuda.Stream(device=dev)
        nxt = scatter(main, 0)
        barrier(); t0 = time.perf_counter()
        for k in range(reps):
            cur = nxt
            side.wait_stream(main)                               # buffer k+1 was last read two steps ago, on main
            nxt = scatter(side, k + 1)
            rx.rx_blocks(cur, mode, out)
            gather()
            main.wait_stream(side)
        barrier(); t_ov = (time.perf_counter() - t0) / reps
    except Exception as e:                                       # noqa: BLE001
        err = f"overlapped leg: {type(e).__name__}: {e}"[:300]
    if not all_ok(err):
        t_ov = None
    vals = torch.tensor([sum(t_sc[1:]) / reps, sum(t_cp[1:]) / reps, sum(t_ga[1:]) / reps, t_ov if t_ov is not None else -1.0],
                        dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
    if world > 1:
        dist.all_reduce(vals, op=dist.ReduceOp.MAX)
    sc, cp, ga, ov = (float(v) * 1e3 for v in vals.tolist())
    capi_count = (capi.count, capi.cu_device) if capi else (None, None)
    if capi:
        capi.close()
    rec_bytes = rec_rows[0] * 64 + (C + 1) * 4
    return {"fanout_ms": round(sc, 4), "gather_ms": round(ga, 4), "compute_ms_in_this_leg": round(cp, 4),
            "overlapped_ms_per_step": round(ov, 4) if ov > 0 else None,
            "iq_bytes_per_peer": C * nblk * 7680, "records_bytes_per_rank": rec_bytes,
            "records_bytes_unpacked": int(out["recs"].numel()) + 4 * C,
            "scatter_GBps_from_root": round((world - 1) * C * nblk * 7680 / (sc * 1e-3) / 1e9, 2) if world > 1 and sc > 0 else None,
            "reps": reps, "ncclCommCount": capi_count[0], "ncclCommCuDevice": capi_count[1],
            "transport": ("C-ABI m17gpu_shard_scatter_iq / m17gpu_pack_records / m17gpu_shard_gather_packed on an ncclComm_t (RCCL)" if capi
                          else "m17_sdr_amd.shard over torch.distributed: RCCL point-to-point" if backend == "nccl"
                          else f"m17_sdr_amd.shard over torch.distributed: {backend} (rehearsal, staged through host)")}


def run_rank(args):
    import torch
    import torch.distributed as dist
    import m17_sdr_amd as m
    from m17_sdr_amd.shard import channel_range

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (the product path has no CPU fallback)")
    ndev = max(1, torch.cuda.device_count())
    backend = os.environ.get("M17_BENCH_BACKEND", "nccl")   # "gloo" to rehearse the N>1 path on a 1-GPU box
    if world > ndev and backend == "nccl":
        raise SystemExit(f"bench.py: {world} ranks need {world} GPUs, {ndev} visible (M17_BENCH_BACKEND=gloo rehearses on fewer)")
    local = local % ndev                                    # rehearsal: several ranks on one card
    torch.cuda.set_device(local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend)

    C, nblk = args.channels, args.blocks
    mode = 0 if args.workload == "frontend" else 1
    T = args.warmup + args.steps
    rx = m.Receiver(C, nblk, device=local)
    for kv in args.option:
        name, value = kv.split("=")
        rx.set_option(name, int(value))
    iq, sig, Tg = make_input(args, rank, torch, rx, C, nblk, T)
    out = rx.alloc_outputs(nblk, want_syms=(mode == 0 and not args.no_syms))

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    settled = 0
    if not args.no_settle:
        settled = settle(torch, local, rx, iq[:Tg], mode, out)
    for k in range(args.warmup):
        rx.rx_blocks(iq[k], mode, out)
    barrier()
    rx.set_profiling(True)
    t0 = time.perf_counter()
    for k in range(args.steps):
        rx.rx_blocks(iq[args.warmup + k], mode, out)
    barrier()
    dt = time.perf_counter() - t0
    rx.set_profiling(False)
    kms, ncalls = rx.kernel_ms()
    path = rx.last_path()

    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    # who took part, as each rank sees itself: its device (PCI bus id) and the channel range it owns -- the day a
    # scaling run happens, "did N ranks on N different GPUs each process their shard" is answerable from the line
    me = {"rank": rank, "local_rank": local, "host": socket.gethostname(), "device": local, "pci_bus_id": pci_bus_id(torch, local),
          "shard_range": list(channel_range(rank, world, world * C)), "channels": C}
    if world > 1:
        ranks = [None] * world
        dist.all_gather_object(ranks, me)
    else:
        ranks = [me]

    syms = world * C * nblk * 192 * args.steps
    msym = syms / dt / 1e6
    ms_step = dt / args.steps * 1e3
    wl = ("full chain incl. soft Viterbi + depuncture/deinterleave/Golay + LICH/LSF bookkeeping, %s channels per GPU x %d blocks per "
          "step = %d ms of signal buffered per call (BASELINE configs[3]/[4] per-GPU size; configs[2] at 1,024)" % (f"{C:,}", nblk, nblk * 40)) if mode == 1 else \
         ("front end: limiter/discriminator + polyphase RRC timing recovery + sync correlator, %s channels per GPU "
          "(BASELINE configs[1] at 1,024)" % f"{C:,}")
    line = {
        "metric": "M17 symbols demodulated + decoded per second (real-time 48 kHz channels = value*1e6/4800)",
        "value": round(msym, 3), "unit": "Msym/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(ms_step, 4), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": wl, "mode": "full" if mode == 1 else "frontend",
                   "channels_per_gpu": C, "channels_total": world * C, "blocks_per_step": nblk, "samples_per_block": 1920,
                   "realtime_channels": int(msym * 1e6 / 4800), "realtime_channels_per_gpu": int(msym * 1e6 / 4800 / world),
                   "ebn0_db": args.ebn0,
                   "settle_calls": settled,
                   "settle_note": "untimed conditioning in front of the warm-up steps: the same step back to back for ~0.4 s, so that the "
                                  "shader clock has followed the load (DESIGN.md section 6; --no-settle to measure the ramp instead)",
                   "signal_source": ("m17gpu_gen_batch (device), one continuous stream per channel, %d distinct steps "
                                     "(%.1f s) cut into %d-block steps%s" % (Tg, 0.04 * nblk * Tg, nblk, ", wrapping" if Tg < T else "")
                                     if args.gen == "gpu" else "m17gen_batch (host, tiled), the same slab every step"),
                   "parallelism": f"channel-sharded x{world}, one process per GPU, no data-path collective in the timed region"},
        "roofline": roofline_obj(kms, ncalls, mode, C * nblk,
                                 f"{args.workload}{'-noisy' if (args.ebn0 < 100.0 and args.noise_cutoff > 0) else ''}:{C}x{nblk}",
                                 path, ms_step),
        "ranks": ranks,
        "collectives": {"backend": ("RCCL (torch.distributed 'nccl')" if backend == "nccl" else backend) if world > 1 else None,
                        "world_size": dist.get_world_size() if world > 1 else 1,
                        "distinct_devices": len({(r["host"], r["pci_bus_id"]) for r in ranks}),
                        "ncclCommCount": None,
                        "note": "ncclCommCount = RCCL's own count of the ranks in the communicator the C-ABI fan-out legs run on "
                                "(filled in by those legs: N > 1, or --fanout capi at N = 1); world_size = torch.distributed's"},
    }
    if args.option:
        line["config"]["options"] = list(args.option)
    if (world > 1 or args.fanout) and not args.no_fanout:
        # After the timed region and outside `value`: neither a failure nor a HANG of the transfer legs may cost the
        # measurement.  A collective that never returns cannot be cancelled, so a watchdog prints the line as it stands
        # (rank 0) -- with whatever legs have finished -- and ends the process on every rank when a set of legs overruns
        # its allowance.  N > 1 without --fanout runs BOTH bindings one after the other: the torch.distributed one
        # (rehearsed over gloo with two ranks) first, then the C-ABI entries on their own communicator (rehearsed with
        # one rank only: no multi-GPU box so far), so that a node that has the GPUs exercises both.
        import threading
        allowance = float(os.environ.get("M17_BENCH_FANOUT_TIMEOUT", "120"))
        transports = [args.fanout] if args.fanout else (["torch", "capi"] if backend == "nccl" else ["torch"])
        current = ["fanout"]

        def overrun():
            if rank == 0:
                line[current[0]] = {"fanout_error": f"the transfer legs did not finish within {allowance:g} s; abandoned"}
                line["cpu_baseline"] = None
                sys.stdout.write(json.dumps(line) + "\n")
                sys.stdout.flush()
            os._exit(EXIT_FANOUT_ABANDONED)     # not 0: the legs are missing from the line; not a crash either

        for ti, tr in enumerate(transports):
            key = "fanout" if ti == 0 else f"fanout_{tr}"
            current[0] = key
            dog = threading.Timer(allowance, overrun)
            dog.daemon = True
            dog.start()
            args.fanout = tr
            try:
                fan = fanout_legs(args, torch, dist, rx, out, iq[args.warmup % len(iq)], world, rank, backend, C, nblk, mode)
            except Exception as e:                               # noqa: BLE001 -- a dead peer, a failed barrier: reported, the line survives
                fan = {"fanout_error": f"{type(e).__name__}: {e}"[:300]}
            dog.cancel()
            line[key] = fan
            if fan.get("ncclCommCount") is not None:
                line["collectives"]["ncclCommCount"] = fan["ncclCommCount"]
            sfx = "" if ti == 0 else f"_{tr}"
            if "fanout_ms" in fan:
                wf = ms_step + fan["fanout_ms"] + fan["gather_ms"]
                line["with_fanout" + sfx] = {"ms_per_step": round(wf, 4), "value": round(world * C * nblk * 192 / (wf * 1e-3) / 1e6, 3),
                                             "unit": "Msym/s", "note": "compute step + scatter of the IQ from rank 0 + gather of the packed records, one after the other"}
                if fan.get("overlapped_ms_per_step"):
                    wo = fan["overlapped_ms_per_step"]
                    line["with_fanout_overlapped" + sfx] = {"ms_per_step": round(wo, 4), "value": round(world * C * nblk * 192 / (wo * 1e-3) / 1e6, 3),
                                                            "unit": "Msym/s", "note": "double-buffered: the scatter of step k+1 on a second stream beside "
                                                                                       "the compute and the packed gather of step k"}
    if rank == 0:
        rx.close()
        del iq
        torch.cuda.empty_cache()
        if world == 1 and not args.no_noisy and mode == 1 and args.ebn0 >= 100.0:
            line["noisy"] = noisy_leg(args, torch, local, C, nblk)
        if world == 1 and not args.no_step12 and mode == 1 and (C, nblk) == (16384, 16) and args.ebn0 >= 100.0:
            line["step_12_blocks"] = step12_leg(args, torch, local, C)
        if world == 1 and not args.no_step12 and mode == 1 and (C, nblk) == (16384, 16) and args.ebn0 >= 100.0:
            line["two_contexts"] = two_contexts_leg(args, torch, local, C, nblk)
        if world == 1 and not args.no_fir_stage and mode == 1:
            line["fir_stage"] = fir_stage(args, torch, local)
            line["fir_stage_16384"] = fir_stage(args, torch, local, C=16384, nblk=16, steps=20, warm=3)
            line["fir_stage_16384x12"] = fir_stage(args, torch, local, C=16384, nblk=12, steps=20, warm=3)
            line["fir_stage_16384x48"] = fir_stage(args, torch, local, C=16384, nblk=48, steps=8, warm=2)
        if not args.no_cpu_baseline and world == 1:
            line["cpu_baseline"] = cpu_baseline(mode, sig)
        else:
            line["cpu_baseline"] = None
        print(json.dumps(line), flush=True)
    else:
        rx.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    return 0


def main():
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return launch_ranks(args)           # before anything touches the GPU
    if args.gpus > 1 and int(os.environ.get("WORLD_SIZE", "1")) != args.gpus:
        sys.stderr.write("bench.py: --gpus disagrees with WORLD_SIZE; using WORLD_SIZE\n")
    return run_rank(args)


if __name__ == "__main__":
    sys.exit(main())
