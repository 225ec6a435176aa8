#!/usr/bin/env python3
"""bench.py -- throughput of the batched M17 receive chain on N MI355X.

A "step" is one pass of the hot path (m17gpu_rx_blocks) over one batch of
synthetic IQ already resident in HBM: C channels x NBLK 1920-sample blocks per
GPU.  Default workload = BASELINE.json configs[1]: 1,024 channels, front end
only (discriminator + polyphase RRC timing recovery + sync correlator/framer).
`--workload full` runs configs[2] (adds demap/Viterbi/Golay/LSF bookkeeping).

One JSON line is printed by rank 0 (contract in the task statement), carrying
`roofline` (live HIP-event kernel durations against algorithmic bytes) and
`cpu_baseline` (the CPU oracle timed on this host on a bounded sample).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0                 # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
BYTES_FRONT = 7680 + 768              # IQ in + symbols out per channel-block
BYTES_FULL = 7680 + 64                # IQ in + 64-byte record out


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--channels", type=int, default=1024, help="channels per GPU")
    ap.add_argument("--blocks", type=int, default=50, help="1920-sample blocks per channel per step (2 s)")
    ap.add_argument("--workload", choices=["frontend", "full"], default="frontend")
    ap.add_argument("--ebn0", type=float, default=200.0, help="AWGN level of the synthetic IQ (>=100: none)")
    ap.add_argument("--unique", type=int, default=256, help="host generator: distinct generated channels (tiled to --channels)")
    ap.add_argument("--gen", choices=["gpu", "host"], default="gpu",
                    help="signal source: m17gpu_gen_batch on the device (every channel distinct) or the host generator")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-syms", action="store_true", help="front end: do not write the symbol stream")
    return ap.parse_args()


def make_input(args, rank, torch, rx):
    """Synthetic IQ for this rank's channel shard, resident in HBM before the timed region.
    Default: ONE continuous signal per channel, (warmup + steps) x blocks long, made on the device
    by m17gpu_gen_batch (every channel distinct, seeded by global channel id) and
    cut into per-step slabs [step][C][blocks] -- each step receives the next 2 s of every
    channel, as a live receiver would, instead of the same 2 s again (which breaks the frame
    phase at every step boundary and sends the framer hunting).  --gen host: the host generator,
    tiled to C channels, the same slab every step.
    Returns (list of per-step device tensors, host copy of <= 256 channels of the first timed slab)."""
    import m17_sdr_amd as m
    C, nblk = args.channels, args.blocks
    T = args.warmup + args.steps
    if args.gen == "gpu":
        # at most ~20 GB of distinct signal (plus as much again while it is cut into slabs): longer runs wrap
        # around, i.e. one frame-phase break per pass over the slabs instead of one per step
        Tg = max(1, min(T, int(20e9 // (C * nblk * 7680))))
        big = rx.gen_batch(nblk * Tg, n_stream_frames=40, ebn0_db=args.ebn0, first_channel=rank * C)["iq"]
        torch.cuda.synchronize()
        slabs = big.view(C, Tg, nblk, 1920, 2).permute(1, 0, 2, 3, 4).contiguous()
        del big
        torch.cuda.empty_cache()
        steps = [slabs[k % Tg] for k in range(T)]
        return steps, {"iq": steps[min(args.warmup, Tg - 1)][:min(256, C)].cpu().numpy()}
    uniq = min(args.unique, C)
    nthreads = max(1, min(16, (os.cpu_count() or 8) // max(1, args.gpus)))
    sig = m.generate_batch(uniq, nblk, n_stream_frames=40, ebn0_db=args.ebn0,
                           first_channel=rank * C, nthreads=nthreads)
    host = torch.from_numpy(sig["iq"])
    dev = torch.empty((C, nblk, 1920, 2), dtype=torch.int16, device="cuda")
    for c0 in range(0, C, uniq):
        n = min(uniq, C - c0)
        dev[c0:c0 + n].copy_(host[:n], non_blocking=False)
    return [dev] * T, sig


def cpu_baseline(args, sig):
    """The CPU oracle (port of the reference path) on a bounded sample of the
    same workload, all host threads, same run."""
    from tests import oracle
    # the GPU box gives one GPU's job a share of 16 host threads (see task notes)
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    cores = max(1, min(avail, 16))
    mode = 0 if args.workload == "frontend" else 1
    nch = sig["iq"].shape[0]
    iq = np.ascontiguousarray(sig["iq"][:nch])
    ch = oracle.Channels(nch)
    ch.rx_blocks(iq, mode=mode, want_syms=False, nthreads=cores)      # warm (tables, page faults)
    reps, t_used = 0, 0.0
    t0 = time.perf_counter()
    while t_used < 1.5 and reps < 400:             # ~24 core-seconds on 16 threads
        ch.rx_blocks(iq, mode=mode, want_syms=False, nthreads=cores)
        reps += 1
        t_used = time.perf_counter() - t0
    syms = nch * iq.shape[1] * 192 * reps
    # single-thread figure for comparison with SURVEY's 47.8 us/block
    ch1 = oracle.Channels(1)
    t1 = time.perf_counter()
    ch1.rx_blocks(iq[:1], mode=mode, want_syms=False, nthreads=1)
    us_blk = (time.perf_counter() - t1) / iq.shape[1] * 1e6
    return {"value": round(syms / t_used / 1e6, 3), "unit": "Msym/s", "cores": cores, "kind": "port",
            "sample": f"{nch} channels x {iq.shape[1]} blocks x {reps} passes of the same synthetic IQ, "
                      f"OpenMP over channels ({t_used:.2f} s wall); 1 thread: {us_blk:.1f} us/block",
            "realtime_channels": int(syms / t_used / 4800)}


def main():
    args = parse_args()
    import torch
    import torch.distributed as dist
    import m17_sdr_amd as m

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (the product path has no CPU fallback)")
    local = local % max(1, torch.cuda.device_count())      # rehearsal: several ranks on one card
    torch.cuda.set_device(local)
    backend = os.environ.get("M17_BENCH_BACKEND", "nccl")   # "gloo" to rehearse the N>1 path on a 1-GPU box
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend)

    C, nblk = args.channels, args.blocks
    mode = 0 if args.workload == "frontend" else 1
    rx = m.Receiver(C, nblk, device=local)
    iq, sig = make_input(args, rank, torch, rx)
    out = rx.alloc_outputs(nblk, want_syms=(mode == 0 and not args.no_syms))

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

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

    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    syms = world * C * nblk * 192 * args.steps
    msym = syms / dt / 1e6
    cb_per_launch = C * nblk
    per_unit = BYTES_FRONT if mode == 0 else BYTES_FULL
    names = ["k_frontend", "k_sync_frame", "k_worklist+k_decode", "k_bookkeeping"]
    used = [i for i in range(4) if kms[i] > 0.002]
    t_path_ms = sum(kms[i] for i in used)
    dom = max(used, key=lambda i: kms[i]) if used else 0
    achieved = per_unit * cb_per_launch / (t_path_ms * 1e-3) / 1e9 if t_path_ms > 0 else 0.0
    traffic = None
    tpath = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(tpath):
        try:
            tj = json.load(open(tpath))
            key = f"{args.workload}:{C}x{nblk}"
            traffic = tj.get(key)
        except Exception:
            traffic = None
    line = {
        "metric": "M17 symbols demodulated per second (real-time 48 kHz channels = value*1e6/4800)",
        "value": round(msym, 3), "unit": "Msym/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(dt / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": ("1,024-channel-class front end: limiter/discriminator + polyphase RRC timing "
                                "recovery + sync correlator (BASELINE configs[1])" if mode == 0 else
                                "full chain incl. soft Viterbi + depuncture/deinterleave/Golay (BASELINE configs[2])"),
                   "channels_per_gpu": C, "blocks_per_step": nblk, "samples_per_block": 1920,
                   "realtime_channels": int(msym * 1e6 / 4800), "ebn0_db": args.ebn0,
                   "signal_source": ("m17gpu_gen_batch (device), one continuous stream per channel (up to %.0f s) cut into steps" % (0.04 * nblk * (args.warmup + args.steps))
                                     if args.gen == "gpu" else "m17gen_batch (host, tiled), the same slab every step"),
                   "parallelism": f"channel-sharded x{world}, no data-path collective"},
        "roofline": {"bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic,
                     "kernel": "+".join(names[i] for i in used),
                     "dominant": names[dom],
                     "algorithmic_bytes_per_channel_block": per_unit,
                     "channel_blocks_per_launch": cb_per_launch,
                     "avg_ms": {names[i]: round(kms[i], 4) for i in used},
                     "calls_timed": ncalls},
    }
    if rank == 0:
        if not args.no_cpu_baseline and world == 1:
            line["cpu_baseline"] = cpu_baseline(args, sig)
        else:
            line["cpu_baseline"] = None
        print(json.dumps(line), flush=True)
    rx.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
