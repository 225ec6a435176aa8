"""What would a call split into two channel halves on two streams save WHEN EVERY CALL IS JOINED on the caller's stream?
(profiles/r05_two_contexts_two_streams.txt measured free-running halves: their steps also overlap across calls.)
One receiver of C channels against two receivers of C / 2 -- half A on a high-priority stream, half B on the caller's --
forked from and joined on the caller's stream at every step, steps back to back.
   python scripts/exp_split_join.py [channels] [blocks] [mode]"""
import sys, os, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import m17_sdr_amd as m
C = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
nblk = int(sys.argv[2]) if len(sys.argv) > 2 else 16
mode = int(sys.argv[3]) if len(sys.argv) > 3 else 1
T = 8
gen = m.Receiver(C, nblk)
big = gen.gen_batch(nblk * T)["iq"]
slabs = torch.empty((T, C, nblk, 1920, 2), dtype=torch.int16, device=big.device)
slabs.copy_(big.view(C, T, nblk, 1920, 2).permute(1, 0, 2, 3, 4))
del big
gen.close()
def run(parts, prio):
    n = C // parts
    rxs = [m.Receiver(n, nblk) for _ in range(parts)]
    for rx in rxs:
        rx.set_option("fir_impl", 4)
    outs = [rx.alloc_outputs(nblk, want_syms=(mode == 0)) for rx in rxs]
    main = torch.cuda.current_stream()
    side = [torch.cuda.Stream(priority=(-1 if prio else 0)) for _ in range(parts - 1)]
    def step(k):
        if parts > 1:
            fork = torch.cuda.Event(); fork.record(main)
        for p in range(parts - 1):
            side[p].wait_event(fork)
            with torch.cuda.stream(side[p]):
                rxs[p].rx_blocks(slabs[k][p * n:(p + 1) * n], mode, outs[p])
        rxs[-1].rx_blocks(slabs[k][(parts - 1) * n:], mode, outs[-1])
        for p in range(parts - 1):
            ev = torch.cuda.Event(); ev.record(side[p]); main.wait_event(ev)
    t_end = time.perf_counter() + 0.5
    while time.perf_counter() < t_end:
        for k in range(T): step(k)
        torch.cuda.synchronize()
    reps = 5
    t0 = time.perf_counter()
    for _ in range(reps):
        for k in range(T): step(k)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / (reps * T) * 1e3
    for rx in rxs: rx.close()
    return dt
for rep in range(2):
    for parts, prio in ((1, 0), (2, 0), (2, 1), (4, 1)):
        print(f"{parts} receiver(s) of {C // parts} channels, joined per step, side streams {'high priority' if prio else 'default priority'}, "
              f"{nblk} blocks, mode {mode}: {run(parts, prio):.4f} ms per step of all {C} channels", flush=True)
