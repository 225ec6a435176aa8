"""Would two channel halves on two streams fill each other's kernel tails?  One receiver of C channels against two
receivers of C / 2 each on two streams (each with its own context: the same kernels on half the grid, concurrently).
   python scripts/exp_two_halves.py [channels] [blocks] [mode]"""
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
def run(parts):
    n = C // parts
    rxs = [m.Receiver(n, nblk) for _ in range(parts)]
    for rx in rxs:
        if nblk % 16 == 0: rx.set_option("fir_impl", 4)
    outs = [rx.alloc_outputs(nblk, want_syms=(mode == 0)) for rx in rxs]
    streams = [torch.cuda.Stream() for _ in range(parts)]
    def step(k):
        for p in range(parts):
            with torch.cuda.stream(streams[p]):
                rxs[p].rx_blocks(slabs[k][p * n:(p + 1) * n], mode, outs[p])
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
    for parts in (1, 2, 4):
        print(f"{parts} receiver(s) of {C // parts} channels on {parts} stream(s), {nblk} blocks, mode {mode}: {run(parts):.4f} ms per step of all {C} channels", flush=True)
