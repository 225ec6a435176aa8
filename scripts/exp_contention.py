"""Which resource do the front end and the timing stage fight over when they run side by side?  Each stage beside a
synthetic partner on another stream: a register-light arithmetic kernel (torch elementwise chain on an L2-resident
tensor) and a streaming copy (1 GB).   python scripts/exp_contention.py"""
import sys, os, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import m17_sdr_amd as m
C, nblk = 16384, 12
rxA, rxB = m.Receiver(C, nblk), m.Receiver(C, nblk)
rxB.set_option("sync_impl", 8)
iq = rxA.gen_batch(nblk)["iq"]
disc, offs = rxB.frontend(iq)
outB = rxB.alloc_outputs(nblk)
small = torch.rand(1 << 20, device="cuda")           # 4 MB: cache-resident
big_a = torch.empty(1 << 28, dtype=torch.uint8, device="cuda"); big_b = torch.empty_like(big_a)
def valu():                                           # ~arithmetic only
    x = small
    for _ in range(24): x = torch.sin(x) * 1.0001 + 0.5
    return x
def copy(): big_b.copy_(big_a)
def fe(): m.api._check(m.lib().m17gpu_rx_blocks(rxA._ctx, m.api._ptr(iq), nblk, 0, None, 0, None, None, None, rxA._stream()), "fe") if False else rxA.frontend(iq)
def sync(): rxB.sync_frame(disc, outB)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
def t(fa, fb=None, n=10):
    for _ in range(2):
        with torch.cuda.stream(s1): fa()
        if fb:
            with torch.cuda.stream(s2): fb()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        with torch.cuda.stream(s1): fa()
        if fb:
            with torch.cuda.stream(s2): fb()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3
names = {"front end (+dc_remove)": fe, "timing stage": sync, "arith partner": valu, "copy partner": copy}
alone = {k: t(f) for k, f in names.items()}
for k, v in alone.items(): print(f"{k:24s} alone {v:.3f} ms")
for a in ("front end (+dc_remove)", "timing stage"):
    for b in ("arith partner", "copy partner", "timing stage" if a.startswith("front") else "front end (+dc_remove)"):
        both = t(names[a], names[b])
        print(f"{a:24s} beside {b:24s}: {both:.3f} ms  (sum of alone {alone[a] + alone[b]:.3f}, max {max(alone[a], alone[b]):.3f})")
