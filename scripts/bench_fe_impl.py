"""A/B of the front-end kernel variants: python scripts/bench_fe_impl.py C nblk impl [impl...]"""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import m17_sdr_amd as m
C, nblk = int(sys.argv[1]), int(sys.argv[2])
impls = [int(x) for x in sys.argv[3:]] or [0]
T = 8
gen = m.Receiver(C, nblk)
big = gen.gen_batch(nblk * T)["iq"]
slabs = big.view(C, T, nblk, 1920, 2).permute(1, 0, 2, 3, 4).contiguous()
del big
gen.close()
for impl in impls:
    rx = m.Receiver(C, nblk)
    rx.set_option("fe_impl", impl)
    out = rx.alloc_outputs(nblk, want_syms=True)
    for k in range(2): rx.rx_blocks(slabs[k], 0, out)
    torch.cuda.synchronize()
    rx.set_profiling(True)
    for k in range(2, T): rx.rx_blocks(slabs[k], 0, out)
    torch.cuda.synchronize()
    ms, n = rx.kernel_ms()
    gbs = C * nblk * 7680 / (ms[0] * 1e-3) / 1e9
    print(f"fe_impl {impl}: C={C} nblk={nblk}  k_frontend {ms[0]:.4f} ms = {gbs:.0f} GB/s of IQ   k_sync {ms[1]:.4f}")
    rx.close()
