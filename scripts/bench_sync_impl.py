"""A/B of the timing/framer kernel variants on one workload: python scripts/bench_sync_impl.py C nblk mode impl [impl...]"""
import sys, os, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import m17_sdr_amd as m
C, nblk, mode = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
impls = [int(x) for x in sys.argv[4:]] or [6]
T = 12
gen = m.Receiver(C, nblk)
big = gen.gen_batch(nblk * T)["iq"]
slabs = big.view(C, T, nblk, 1920, 2).permute(1, 0, 2, 3, 4).contiguous()
del big
gen.close()
for impl in impls:
    rx = m.Receiver(C, nblk)
    rx.set_option("sync_impl", impl)
    out = rx.alloc_outputs(nblk, want_syms=(mode == 0))
    for k in range(2): rx.rx_blocks(slabs[k], mode, out)
    torch.cuda.synchronize()
    rx.set_profiling(True)
    for k in range(2, T): rx.rx_blocks(slabs[k], mode, out)
    torch.cuda.synchronize()
    ms, n = rx.kernel_ms()
    print(f"sync_impl {impl}: C={C} nblk={nblk} mode={mode}  k_frontend {ms[0]:.4f}  k_sync {ms[1]:.4f}  decode {ms[2]:.4f}  book {ms[3]:.4f} ms  (locked {int(rx.lock().sum())})")
    rx.close()
