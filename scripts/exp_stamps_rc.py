"""k_rx_chan6 (fir_impl 4), instrumented build (make stamps): per-wave ticks in the front-end tiles and in the timing phases,
wave lifetimes and residency.   python scripts/exp_stamps_rc.py [channels] [blocks] [mode]"""
import sys, os, ctypes as C, time, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import m17_sdr_amd._lib as L_
L_.LIB_PATH = L_.LIB_PATH.replace("libm17gpu.so", "libm17gpu_stamps.so")
import m17_sdr_amd as m
Cn = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
nblk = int(sys.argv[2]) if len(sys.argv) > 2 else 16
mode = int(sys.argv[3]) if len(sys.argv) > 3 else 0
T = 6
rx = m.Receiver(Cn, nblk)
rx.set_option("fir_impl", int(os.environ.get("M17_FIR_IMPL", "4")))
big = rx.gen_batch(nblk * T)["iq"]
slabs = torch.empty((T, Cn, nblk, 1920, 2), dtype=torch.int16, device=big.device)
slabs.copy_(big.view(Cn, T, nblk, 1920, 2).permute(1, 0, 2, 3, 4))
del big
out = rx.alloc_outputs(nblk, want_syms=(mode == 0))
t_end = time.perf_counter() + 0.4
while time.perf_counter() < t_end:
    for k in range(T): rx.rx_blocks(slabs[k], mode, out)
    torch.cuda.synchronize()
rx.set_profiling(True)
rx.rx_blocks(slabs[0], mode, out)
torch.cuda.synchronize()
print("kernel ms:", [round(v, 4) for v in rx.kernel_ms()[0]])
st = np.zeros((16384, 4), np.uint64)
m.lib().m17gpu_debug_rc_stamps(st.ctypes.data_as(C.c_void_p))
st = st[:min(16384, Cn)].astype(np.float64)
fe, tm = st[:, 0], st[:, 1]
life = (st[:, 3] - st[:, 2]) / 100.0
span = (st[:, 3].max() - st[:, 2].min()) / 100.0
print(f"per wave: front-end tiles {fe.mean():.0f} ticks ({fe.mean() / nblk:.0f} per block, {100 * fe.sum() / (fe.sum() + tm.sum()):.1f} %), "
      f"timing phases {tm.mean():.0f} ticks ({tm.mean() / nblk:.0f} per block); p10/p50/p90 of the sum: "
      + " ".join(f"{np.percentile(fe + tm, q):.0f}" for q in (10, 50, 90)))
print(f"wave lifetime mean {life.mean():.1f} us (min {life.min():.1f}, max {life.max():.1f}); kernel span {span:.1f} us; "
      f"resident on average {life.sum() / span:.0f} waves; ticks per us {((fe + tm) / life).mean():.0f}")
for q in (0.1, 0.25, 0.5, 0.75, 0.9):
    tt = st[:, 2].min() + q * (st[:, 3].max() - st[:, 2].min())
    print("   resident at %2.0f %% of the span: %d" % (100 * q, int(((st[:, 2] <= tt) & (st[:, 3] > tt)).sum())))
