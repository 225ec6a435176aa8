"""Phase shares of k_rx_fused (M17_FIR_IMPL=2) or k_sync_frame_wave (=1) at the headline size (instrumented build: make -C m17_sdr_amd/csrc stamps)."""
import sys, os, ctypes as C, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import m17_sdr_amd._lib as L
L.LIB_PATH = L.LIB_PATH.replace("libm17gpu.so", "libm17gpu_stamps.so")
import m17_sdr_amd as m
Cn, nblk, T = int(sys.argv[2]) if len(sys.argv) > 2 else 16384, 12, int(os.environ.get('M17_STEPS', '4'))
mode = int(sys.argv[1]) if len(sys.argv) > 1 else 1
rx = m.Receiver(Cn, nblk)
rx.set_option("fir_impl", int(os.environ.get("M17_FIR_IMPL", "2")))
ebn0 = float(sys.argv[3]) if len(sys.argv) > 3 else 200.0
big = rx.gen_batch(nblk * T, ebn0_db=ebn0, noise_cutoff_hz=6250.0 if ebn0 < 100 else 0.0)["iq"]
slabs = big.view(Cn, T, nblk, 1920, 2).permute(1, 0, 2, 3, 4).contiguous()
out = rx.alloc_outputs(nblk, want_syms=(mode == 0))
for k in range(T):
    if k == T - 1: rx.set_profiling(True)
    rx.rx_blocks(slabs[k], mode, out)
torch.cuda.synchronize()
print('last call, kernel ms (front end, timing+framer, decode, bookkeeping):', [round(v, 4) for v in rx.kernel_ms()[0]])
st = np.zeros((4096, 8), np.uint64)
m.lib().m17gpu_debug_chan_stamps(st.ctypes.data_as(C.c_void_p))
st = st[:min(4096, Cn)].astype(np.float64)
tot = st[:, :7].sum(1)
print("per-wave ticks: mean %.0f; sum over the %d waves / 4096 wave slots = %.0f ticks" % (tot.mean(), len(tot), tot.sum() * (Cn / len(tot)) / 4096))
names = ["round top/tick", "FIR asm", "vote+commit of round", "syms out", "framer", "block head+tail (x commit)", "front-end phase (fused kernel)", "rounds"]
print("per-wave time of the last step (ticks): min %.0f  median %.0f  p90 %.0f  max %.0f" % (tot.min(), np.median(tot), np.percentile(tot, 90), tot.max()))
for i, n in enumerate(names):
    print(f"  {n:45s} {st[:, i].mean() / nblk:10.1f} per block" + (f"  ({100 * st[:, i].sum() / tot.sum():.1f} %)" if i < 7 else ""))
r = st[:, 7] / nblk
print("rounds per block, by channel (this step): " + "  ".join(f"p{q} {np.percentile(r, q):.1f}" for q in (0, 10, 50, 90, 99, 100)))
print("channels with more than 8 rounds per block: %d of %d; their share of all rounds: %.1f %%" % ((r > 8).sum(), len(r), 100 * r[r > 8].sum() / r.sum()))
