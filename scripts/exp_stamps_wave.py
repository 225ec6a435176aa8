"""Phase shares of k_sync_frame_wave at the headline size (instrumented build: make -C m17_sdr_amd/csrc stamps)."""
import sys, os, ctypes as C, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import m17_sdr_amd._lib as L
L.LIB_PATH = L.LIB_PATH.replace("libm17gpu.so", "libm17gpu_stamps.so")
import m17_sdr_amd as m
Cn, nblk, T = int(sys.argv[2]) if len(sys.argv) > 2 else 16384, 12, int(os.environ.get('M17_STEPS', '4'))
mode = int(sys.argv[1]) if len(sys.argv) > 1 else 1
rx = m.Receiver(Cn, nblk)
rx.set_option("sync_impl", 8)
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
print("per-wave ticks: mean %.0f; sum over the %d waves / 6144 wave slots = %.0f ticks" % (tot.mean(), len(tot), tot.sum() * (Cn / len(tot)) / 6144))
names = ["round top/tick", "FIR asm", "vote+commit of round", "syms out", "framer", "block head+tail (x commit)", "PROLOGUE: state + first block in (whole call, shown / nblk)", "rounds"]
print("per-wave time of the last step (ticks): min %.0f  median %.0f  p90 %.0f  max %.0f" % (tot.min(), np.median(tot), np.percentile(tot, 90), tot.max()))
for i, n in enumerate(names):
    print(f"  {n:45s} {st[:, i].mean() / nblk:10.1f} per block" + (f"  ({100 * st[:, i].sum() / tot.sum():.1f} %)" if i < 7 else ""))
r = st[:, 7] / nblk
print("rounds per block, by channel (this step): " + "  ".join(f"p{q} {np.percentile(r, q):.1f}" for q in (0, 10, 50, 90, 99, 100)))
print("channels with more than 8 rounds per block: %d of %d; their share of all rounds: %.1f %%" % ((r > 8).sum(), len(r), 100 * r[r > 8].sum() / r.sum()))
sx = np.zeros((4096, 2), np.uint64)
m.lib().m17gpu_debug_chan_stamps_x(sx.ctypes.data_as(C.c_void_p))
print("per wave, whole call: prologue (state + first block in) %.0f ticks, main loop %.0f, epilogue (state out, stores complete) %.0f" %
      (st[:, 6].mean(), st[:, :6].sum(1).mean(), sx[:len(st), 0].astype(np.float64).mean()))
sp = np.zeros((16384, 3), np.uint64)
m.lib().m17gpu_debug_wave_span(sp.ctypes.data_as(C.c_void_p))
sp = sp[:min(16384, Cn)]
t0, t1 = sp[:, 0].astype(np.float64), sp[:, 1].astype(np.float64)
span = (t1.max() - t0.min()) / 100.0                       # us (s_memrealtime: 100 MHz)
life = (t1 - t0) / 100.0
print("wave spans (s_memrealtime): first entry to last exit %.1f us; wave lifetime mean %.1f us (min %.1f, max %.1f); "
      "sum of lifetimes / span = %.0f waves resident on average (6,144 slots)" % (span, life.mean(), life.min(), life.max(), life.sum() / span))
for q in (0.1, 0.25, 0.5, 0.75, 0.9):
    tt = t0.min() + q * (t1.max() - t0.min())
    print("   resident at %2.0f %% of the span: %d" % (100 * q, int(((t0 <= tt) & (t1 > tt)).sum())))
hw = sp[:, 2].astype(np.uint64) & np.uint64(0xFFFFFFFF)
tk = (sp[:, 2].astype(np.uint64) >> np.uint64(32)).astype(np.float64)
mhz = tk / life
print("   s_memtime ticks per us of s_memrealtime over each wave's life: mean %.0f  p10 %.0f  p50 %.0f  p90 %.0f  (= shader clock in MHz while this kernel runs)" % (mhz.mean(), np.percentile(mhz, 10), np.percentile(mhz, 50), np.percentile(mhz, 90)))
print("   distinct (se, sh, cu) ids seen:", len(set(((int(h) >> 8) & 0xFFF) for h in hw)), " ticks per us inside the main loop: %.0f" % (st[:, :6].sum(1).mean() / max(1e-9, (life[:len(st)].mean()))))
