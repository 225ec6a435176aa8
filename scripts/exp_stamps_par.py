"""Phase shares of k_sync_frame_par's control wave (instrumented build: make -C m17_sdr_amd/csrc stamps).
   python scripts/exp_stamps_par.py [C] [nblk] [ebn0]"""
import sys, os, ctypes as C, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import m17_sdr_amd._lib as L
L.LIB_PATH = L.LIB_PATH.replace("libm17gpu.so", os.environ.get("M17_STAMPS_LIB", "libm17gpu_stamps.so"))
import m17_sdr_amd as m
Cn = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
nblk = int(sys.argv[2]) if len(sys.argv) > 2 else 50
ebn0 = float(sys.argv[3]) if len(sys.argv) > 3 else 200.0
T = 4
rx = m.Receiver(Cn, nblk)
rx.set_option("sync_impl", 8)
big = rx.gen_batch(nblk * T, ebn0_db=ebn0, noise_cutoff_hz=6250.0 if ebn0 < 100 else 0.0)["iq"]
slabs = big.view(Cn, T, nblk, 1920, 2).permute(1, 0, 2, 3, 4).contiguous()
out = rx.alloc_outputs(nblk, want_syms=True)
for k in range(T): rx.rx_blocks(slabs[k], 0, out)
torch.cuda.synchronize()
st = np.zeros((4096, 8), np.uint64)
m.lib().m17gpu_debug_chan_stamps(st.ctypes.data_as(C.c_void_p))
st = st[:min(4096, Cn)].astype(np.float64)
tot = st[:, :7].sum(1)
names = ["filter jobs (post .. done)", "run: top, slot lookup", "run: masks, popcounts", "run: prefix counts", "run: symbol copy", "run: state update, carried sum/dif", "lock flag wait + rest of the block", "runs"]
print("control wave, last step (ticks): min %.0f  median %.0f  max %.0f" % (tot.min(), np.median(tot), tot.max()))
for i, n in enumerate(names):
    print(f"  {n:40s} {st[:, i].mean() / nblk:10.1f} per block" + (f"  ({100 * st[:, i].sum() / tot.sum():.1f} %)" if i < 7 else ""))
