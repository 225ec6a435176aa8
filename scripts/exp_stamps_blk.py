"""Phase shares of k_sync_frame_blk (instrumented build), channel 7."""
import sys, os, ctypes as C, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import m17_sdr_amd._lib as L
L.LIB_PATH = L.LIB_PATH.replace("libm17gpu.so", "libm17gpu_stamps.so")
import m17_sdr_amd as m
Cn, nblk, T = int(sys.argv[1]), int(sys.argv[2]), 4
mode = int(sys.argv[3]) if len(sys.argv) > 3 else 0
rx = m.Receiver(Cn, nblk); rx.set_option("sync_impl", 10)
big = rx.gen_batch(nblk * T)["iq"]
slabs = big.view(Cn, T, nblk, 1920, 2).permute(1, 0, 2, 3, 4).contiguous()
out = rx.alloc_outputs(nblk, want_syms=(mode == 0))
for k in range(T): rx.rx_blocks(slabs[k], mode, out)
torch.cuda.synchronize()
st = np.zeros(16, np.uint64)
m.lib().m17gpu_debug_stamps(st.ctypes.data_as(C.c_void_p))
st = st.astype(np.float64)
names = ["loop", "commit+fetch", "taps+3 FIR", "votes+test", "accept | general", "framer"]
print("blocks accepted as computed %d, through the general code %d (of %d)" % (st[8], st[9], nblk))
print("ticks per block: " + "  ".join(f"{n}={st[i]/nblk:.0f}" for i, n in enumerate(names)), " total=%.0f" % (st[:6].sum() / nblk))
