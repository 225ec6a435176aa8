"""Phase shares of k_sync_frame_tri (instrumented build: make -C m17_sdr_amd/csrc stamps), wave 0 and wave 1 of one channel."""
import sys, os, ctypes as C, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import m17_sdr_amd._lib as L
L.LIB_PATH = L.LIB_PATH.replace("libm17gpu.so", "libm17gpu_stamps.so")
import m17_sdr_amd as m
Cn, nblk, T = 1024, 50, 4
mode = int(sys.argv[1]) if len(sys.argv) > 1 else 0
rx = m.Receiver(Cn, nblk)
big = rx.gen_batch(nblk * T)["iq"]
slabs = big.view(Cn, T, nblk, 1920, 2).permute(1, 0, 2, 3, 4).contiguous()
out = rx.alloc_outputs(nblk, want_syms=(mode == 0))
for k in range(T): rx.rx_blocks(slabs[k], mode, out)
torch.cuda.synchronize()
st = np.zeros(16, np.uint64)
m.lib().m17gpu_debug_stamps(st.ctypes.data_as(C.c_void_p))
names = ["block top", "FIR+votes", "barrier", "scan", "accept/step", "framer+end"]
for w in range(2):
    v = st[6 * w: 6 * w + 6].astype(np.float64) / nblk
    print(f"wave {w}: " + "  ".join(f"{n}={x:.0f}" for n, x in zip(names, v)) + f"  total={v.sum():.0f} ticks/block")
