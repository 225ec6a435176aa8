import sys, os, ctypes as C, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import m17_sdr_amd._lib as L
L.LIB_PATH = L.LIB_PATH.replace("libm17gpu.so", "libm17gpu_stamps.so")
import m17_sdr_amd as m
Cn, nblk, T = 1024, 50, 4
rx = m.Receiver(Cn, nblk)
big = rx.gen_batch(nblk * T)["iq"]
slabs = big.view(Cn, T, nblk, 1920, 2).permute(1, 0, 2, 3, 4).contiguous()
out = rx.alloc_outputs(nblk, want_syms=True)
for k in range(T): rx.rx_blocks(slabs[k], 0, out)
torch.cuda.synchronize()
st = (C.c_ulonglong * 16)()
m.lib().m17gpu_debug_stamps(st)
print(f"timing wave: work {st[0]/nblk:.0f} ticks/block, waiting for the framer {st[1]/nblk:.0f}")
print(f"framer wave: work {st[2]/nblk:.0f} ticks/block, waiting for the timing wave {st[3]/nblk:.0f}")
