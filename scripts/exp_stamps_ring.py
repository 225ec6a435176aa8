import sys, os, ctypes as C, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import m17_sdr_amd._lib as L
L.LIB_PATH = L.LIB_PATH.replace("libm17gpu.so", "libm17gpu_stamps.so")
import m17_sdr_amd as m
Cn, nblk, T = 1024, 50, 4
rx = m.Receiver(Cn, nblk); rx.set_option("sync_impl", 5)
big = rx.gen_batch(nblk * T)["iq"]
slabs = big.view(Cn, T, nblk, 1920, 2).permute(1, 0, 2, 3, 4).contiguous()
out = rx.alloc_outputs(nblk, want_syms=True)
for k in range(T): rx.rx_blocks(slabs[k], 0, out)
torch.cuda.synchronize()
st = (C.c_ulonglong * 16)()
m.lib().m17gpu_debug_stamps(st)
names = ["timing loop", "prefetch issue + nsyms", "framer: loop entry -> frame symbols read", "sync_check_grp", "flags + record + frame symbols out",
         "rest of framer (second pass, hunt)", "(unused)", "commit of next block + loop back"]
tot = sum(st[:8])
for i, n in enumerate(names): print(f"{n:48s} {st[i]/nblk:8.0f} ticks/block {100*st[i]/tot:5.1f}%")
print("total", tot / nblk)
