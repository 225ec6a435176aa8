import sys, ctypes as C, numpy as np, torch
sys.path.insert(0, '.')
import m17_sdr_amd._lib as L
L.LIB_PATH = L.LIB_PATH.replace("libm17gpu.so", "libm17gpu_stamps.so")
import m17_sdr_amd as m
Cn, nblk = int(sys.argv[1]) if len(sys.argv) > 1 else 256, 50
sig = m.generate_batch(min(Cn,128), nblk, n_stream_frames=400)
iq = torch.from_numpy(sig["iq"]).cuda()
if Cn > iq.shape[0]: iq = iq.repeat((Cn+iq.shape[0]-1)//iq.shape[0],1,1,1)[:Cn].contiguous()
rx = m.Receiver(Cn, nblk); out = rx.alloc_outputs(nblk, want_syms=True)
for _ in range(3): rx.rx_blocks(iq, 0, out)
torch.cuda.synchronize()
st = (C.c_ulonglong*16)()
m.lib().m17gpu_debug_stamps(st)
names = ["round top (tap load / vote tick)", "FIR", "vote scan + decide", "fence, prefetch issue, syms out", "framer", "end-of-block commit"]
tot = sum(st[:6])
for i, nme in enumerate(names): print(f"{nme:36s} {st[i]/nblk:9.1f} ticks/block  {100*st[i]/tot:5.1f}%")
print("total ticks/block", tot/nblk, " rounds/block", st[8]/nblk, " lone vote ticks/block", st[9]/nblk, " framer iters/block", st[10]/nblk)
