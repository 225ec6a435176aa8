import sys, ctypes as C, numpy as np, torch
sys.path.insert(0, '.')
import m17_sdr_amd._lib as L
L.LIB_PATH = L.LIB_PATH.replace("libm17gpu.so", "libm17gpu_stamps.so")
import m17_sdr_amd as m
Cn = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
nblk = int(sys.argv[2]) if len(sys.argv) > 2 else 50
sig = m.generate_batch(min(Cn,128), nblk, n_stream_frames=400)
iq = torch.from_numpy(sig["iq"]).cuda()
if Cn > iq.shape[0]: iq = iq.repeat((Cn+iq.shape[0]-1)//iq.shape[0],1,1,1)[:Cn].contiguous()
rx = m.Receiver(Cn, nblk); out = rx.alloc_outputs(nblk)
for _ in range(3): rx.rx_blocks(iq, 1, out)
torch.cuda.synchronize()
st = (C.c_ulonglong*16)()
m.lib().m17gpu_debug_stamps(st)
names = ["frame pick", "cor + LICH", "chunk gathers issued", "butterflies + soft-bit commit", "traceback", "record out", "task loop"]
tot = sum(st[:7]); ntask = st[8]
print("tasks total", ntask, "; workgroup 0 ticks by phase (all its tasks):")
for i, nme in enumerate(names): print(f"{nme:20s} {st[i]:10d} ticks  {100*st[i]/max(tot,1):5.1f}%")
print("total", tot)
