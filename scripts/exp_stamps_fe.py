"""Phase shares of k_frontend_q (one wave, workgroup 777) at the headline size: instrumented build (make stamps)."""
import sys, os, ctypes as C, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import m17_sdr_amd._lib as L
L.LIB_PATH = L.LIB_PATH.replace("libm17gpu.so", "libm17gpu_stamps.so")
import m17_sdr_amd as m
Cn, nblk = 16384, 12
rx = m.Receiver(Cn, nblk)
iq = rx.gen_batch(nblk)["iq"]
out = rx.alloc_outputs(nblk)
for _ in range(3): rx.rx_blocks(iq, 0, out)
torch.cuda.synchronize()
st = (C.c_ulonglong * 8)()
m.lib().m17gpu_debug_fe_stamps(st)
names = ["tile write, next loads issued, own samples read", "convert / limit / discriminate 16 samples, u to LDS", "DC chain + picks (lane 0 of the quad)", "-", "-", "between chunks (output rows every 5th)"]
tot = sum(st[:6])
for i, n in enumerate(names):
    if st[i]: print(f"{n:55s} {st[i] / 30:9.0f} ticks per chunk  {100 * st[i] / tot:5.1f} %")
print(f"wave total {tot} ticks for 30 chunks of 16 rows")

import numpy as np
sp = np.zeros((16384, 2), np.uint64)
m.lib().m17gpu_debug_fe_span(sp.ctypes.data_as(C.c_void_p))
sp = sp[:12288].astype(np.float64)
dur = sp[:, 1] - sp[:, 0]
span = sp[:, 1].max() - sp[:, 0].min()
print(f"12,288 front-end waves: lifetime median {np.median(dur):.0f} ticks (p10 {np.percentile(dur,10):.0f}, p90 {np.percentile(dur,90):.0f}); "
      f"kernel span {span:.0f} ticks; average resident waves {dur.sum()/span:.0f} = {dur.sum()/span/1024:.2f} per SIMD")
