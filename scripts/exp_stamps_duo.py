"""Timing wave and framer wave of k_sync_frame_duo at 1,024 x 50 (instrumented build: make -C m17_sdr_amd/csrc stamps):
work / wait per block, the spread over the channels, and the per-block work of the fastest and slowest of the first 64."""
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
st2 = np.zeros((4096, 8), np.uint64)
m.lib().m17gpu_debug_chan_stamps(st2.ctypes.data_as(C.c_void_p))
allw = st2[:Cn, 5].astype(np.int64); allq = st2[:Cn, 4].astype(np.int64)
print("all %d channels, last call: timing-wave work ticks min %d  median %d  p90 %d  max %d (channel %d); its waits for the framer: median %d  max %d"
      % (Cn, allw.min(), np.median(allw), np.percentile(allw, 90), allw.max(), allw.argmax(), np.median(allq), allq.max()))
per = st2[:, 6].reshape(64, 64)[:, :min(nblk, 64)].astype(np.int64)
flg = st2[:, 7].reshape(64, 64)[:, :min(nblk, 64)].astype(np.int64)
tot = per.sum(1)
for c in (int(tot.argmin()), int(tot.argmax())):
    print(f"channel {c}: ticks per block (c = taken as calm, x = a branch step in it, L = in lock):")
    print("  " + "  ".join(f"{int(per[c, b])}{'c' if flg[c, b] & 1 else ''}{'x' if flg[c, b] & 2 else ''}{'L' if flg[c, b] & 4 else ''}" for b in range(per.shape[1])))
# where the timing waves' time goes, over the first 64 channels: calm blocks, blocks with branch steps in lock, unlocked blocks
calm = (flg & 1) != 0; stepx = (flg & 2) != 0; lock = (flg & 4) != 0
for name, sel in (("calm, no step", calm & ~stepx), ("locked, with steps", lock & stepx), ("unlocked", ~lock)):
    print(f"  {name:20s}: {int(sel.sum()):5d} blocks of {sel.size}, {100.0 * per[sel].sum() / per.sum():5.1f} % of the ticks, {per[sel].mean() if sel.any() else 0:8.0f} ticks per block")
