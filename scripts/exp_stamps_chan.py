import sys, os, ctypes as C, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import m17_sdr_amd._lib as L
L.LIB_PATH = L.LIB_PATH.replace("libm17gpu.so", "libm17gpu_stamps.so")
import m17_sdr_amd as m
Cn, nblk, T = 1024, 50, 4
rx = m.Receiver(Cn, nblk); rx.set_option("sync_impl", 4)
big = rx.gen_batch(nblk * T)["iq"]
slabs = big.view(Cn, T, nblk, 1920, 2).permute(1, 0, 2, 3, 4).contiguous()
out = rx.alloc_outputs(nblk, want_syms=True)
for k in range(T): rx.rx_blocks(slabs[k], 0, out)
torch.cuda.synchronize()
st = np.zeros((4096, 8), np.uint64)
m.lib().m17gpu_debug_chan_stamps(st.ctypes.data_as(C.c_void_p))
st = st[:Cn].astype(np.float64)
tot = st[:, :6].sum(1)
names = ["round top", "FIR(+)", "vote+decide", "prefetch/syms", "framer", "commit", "rounds", "framer iters"]
order = np.argsort(tot)
print("per-channel wave time of the last step (ticks): min %.0f  median %.0f  p90 %.0f  max %.0f" % (tot.min(), np.median(tot), np.percentile(tot, 90), tot.max()))
for label, idx in (("median channel", order[len(order)//2]), ("slowest channel", order[-1]), ("fastest channel", order[0])):
    print(label, int(idx), " ".join(f"{n}={st[idx,i]/nblk:.0f}" for i, n in enumerate(names)))
lock = rx.lock()
print("locked channels at the end:", int(lock.sum()), "of", Cn)
