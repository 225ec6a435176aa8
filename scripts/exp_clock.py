"""Shader clock while the timing kernel runs (instrumented build: make -C m17_sdr_amd/csrc stamps): s_memtime ticks per us of
s_memrealtime over each wave's life, after N back-to-back calls -- with the front end in front of it (the normal step) and
with the timing kernel alone on a discriminator stream made once.   python scripts/exp_clock.py [channels] [calls]"""
import sys, os, ctypes as C, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import m17_sdr_amd._lib as L
L.LIB_PATH = L.LIB_PATH.replace("libm17gpu.so", "libm17gpu_stamps.so")
import m17_sdr_amd as m
Cn = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
N = int(sys.argv[2]) if len(sys.argv) > 2 else 200
nblk, T = 12, 6
rx = m.Receiver(Cn, nblk)
rx.set_option("sync_impl", 8)
big = rx.gen_batch(nblk * T)["iq"]
slabs = big.view(Cn, T, nblk, 1920, 2).permute(1, 0, 2, 3, 4).contiguous()
del big
out = rx.alloc_outputs(nblk, want_syms=True)

def clock(label):
    sp = np.zeros((16384, 3), np.uint64)
    m.lib().m17gpu_debug_wave_span(sp.ctypes.data_as(C.c_void_p))
    sp = sp[:min(16384, Cn)]
    life = (sp[:, 1].astype(np.float64) - sp[:, 0].astype(np.float64)) / 100.0
    tk = (sp[:, 2] >> np.uint64(32)).astype(np.float64)
    mhz = tk / life
    span = (sp[:, 1].max() - sp[:, 0].min()) / 100.0
    print(f"{label:58s} clock {mhz.mean():6.0f} MHz (p10 {np.percentile(mhz, 10):.0f}, p90 {np.percentile(mhz, 90):.0f}); wave life {life.mean():6.1f} us = {tk.mean():8.0f} ticks; kernel span {span:6.1f} us", flush=True)

for n in (3, N):
    for k in range(n):
        rx.rx_blocks(slabs[k % T], 0, out)
    torch.cuda.synchronize()
    clock(f"front end + timing kernel, after {n} back-to-back calls:")
disc, offs = rx.frontend(slabs[0])
torch.cuda.synchronize()
for n in (3, N):
    for k in range(n):
        rx.sync_frame(disc, out)
    torch.cuda.synchronize()
    clock(f"timing kernel alone (same disc stream), after {n} calls:")
import time
time.sleep(0.5)
rx.sync_frame(disc, out)
torch.cuda.synchronize()
clock("timing kernel alone, one call after 0.5 s of idle:")
