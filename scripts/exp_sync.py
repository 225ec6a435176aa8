import sys, time, numpy as np, torch
sys.path.insert(0, '.')
import m17_sdr_amd as m
def run(C, nblk, syms=True, reps=5, mode=0):
    sig = m.generate_batch(min(C,128), nblk, n_stream_frames=40)
    iq = torch.from_numpy(sig["iq"]).cuda()
    if C > iq.shape[0]:
        iq = iq.repeat((C + iq.shape[0]-1)//iq.shape[0], 1, 1, 1)[:C].contiguous()
    rx = m.Receiver(C, nblk)
    out = rx.alloc_outputs(nblk, want_syms=syms)
    for _ in range(2): rx.rx_blocks(iq, mode, out)
    torch.cuda.synchronize()
    rx.set_profiling(True)
    for _ in range(reps): rx.rx_blocks(iq, mode, out)
    torch.cuda.synchronize()
    ms, n = rx.kernel_ms()
    rx.close()
    return ms[0], ms[1]
import ctypes as C
def run2(Cn, nblk, recs=True, syms=True, counts=True, reps=5, ebn0=200.0, nsf=40):
    sig = m.generate_batch(min(Cn,128), nblk, n_stream_frames=nsf, ebn0_db=ebn0)
    iq = torch.from_numpy(sig["iq"]).cuda()
    if Cn > iq.shape[0]:
        iq = iq.repeat((Cn + iq.shape[0]-1)//iq.shape[0], 1, 1, 1)[:Cn].contiguous()
    rx = m.Receiver(Cn, nblk)
    out = rx.alloc_outputs(nblk, want_syms=syms)
    if not recs: out["recs"] = None
    for _ in range(2): rx.rx_blocks(iq, 0, out)
    torch.cuda.synchronize()
    rx.set_profiling(True)
    for _ in range(reps): rx.rx_blocks(iq, 0, out)
    torch.cuda.synchronize()
    ms, n = rx.kernel_ms()
    rx.close()
    return ms[1]*1e3
print("full            ", run2(1024,50))
print("no recs         ", run2(1024,50,recs=False))
print("no recs no syms ", run2(1024,50,recs=False,syms=False))
print("nsf=400 (no gaps)", run2(1024,50,nsf=400))
print("nsf=400 C=256    ", run2(256,50,nsf=400))
