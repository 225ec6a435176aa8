"""Does the front end's time depend on WHERE the context's buffers lie?  (round 5: 0.320 / 0.354 ms alternating from one
Receiver to the next on the same box and input.)  Instrumented build (make stamps): prints the buffer addresses next to the
kernel times of each of a series of contexts, with dummy allocations of varying size in between.
   python scripts/exp_alloc_placement.py [channels] [blocks]"""
import sys, os, ctypes as C, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import m17_sdr_amd._lib as L_
L_.LIB_PATH = L_.LIB_PATH.replace("libm17gpu.so", "libm17gpu_stamps.so")
import m17_sdr_amd as m
Cn = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
nblk = int(sys.argv[2]) if len(sys.argv) > 2 else 12
T = 8
gen = m.Receiver(Cn, nblk)
big = gen.gen_batch(nblk * T)["iq"]
slabs = torch.empty((T, Cn, nblk, 1920, 2), dtype=torch.int16, device=big.device)
slabs.copy_(big.view(Cn, T, nblk, 1920, 2).permute(1, 0, 2, 3, 4))
del big
gen.close()
lib = m.lib()
keep = []
print("slabs at 0x%x" % slabs.data_ptr())
for trial in range(14):
    if trial >= 4:
        keep.append(torch.empty((1 << 20) * (1 + 3 * (trial % 5)) + 4096 * trial, dtype=torch.uint8, device="cuda"))   # shift what comes next
    rx = m.Receiver(Cn, nblk)
    out = rx.alloc_outputs(nblk, want_syms=True)
    t_end = time.perf_counter() + 0.3
    while time.perf_counter() < t_end:
        for k in range(T): rx.rx_blocks(slabs[k], 0, out)
        torch.cuda.synchronize()
    rx.set_profiling(True)
    for k in range(2 * T): rx.rx_blocks(slabs[k % T], 0, out)
    torch.cuda.synchronize()
    ms, n = rx.kernel_ms()
    ptrs = (C.c_ulonglong * 6)()
    lib.m17gpu_debug_ptrs(rx._ctx, ptrs)
    print("trial %2d  fe %.4f  sync %.4f   state %x disc %x offs %x fsym %x | syms %x nsyms %x" %
          (trial, ms[0], ms[1], ptrs[0], ptrs[1], ptrs[2], ptrs[3], out["syms"].data_ptr(), out["nsyms"].data_ptr()), flush=True)
    rx.close()
    del out
