"""What a producer / consumer workgroup could reach at best (k_pc_mock, m17_sync_wave.hip): the front end's and the timing
kernel's work of a step in ONE launch, co-resident (12 timing waves + 4 front-end waves per workgroup), no hand-over.
   python scripts/exp_pc_mock.py [channels] [blocks]"""
import sys, os, ctypes as C, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import m17_sdr_amd._lib as L_
L_.LIB_PATH = L_.LIB_PATH.replace("libm17gpu.so", "libm17gpu_stamps.so")     # the experiment lives in the instrumented build only (make stamps)
import m17_sdr_amd as m
Cn = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
nblk = int(sys.argv[2]) if len(sys.argv) > 2 else 12
T = 6
rx = m.Receiver(Cn, nblk)
rx.set_option("sync_impl", 8)
big = rx.gen_batch(nblk * T)["iq"]
slabs = big.view(Cn, T, nblk, 1920, 2).permute(1, 0, 2, 3, 4).contiguous()
del big
out = rx.alloc_outputs(nblk, want_syms=True)
L = m.lib()
L.m17gpu_debug_pc_mock.restype = C.c_int
L.m17gpu_debug_pc_mock.argtypes = [C.c_void_p] * 2 + [C.c_int] + [C.c_void_p] * 3
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
for k in range(300):                                   # clock settled
    rx.rx_blocks(slabs[k % T], 0, out)
torch.cuda.synchronize()
rx.set_profiling(True)
for k in range(24):
    rx.rx_blocks(slabs[k % T], 0, out)
torch.cuda.synchronize()
ms, n = rx.kernel_ms()
print(f"two kernels:  front end {ms[0]:.4f} + timing {ms[1]:.4f} = {ms[0] + ms[1]:.4f} ms")
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for rep in range(3):
    rx.rx_blocks(slabs[0], 0, out)                     # the stream the mock's producers rewrite
    torch.cuda.synchronize()
    e0.record()
    for k in range(20):
        assert L.m17gpu_debug_pc_mock(rx._ctx, slabs[0].data_ptr(), nblk, out["syms"].data_ptr(), out["nsyms"].data_ptr(), st) == 0
    e1.record()
    torch.cuda.synchronize()
    print(f"k_pc_mock (12 timing waves + 4 front-end waves per workgroup, one workgroup per CU): {e0.elapsed_time(e1) / 20:.4f} ms per launch")
