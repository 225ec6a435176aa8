"""What bounds k_frontend_q: stage time with its input cache-resident (fe_debug 1), without its output stream (2), both (3).
Instrumented build only (make -C m17_sdr_amd/csrc stamps); results of the debug modes are wrong by construction."""
import sys, os, statistics, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import m17_sdr_amd._lib as L
L.LIB_PATH = L.LIB_PATH.replace("libm17gpu.so", "libm17gpu_stamps.so")
import m17_sdr_amd as m
C, nblk, T = int(sys.argv[1]) if len(sys.argv) > 1 else 16384, int(sys.argv[2]) if len(sys.argv) > 2 else 12, 8
gen = m.Receiver(C, nblk)
big = gen.gen_batch(nblk * T)["iq"]
slabs = big.view(C, T, nblk, 1920, 2).permute(1, 0, 2, 3, 4).contiguous()
del big
for rnd in range(2):
    for dbg in (0, 1, 2, 3):
        rx = m.Receiver(C, nblk)
        rx.set_option("fe_debug", dbg)
        rx.set_option("sync_impl", 8)
        out = rx.alloc_outputs(nblk)
        for k in range(2): rx.rx_blocks(slabs[k], 0, out)
        torch.cuda.synchronize()
        rx.set_profiling(True)
        for k in range(2, T): rx.rx_blocks(slabs[k], 0, out)
        torch.cuda.synchronize()
        ms, n = rx.kernel_ms()
        print(f"fe_debug={dbg}  front end {ms[0]:.4f} ms   (timing stage {ms[1]:.4f})", flush=True)
        rx.close()
