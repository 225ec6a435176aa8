"""A/B of the FIR-stage variants on one box: fir_impl 1 (front end + timing kernel) against 4 (k_rx_chan6), same input, same
process.  usage: python scripts/ab_fir.py [channels] [blocks] [mode]"""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import m17_sdr_amd as m
C = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
nblk = int(sys.argv[2]) if len(sys.argv) > 2 else 12
mode = int(sys.argv[3]) if len(sys.argv) > 3 else 0
T = 24
gen = m.Receiver(C, nblk)
big = gen.gen_batch(nblk * T, n_stream_frames=40)["iq"]
slabs = torch.empty((T, C, nblk, 1920, 2), dtype=torch.int16, device=big.device)
slabs.copy_(big.view(C, T, nblk, 1920, 2).permute(1, 0, 2, 3, 4))
del big
gen.close()
for rep in range(2):
    for impl in (1, 4):
        rx = m.Receiver(C, nblk)
        rx.set_option("fir_impl", impl)
        out = rx.alloc_outputs(nblk, want_syms=(mode == 0))
        for k in range(4):
            rx.rx_blocks(slabs[k], mode, out)
        torch.cuda.synchronize()
        rx.set_profiling(True)
        t0 = time.perf_counter()
        for k in range(4, T):
            rx.rx_blocks(slabs[k], mode, out)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / (T - 4) * 1e3
        kms, n = rx.kernel_ms()
        print(f"fir_impl={impl}  {C}x{nblk} mode {mode}: {dt:.4f} ms/step wall; stages " + " ".join(f"{x:.4f}" for x in kms), flush=True)
        rx.close()
