"""EXPERIMENT (instrumented build): what k_rx_chan6 would take if the discriminator rows did not travel through HBM -- its
rows in a compact region of `slots` x 16 rows that is reused by every generation of waves and so can stay in L2 /
Infinity Cache.  Rows of two resident waves may collide: the results are WRONG, only the time is of interest.
   python scripts/exp_rc_slots.py [blocks] [slots ...]"""
import sys, os, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import m17_sdr_amd._lib as L_
L_.LIB_PATH = L_.LIB_PATH.replace("libm17gpu.so", "libm17gpu_stamps.so")
import m17_sdr_amd as m
Cn = 16384
nblk = int(sys.argv[1]) if len(sys.argv) > 1 else 16
slots = [int(x) for x in sys.argv[2:]] or [0, 8192, 6144]
T = 8
gen = m.Receiver(Cn, nblk)
big = gen.gen_batch(nblk * T)["iq"]
slabs = torch.empty((T, Cn, nblk, 1920, 2), dtype=torch.int16, device=big.device)
slabs.copy_(big.view(Cn, T, nblk, 1920, 2).permute(1, 0, 2, 3, 4))
del big
gen.close()
for rep in range(2):
    for ns in slots:
        rx = m.Receiver(Cn, nblk)
        rx.set_option("fir_impl", 4)
        rx.set_option("rc_slots", ns)
        out = rx.alloc_outputs(nblk, want_syms=True)
        t_end = time.perf_counter() + 0.4
        while time.perf_counter() < t_end:
            for k in range(T): rx.rx_blocks(slabs[k], 0, out)
            torch.cuda.synchronize()
        rx.set_profiling(True)
        for k in range(2 * T): rx.rx_blocks(slabs[k % T], 0, out)
        torch.cuda.synchronize()
        ms, call, n = rx.call_ms()
        print(f"rc_slots={ns:6d} ({ns * 16 * 1536 / 1e6:6.1f} MB of rows)  nblk={nblk}  kernel {ms[1]:.4f} ms = {ms[1] * 12 / nblk:.4f} per 12 blocks", flush=True)
        rx.close()
