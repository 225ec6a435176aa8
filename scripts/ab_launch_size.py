"""Stage times against blocks per launch, clock settled (bench.py's settle): the same continuous stream per channel cut
into calls of nblk blocks.  Per option set: kernel times of a call, and the same scaled to 12 blocks.
   [M17_LIB=libm17gpu_prev.so] [M17_EBN0=8] python scripts/ab_launch_size.py C mode "nblk,nblk,..." ["name=v,name=v" | "-"] ..."""
import sys, os, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import m17_sdr_amd._lib as L_
if os.environ.get("M17_LIB"):                      # another build of the library (same-box A/B of two builds)
    L_.LIB_PATH = os.path.join(os.path.dirname(L_.LIB_PATH), os.environ["M17_LIB"])
import m17_sdr_amd as m
C, mode = int(sys.argv[1]), int(sys.argv[2])
sizes = [int(x) for x in sys.argv[3].split(",")]
sets = sys.argv[4:] or ["-"]
BYTES = 8448 if mode == 0 else 7744
for nblk in sizes:
    T = max(4, min(12, 288 // nblk))
    gen = m.Receiver(C, nblk)
    eb = float(os.environ.get("M17_EBN0", "200"))
    big = gen.gen_batch(nblk * T, n_stream_frames=40, ebn0_db=eb, noise_cutoff_hz=6250.0 if eb < 100 else 0.0)["iq"]
    slabs = torch.empty((T, C, nblk, 1920, 2), dtype=torch.int16, device=big.device)
    slabs.copy_(big.view(C, T, nblk, 1920, 2).permute(1, 0, 2, 3, 4))
    del big
    gen.close()
    torch.cuda.empty_cache()
    for rep in range(2):
        for opt in sets:
            rx = m.Receiver(C, nblk)
            if opt != "-":
                for kv in opt.split(","):
                    k, v = kv.split("=")
                    rx.set_option(k, int(v))
            out = rx.alloc_outputs(nblk, want_syms=(mode == 0))
            t_end, calls = time.perf_counter() + 0.5, 0
            while time.perf_counter() < t_end:
                for k in range(T): rx.rx_blocks(slabs[k], mode, out)
                torch.cuda.synchronize()
                calls += T
            rx.set_profiling(True)
            for k in range(2 * T): rx.rx_blocks(slabs[k % T], mode, out)
            torch.cuda.synchronize()
            ms, call, n = rx.call_ms()
            ksum = sum(ms)
            frac = BYTES * C * nblk / (ksum * 1e-3) / 8e12
            print(f"{opt:24s} C={C} nblk={nblk:3d} mode={mode} settle={calls:4d}  call {call:.4f} ms  stages " +
                  " ".join(f"{x:.4f}" for x in ms) + f"  | per 12 blocks: sum {ksum * 12 / nblk:.4f} stages " +
                  " ".join(f"{x * 12 / nblk:.4f}" for x in ms) + f"  frac {frac:.4f}", flush=True)
            rx.close()
    del slabs
    torch.cuda.empty_cache()
