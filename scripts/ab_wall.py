"""Same-box A/B of option sets by WALL CLOCK: steps back to back on one stream, clock settled, ms per step.
   python scripts/ab_wall.py C nblk mode "name=v,name=v" "name=v" ...   ('-' = defaults)"""
import sys, os, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import m17_sdr_amd as m
C, nblk, mode = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
sets = sys.argv[4:] or ["-"]
T = 8
gen = m.Receiver(C, nblk)
big = gen.gen_batch(nblk * T, ebn0_db=float(os.environ.get("M17_EBN0", "200")), noise_cutoff_hz=6250.0 if "M17_EBN0" in os.environ else 0.0)["iq"]
slabs = torch.empty((T, C, nblk, 1920, 2), dtype=torch.int16, device=big.device)
slabs.copy_(big.view(C, T, nblk, 1920, 2).permute(1, 0, 2, 3, 4))
del big
gen.close()
def run(opt):
    rx = m.Receiver(C, nblk)
    if opt != "-":
        for kv in opt.split(","):
            k, v = kv.split("=")
            rx.set_option(k, int(v))
    out = rx.alloc_outputs(nblk, want_syms=(mode == 0))
    t_end = time.perf_counter() + 0.5
    while time.perf_counter() < t_end:
        for k in range(T): rx.rx_blocks(slabs[k], mode, out)
        torch.cuda.synchronize()
    reps = 6
    t0 = time.perf_counter()
    for _ in range(reps):
        for k in range(T): rx.rx_blocks(slabs[k], mode, out)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / (reps * T) * 1e3
    rx.close()
    return dt
for rep in range(3):
    print("  ".join(f"{opt}: {run(opt):.4f}" for opt in sets) + f"   ms per step, C={C} nblk={nblk} mode={mode}", flush=True)
