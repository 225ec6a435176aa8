"""Per-call stage times over a run of consecutive calls on one continuous stream (the transmissions' carrier /
preamble / end-of-transmission blocks fall into some calls and not others):
   python scripts/step_times.py [C] [nblk] [calls] [ebn0]"""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import m17_sdr_amd as m
C = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
nblk = int(sys.argv[2]) if len(sys.argv) > 2 else 12
T = int(sys.argv[3]) if len(sys.argv) > 3 else 24
ebn0 = float(sys.argv[4]) if len(sys.argv) > 4 else 200.0
rx = m.Receiver(C, nblk)
big = rx.gen_batch(nblk * T, ebn0_db=ebn0, noise_cutoff_hz=6250.0 if ebn0 < 100 else 0.0)["iq"]
slabs = big.view(C, T, nblk, 1920, 2).permute(1, 0, 2, 3, 4).contiguous()
del big
out = rx.alloc_outputs(nblk)
rows = []
for k in range(T):
    rx.set_profiling(True)
    rx.rx_blocks(slabs[k], 1, out)
    torch.cuda.synchronize()
    ms, n = rx.kernel_ms()
    rx.set_profiling(False)
    rows.append(ms)
print("call  front end  timing+framer  decode  bookkeeping (ms)")
for k, r in enumerate(rows):
    print(f"{k:4d}  {r[0]:.4f}  {r[1]:.4f}  {r[2]:.4f}  {r[3]:.4f}")
print("timing+framer: min %.4f  max %.4f" % (min(r[1] for r in rows), max(r[1] for r in rows)))
