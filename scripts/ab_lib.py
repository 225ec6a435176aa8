"""Stage times of one build of the library on one workload, for A/B runs on the SAME box (boxes differ by +-10 %):
   python scripts/ab_lib.py LIBNAME C nblk mode [reps]      (LIBNAME e.g. libm17gpu.so / libm17gpu_base.so, in m17_sdr_amd/)
   alternate the builds in one shell loop and compare the medians."""
import sys, os, statistics, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import m17_sdr_amd._lib as L
L.LIB_PATH = os.path.join(os.path.dirname(L.LIB_PATH), sys.argv[1])
import m17_sdr_amd as m
C, nblk, mode = int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
reps = int(sys.argv[5]) if len(sys.argv) > 5 else 3
T = 12
gen = m.Receiver(C, nblk)
big = gen.gen_batch(nblk * T)["iq"]
slabs = big.view(C, T, nblk, 1920, 2).permute(1, 0, 2, 3, 4).contiguous()
del big
gen.close()
rows = []
for rep in range(reps):
    rx = m.Receiver(C, nblk)
    out = rx.alloc_outputs(nblk, want_syms=(mode == 0))
    for k in range(2): rx.rx_blocks(slabs[k], mode, out)
    torch.cuda.synchronize()
    rx.set_profiling(True)
    for k in range(2, T): rx.rx_blocks(slabs[k], mode, out)
    torch.cuda.synchronize()
    ms, n = rx.kernel_ms()
    rows.append(list(ms))
    rx.close()
med = [statistics.median(r[i] for r in rows) for i in range(4)]
print(f"{sys.argv[1]:24s} C={C} nblk={nblk} mode={mode}  fe {med[0]:.4f}  sync {med[1]:.4f}  decode {med[2]:.4f}  book {med[3]:.4f}  sum {sum(med):.4f} ms")
