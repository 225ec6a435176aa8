"""Stage times of option settings on one workload, same process / same box:
   python scripts/ab_opts.py C nblk mode name=value[,name=value] [more settings ...]   (e.g. lanes_per_channel=64)"""
import sys, os, statistics, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import m17_sdr_amd as m
C, nblk, mode = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
settings = sys.argv[4:] or [""]
T = 12
gen = m.Receiver(C, nblk)
big = gen.gen_batch(nblk * T)["iq"]
slabs = big.view(C, T, nblk, 1920, 2).permute(1, 0, 2, 3, 4).contiguous()
del big
gen.close()
for rnd in range(2):
    for st in settings:
        rows = []
        for rep in range(3):
            rx = m.Receiver(C, nblk)
            for kv in filter(None, st.split(",")):
                k, v = kv.split("="); rx.set_option(k, int(v))
            out = rx.alloc_outputs(nblk, want_syms=(mode == 0))
            for k in range(2): rx.rx_blocks(slabs[k], mode, out)
            torch.cuda.synchronize()
            rx.set_profiling(True)
            for k in range(2, T): rx.rx_blocks(slabs[k], mode, out)
            torch.cuda.synchronize()
            ms, n = rx.kernel_ms()
            rows.append(list(ms)); rx.close()
        med = [statistics.median(r[i] for r in rows) for i in range(4)]
        print(f"{st or 'default':28s} C={C} nblk={nblk} mode={mode}  fe {med[0]:.4f}  sync {med[1]:.4f}  decode {med[2]:.4f}  book {med[3]:.4f}  sum {sum(med):.4f} ms", flush=True)
