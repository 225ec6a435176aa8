"""Step time with and without the look-ahead front end (m17gpu_rx_blocks_ahead), same process:
   python scripts/ab_ahead.py C nblk mode"""
import sys, os, statistics, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import m17_sdr_amd as m
C, nblk, mode = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
T = 14
gen = m.Receiver(C, nblk)
big = gen.gen_batch(nblk * T)["iq"]
slabs = big.view(C, T, nblk, 1920, 2).permute(1, 0, 2, 3, 4).contiguous()
del big; gen.close()
for rnd in range(2):
    for ahead in (0, 1):
        rows = []
        for rep in range(3):
            rx = m.Receiver(C, nblk)
            out = rx.alloc_outputs(nblk, want_syms=(mode == 0))
            for k in range(2): rx.rx_blocks(slabs[k], mode, out)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for k in range(2, T):
                rx.rx_blocks(slabs[k], mode, out)
                if ahead and k + 1 < T: rx.rx_blocks_ahead(slabs[k + 1])
            e1.record(); torch.cuda.synchronize()
            rows.append(e0.elapsed_time(e1) / (T - 2)); rx.close()
        print(f"look-ahead {ahead}: C={C} nblk={nblk} mode={mode}  step {statistics.median(rows):.4f} ms (min {min(rows):.4f})", flush=True)
