"""Same-box A/B of option sets by the duration of the WHOLE call (HIP events around m17gpu_rx_blocks):
   python scripts/ab_call.py C nblk mode "name=v,name=v" "-" ..."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import m17_sdr_amd as m
C, nblk, mode = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
sets = sys.argv[4:] or ["-"]
T = 20
gen = m.Receiver(C, nblk)
big = gen.gen_batch(nblk * T)["iq"]
slabs = torch.empty((T, C, nblk, 1920, 2), dtype=torch.int16, device=big.device)
slabs.copy_(big.view(C, T, nblk, 1920, 2).permute(1, 0, 2, 3, 4))
del big
gen.close()
for rep in range(3):
    for opt in sets:
        rx = m.Receiver(C, nblk)
        if opt != "-":
            for kv in opt.split(","):
                k, v = kv.split("=")
                rx.set_option(k, int(v))
        out = rx.alloc_outputs(nblk, want_syms=(mode == 0))
        for k in range(4): rx.rx_blocks(slabs[k], mode, out)
        torch.cuda.synchronize()
        rx.set_profiling(True)
        for k in range(4, T): rx.rx_blocks(slabs[k], mode, out)
        torch.cuda.synchronize()
        ms, call, n = rx.call_ms()
        print(f"{opt:28s} C={C} nblk={nblk} mode={mode}  call {call:.4f} ms   stages " + " ".join(f"{x:.4f}" for x in ms), flush=True)
        rx.close()
