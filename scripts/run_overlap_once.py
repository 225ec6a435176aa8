import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import m17_sdr_amd as m
C, nblk, mode = 16384, 12, int(sys.argv[1]) if len(sys.argv) > 1 else 0
rx = m.Receiver(C, nblk)
for kv in sys.argv[2:]:
    k, v = kv.split("="); rx.set_option(k, int(v))
T = 4
big = rx.gen_batch(nblk * T)["iq"]
slabs = big.view(C, T, nblk, 1920, 2).permute(1, 0, 2, 3, 4).contiguous()
out = rx.alloc_outputs(nblk, want_syms=(mode == 0))
for k in range(T): rx.rx_blocks(slabs[k], mode, out)
torch.cuda.synchronize()
