"""Timing stage on squelched input: all-zero IQ (symbols NaN) and a constant carrier (symbols exactly 0.0) against the
normal signal:   python scripts/exp_dead_channels.py [LIBNAME [C nblk]]   (default 16,384 x 12)"""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import m17_sdr_amd._lib as L
if len(sys.argv) > 1: L.LIB_PATH = os.path.join(os.path.dirname(L.LIB_PATH), sys.argv[1])
import m17_sdr_amd as m
C, nblk = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (16384, 12)
gen = m.Receiver(C, nblk)
sig = gen.gen_batch(nblk * 3)["iq"][:, nblk:2 * nblk].contiguous()
gen.close()
g = torch.Generator(device="cuda"); g.manual_seed(1)
noise = torch.randint(-32768, 32768, sig.shape, generator=g, device="cuda", dtype=torch.int32).to(torch.int16)
tiny = torch.randint(-3, 4, sig.shape, generator=g, device="cuda", dtype=torch.int32).to(torch.int16)
alt = torch.empty_like(sig); alt[:, :, 0::2] = 32767; alt[:, :, 1::2] = -32768
half = sig.clone(); half[::2] = 0                                     # every other channel squelched
for name, iq in (("signal", sig), ("all-zero IQ", torch.zeros_like(sig)), ("constant carrier", torch.full_like(sig, 12345)),
                 ("full-scale white noise", noise), ("+-3 LSB noise", tiny), ("alternating extremes", alt), ("every other channel squelched", half)):
    rx = m.Receiver(C, nblk)
    out = rx.alloc_outputs(nblk)
    for k in range(2): rx.rx_blocks(iq, 1, out)
    torch.cuda.synchronize()
    rx.set_profiling(True)
    for k in range(5): rx.rx_blocks(iq, 1, out)
    torch.cuda.synchronize()
    ms, n = rx.kernel_ms()
    print(f"{name:18s}: front end {ms[0]:.4f}  timing+framer {ms[1]:.4f}  decode {ms[2]:.4f}  bookkeeping {ms[3]:.4f} ms")
    rx.close()
