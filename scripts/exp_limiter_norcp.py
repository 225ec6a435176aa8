"""Round-6 review item 6: the limiter's reciprocal without v_rcp_f32 (experiment build `make -C m17_sdr_amd/csrc norcp`:
three Newton steps from the v_rsq_f32 value).  Exhaustive self test of that build (all 2^32 int16 pairs against the literal
expressions) and its stage times:   python scripts/exp_limiter_norcp.py LIBNAME [channels] [blocks]"""
import sys, os, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import m17_sdr_amd._lib as L
L.LIB_PATH = os.path.join(os.path.dirname(L.LIB_PATH), sys.argv[1])
import m17_sdr_amd as m
C = int(sys.argv[2]) if len(sys.argv) > 2 else 16384
nblk = int(sys.argv[3]) if len(sys.argv) > 3 else 16
rx = m.Receiver(C, nblk)
print(sys.argv[1], "selftest mismatches {scale, sqrt, reciprocal (rcp_rn_normal alone), composed limiter}:", rx.selftest(), flush=True)
T = 8
big = rx.gen_batch(nblk * T)["iq"]
slabs = torch.empty((T, C, nblk, 1920, 2), dtype=torch.int16, device=big.device)
slabs.copy_(big.view(C, T, nblk, 1920, 2).permute(1, 0, 2, 3, 4))
del big
out = rx.alloc_outputs(nblk, want_syms=True)
t_end = time.perf_counter() + 0.5
while time.perf_counter() < t_end:
    for k in range(T): rx.rx_blocks(slabs[k], 0, out)
    torch.cuda.synchronize()
for rep in range(3):
    rx.set_profiling(True)
    t0 = time.perf_counter()
    for _ in range(4):
        for k in range(T): rx.rx_blocks(slabs[k], 0, out)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / (4 * T) * 1e3
    ms, n = rx.kernel_ms()
    print(f"{sys.argv[1]:24s} C={C} nblk={nblk} mode 0: wall {dt:.4f} ms per call, FIR-stage kernel {ms[0] + ms[1]:.4f} ms", flush=True)
rx.close()
