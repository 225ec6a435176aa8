"""Upper bound for overlapping the front end with the sync stage: the two kernels on two
streams on independent data (front end of one receiver, sync of another) vs back to back."""
import sys, time, torch
sys.path.insert(0, '.')
import m17_sdr_amd as m
Cn = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
nblk = int(sys.argv[2]) if len(sys.argv) > 2 else 50
sig = m.generate_batch(min(Cn, 128), nblk, n_stream_frames=400)
iq = torch.from_numpy(sig["iq"]).cuda()
if Cn > iq.shape[0]: iq = iq.repeat((Cn + iq.shape[0] - 1) // iq.shape[0], 1, 1, 1)[:Cn].contiguous()
rxA, rxB = m.Receiver(Cn, nblk), m.Receiver(Cn, nblk)
outB = rxB.alloc_outputs(nblk, want_syms=True)
disc, offs = rxB.frontend(iq)
torch.cuda.synchronize()
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
def run(concurrent, n=20):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        if concurrent:
            with torch.cuda.stream(s1): rxA.frontend(iq)
            with torch.cuda.stream(s2): rxB.sync_frame(disc, outB)
        else:
            with torch.cuda.stream(s1):
                rxA.frontend(iq); rxB.sync_frame(disc, outB)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3
def only(which, n=20):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        with torch.cuda.stream(s1):
            if which == 0: rxA.frontend(iq)
            else: rxB.sync_frame(disc, outB)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3
for _ in range(2): run(False, 3); run(True, 3)
print(f"C={Cn} nblk={nblk}: frontend only {only(0):.3f} ms, sync only {only(1):.3f} ms, back to back {run(False):.3f} ms, two streams {run(True):.3f} ms")
