"""Run-to-run determinism of the whole chain: the same IQ through fresh contexts several times, every output compared
bit for bit (records, counts, symbol stream, end state).  A data race shows up here before it shows up as a wrong
result:   python scripts/determinism_check.py C nblk [runs]"""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import m17_sdr_amd as m
C, nblk = int(sys.argv[1]), int(sys.argv[2])
runs = int(sys.argv[3]) if len(sys.argv) > 3 else 4
gen = m.Receiver(C, nblk)
iq = gen.gen_batch(2 * nblk, n_stream_frames=10, ebn0_db=10.0, noise_cutoff_hz=6250.0)["iq"]
gen.close()
parts = [iq[:, :nblk].contiguous(), iq[:, nblk:].contiguous()]
ref = None
for r in range(runs):
    rx = m.Receiver(C, nblk)
    got = []
    for part in parts:                                   # two calls: the state carried between them is covered too
        out = rx.rx_blocks(part, 1, rx.alloc_outputs(nblk, want_syms=True))
        torch.cuda.synchronize()
        got += [out["recs"].clone(), out["counts"].clone(), out["syms"].view(torch.int32).clone(), out["nsyms"].clone()]
    got += [torch.from_numpy(rx.lsf().copy()), torch.from_numpy(rx.counters().copy()), torch.from_numpy(rx.lock().copy())]
    rx.close()
    if ref is None: ref = got
    else:
        bad = [i for i, (a, b) in enumerate(zip(ref, got)) if not torch.equal(a.cpu(), b.cpu())]
        print(f"run {r}: {'identical' if not bad else 'DIFFERS in outputs ' + str(bad)}", flush=True)
        if bad: sys.exit(1)
print(f"{runs} runs of {C} channels x 2 x {nblk} blocks: bit-identical")
