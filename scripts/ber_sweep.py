#!/usr/bin/env python3
"""BASELINE config #4: C channels, AWGN sweep Eb/N0 0..10 dB, BER/FER of the GPU chain vs
the CPU oracle on the same IQ.  Same input => the two curves must coincide exactly;
the script also reports payload BER against the transmitted truth.

    python scripts/ber_sweep.py --channels 16384 --blocks 30 --out profiles/r02_ber_sweep_16384ch.json
Axis: the generator's `ebn0_db` is energy per CHANNEL bit over N0: Es/N0 - 3 dB (two channel bits per
4-FSK symbol), Es = A^2 x 10 samples, N0 = noise PSD (complex variance per 48 kHz sample before the
channel filter).  Per INFORMATION bit of the stream payload the code rate comes on top (SURVEY 8d,
Es = 2 Eb R): R = 144/272 for the P2-punctured K=5 code, i.e. Eb_info/N0 = axis + 2.76 dB; both are
written to the JSON.  EVERY channel of every point goes through the CPU oracle as well.  The reference has no software channel filter ahead of
its limiter, so the noise is band-limited here (default 6.25 kHz one-sided = a 12.5 kHz
channel) like a radio front end would; with white 48 kHz noise nothing decodes below
~17 dB (FM threshold)."""
import argparse, json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def measure(recs, counts, sig, m):
    """payload bit errors / bits over parsed stream frames with a plausible FN, frames seen, frames sent"""
    bit_err = bits = frames = 0
    for c in range(recs.shape[0]):
        r = recs[c, :counts[c]]
        sel = (r["type"] == 2) & ((r["flags"] & m.F_PARSED) != 0) & (r["fn"] < sig["nframes"][c])
        for x in r[sel]:
            want = sig["payload"][c, x["fn"]]
            got = x["data"][8:24]
            bit_err += int(np.unpackbits(np.bitwise_xor(want, got)).sum())
            bits += 128
            frames += 1
    return bit_err, bits, frames, int(sig["nframes"].sum())


def lock_stats(recs, counts, m):
    """What the framer did: lock events, losses, and how many channels ever locked."""
    cap = recs.shape[1]
    valid = np.arange(cap)[None, :] < np.minimum(counts, cap)[:, None]
    fl = recs["flags"]
    aos = valid & ((fl & m.F_AOS) != 0)
    lost = valid & ((fl & m.F_LOST) != 0)
    eot = valid & ((fl & m.F_EOT) != 0)
    parsed = valid & ((fl & m.F_PARSED) != 0)
    return {"aos": int(aos.sum()), "lost": int(lost.sum()), "eot": int(eot.sum()), "parsed_any_type": int(parsed.sum()),
            "channels_locked_ever": int(aos.any(axis=1).sum()),
            "first_lock_block_mean": float(np.where(aos.any(axis=1), recs["block"][np.arange(recs.shape[0]), aos.argmax(axis=1)], 0)[aos.any(axis=1)].mean())
            if aos.any() else None}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--channels", type=int, default=16384)
    ap.add_argument("--blocks", type=int, default=30)
    ap.add_argument("--ebn0", type=float, nargs="*", default=[float(x) for x in range(0, 21, 1)] + [200.0])
    ap.add_argument("--min-bits", type=float, default=1e6, help="payload bits wanted per point with FER < 0.99 (SURVEY 8d)")
    ap.add_argument("--max-passes", type=int, default=8, help="passes of --channels fresh channels per point at most")
    ap.add_argument("--noise-cutoff", type=float, default=6250.0, help="one-sided channel-filter cutoff applied to the noise, Hz (0 = white over 48 kHz)")
    ap.add_argument("--out", default="")
    a = ap.parse_args()
    import torch
    import m17_sdr_amd as m
    from tests import oracle
    rows = []
    n = a.channels
    for eb in a.ebn0:
        t0 = time.time()
        tot = dict(bit_err=0, bits=0, frames=0, sent=0, identical=True, checked=0)
        lk = None
        passes = 0
        while passes < a.max_passes:
            rx = m.Receiver(n, a.blocks)
            # fresh channels every pass: seeds continue where the last pass stopped
            dsig = rx.gen_batch(a.blocks, n_stream_frames=a.blocks - 6, ebn0_db=eb, first_channel=passes * n,
                                noise_cutoff_hz=a.noise_cutoff)
            sig = {"payload": dsig["payload"].cpu().numpy(), "nframes": dsig["nframes"].cpu().numpy(),
                   "iq": dsig["iq"].cpu().numpy()}
            out = rx.rx_blocks(dsig["iq"], 1, rx.alloc_outputs(a.blocks))
            torch.cuda.synchronize()
            recs = out["recs"].cpu().numpy().view(oracle.REC_DTYPE).reshape(n, -1)
            counts = out["counts"].cpu().numpy()
            be, b, f, s_ = measure(recs, counts, sig, m)
            tot["bit_err"] += be; tot["bits"] += b; tot["frames"] += f; tot["sent"] += s_
            st = lock_stats(recs, counts, m)
            lk = st if lk is None else {k: (lk[k] + st[k] if isinstance(st[k], int) else st[k]) for k in st}
            # EVERY channel of every pass also goes through the CPU oracle
            ref = oracle.Channels(n).rx_blocks(sig["iq"], mode=1, want_syms=False, nthreads=16)
            same = np.array_equal(ref["counts"], counts) and all(
                ref["recs"][c, :counts[c]].tobytes() == recs[c, :counts[c]].tobytes() for c in range(n))
            tot["identical"] = tot["identical"] and bool(same)
            tot["checked"] += n
            rx.close()
            passes += 1
            fer = 1.0 - tot["frames"] / max(1, tot["sent"])
            if tot["bits"] >= a.min_bits or fer >= 0.99:
                break
        fer = 1.0 - tot["frames"] / max(1, tot["sent"])
        row = {"ebn0_db": eb if eb < 100 else "noiseless", "ebn0_info_bit_db": round(eb + 10 * np.log10(272 / 144), 2) if eb < 100 else None,
               "passes": passes, "channels_total": passes * n,
               "payload_bits": tot["bits"], "bit_errors": tot["bit_err"],
               "status": "no_lock" if tot["frames"] == 0 else ("ok" if tot["bits"] >= a.min_bits else "few_frames (FER >= 0.99: no further passes)"),
               "ber": (tot["bit_err"] / tot["bits"]) if tot["bits"] else None,
               "frames_decoded": tot["frames"], "frames_sent": tot["sent"], "fer": fer,
               "framer": lk,
               "oracle_channels_compared": tot["checked"], "gpu_equals_oracle": tot["identical"],
               "seconds": round(time.time() - t0, 1)}
        rows.append(row)
        print(json.dumps(row), flush=True)
    if a.out:
        json.dump({"channels": a.channels, "blocks": a.blocks, "noise_cutoff_hz": a.noise_cutoff, "signal_source": "gpu",
                   "axis": "ebn0_db = energy per CHANNEL bit / N0 = Es/N0 - 3.01 dB (2 channel bits per 4-FSK symbol; Es = A^2 x 10 "
                           "samples; N0 = complex noise variance per 48 kHz sample before the 12.5 kHz channel filter); "
                           "ebn0_info_bit_db adds the code-rate term of SURVEY 8(d), R = 144/272 (P2-punctured K=5): +2.76 dB",
                   "oracle": "every channel of every pass also decoded by the CPU oracle; gpu_equals_oracle = all records identical",
                   "sizing": f"passes of {a.channels} fresh channels x {a.blocks} blocks are added until a point holds {a.min_bits:.0e} payload bits, "
                             f"its FER is >= 0.99, or {a.max_passes} passes are done; status no_lock = not one stream frame was parsed",
                   "why_few_frames_at_low_ebn0": "FER counts every transmitted stream frame.  The reference's framer locks only on an 8-symbol "
                       "window with NO sign error against a sync template and an amplitude spread (max|v|-min|v|)/max|v| below 0.3 "
                       "(m17_unlocked_sync_check, m17_rx_frame.cpp:82-92), after the timing loop has settled at threshold 10; at 8 dB "
                       "band-limited noise most channels never pass that test during the 24 transmitted frames (framer.channels_locked_ever) "
                       "and a channel that locks late has lost the frames before (framer.first_lock_block_mean); frames inside a held lock "
                       "decode with the BER shown.  The noiseless point is the reference's own start-up loss: carrier, two preambles and "
                       "the link setup frame pass before the first stream frame can complete, and delivery waits for a CRC-good LICH.",
                   "noiseless_residual": "the noiseless point's BER is not zero: the reference's framer now and then classifies the EOT / "
                       "carrier transition as one more stream frame (frame number below the count sent, payload garbage); GPU and oracle "
                       "agree on every such record",
                   "points": rows}, open(a.out, "w"), indent=1)


if __name__ == "__main__":
    main()
