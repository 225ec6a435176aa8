"""Randomised differential test: GPU (through the C-ABI) against the CPU oracle on randomly drawn call shapes,
signal conditions and kernel-variant options.  The fixed cases of test_gpu_parity.py cover what the reference
tests and what each kernel's design makes risky; this one covers combinations nobody thought of.

The default run is a FIXED list of trials -- M17_FUZZ_TRIALS (default 60) trials of sequence M17_FUZZ_SEED (default 5),
trial k drawn from its own generator seeded (seed, k), plus two forced large-batch trials (the sizes at which the
library picks k_rx_chan6 with its shared last tiles and the lane-per-channel bookkeeping) --
so it is the same on every box whatever its speed, and a failure names the trial: M17_FUZZ_ONLY=k re-runs trial k alone.
M17_FUZZ_SECONDS > 0 turns it into a time-bounded experiment that keeps drawing past the fixed list
(profiles/r05_fuzz_parity.txt, r06_fuzz_parity.txt); a safety cap of 300 s ends a default run that got too slow."""
import os
import time

import numpy as np
import pytest

from tests.test_gpu_parity import _compare_raw, _rx_compare

pytestmark = pytest.mark.gpu

_CHANNELS = [1, 2, 3, 7, 17, 63, 64, 65, 100, 257, 640, 1000, 1024, 1025, 2500]
_OPTION_VALUES = {"fe_impl": [0, 2, 3, 4], "fir_impl": [0, 1, 4, 5], "sync_impl": [0, 6, 8],
                  "slot_impl": [0, 1, 2], "book_impl": [0, 1, 2]}
_FORCED = [dict(C=10003, nblk=14, mode=1, ebn0=9.0, nsf=5, packet_mode=0, calls=2, seed=0x4D313761, options={}),
           dict(C=12001, nblk=33, mode=0, ebn0=6.0, nsf=11, packet_mode=0, calls=1, seed=0x4D313762, options={"book_impl": 2})]


def _draw(rng):
    C = int(rng.choice(_CHANNELS))
    nblk = int(rng.choice([1, 2, 5, 12, 15, 16, 17, 31, 32, 33, 48])) if rng.random() < 0.7 else int(rng.integers(1, 49))
    if C * nblk > 40000:                                   # keeps a trial (generator + oracle on the host) near a second
        nblk = max(1, 40000 // C)
    if rng.random() < 0.02:                                # now and then a batch of the size the wave-per-channel kernels are chosen for
        C = int(rng.choice([8192, 10000, 10240, 12001]))
        nblk = int(rng.choice([12, 14, 16, 27, 32]))
    mode = int(rng.random() < 0.75)
    calls = int(rng.choice([1, 1, 2, 3]))
    ebn0 = float(rng.choice([200.0, 15.0, 9.0, 6.0, 4.0]))
    packet_mode = int(mode == 1 and rng.random() < 0.3)
    nsf = int(rng.integers(1, 25))
    options = {k: int(rng.choice(v)) for k, v in _OPTION_VALUES.items() if rng.random() < 0.4}
    return dict(C=C, nblk=nblk, mode=mode, ebn0=ebn0, nsf=nsf, packet_mode=packet_mode, calls=calls,
                seed=int(rng.integers(1, 2**31)), options=options)


def _mutilate(rng, iq):
    """What a radio does to a signal: squelch gaps, clipping, dead and noise-only channels, isolated zero samples."""
    C, nblk = iq.shape[0], iq.shape[1]
    for _ in range(int(rng.integers(1, 6))):
        c = int(rng.integers(0, C))
        kind = int(rng.integers(0, 6))
        b0 = int(rng.integers(0, nblk)); b1 = int(rng.integers(b0, nblk)) + 1
        if kind == 0: iq[c, b0:b1] = 0
        elif kind == 1: iq[c] = rng.integers(-32768, 32768, size=iq[c].shape, dtype=np.int16)
        elif kind == 2: iq[c, b0:b1] = np.clip(iq[c, b0:b1].astype(np.int32) * 8, -32768, 32767).astype(np.int16)
        elif kind == 3: iq[c, :, ::int(rng.integers(2, 11))] = 0
        elif kind == 4: iq[c, b0:b1] = int(rng.integers(-32768, 32768))
        else: iq[c, b0, int(rng.integers(0, 1000)):int(rng.integers(1000, 1900))] = 0
    return iq


def _trial(k, seed):
    """Trial k of sequence `seed`: parameters and the generator its input mutilation draws from."""
    rng = np.random.default_rng([seed, k])
    if k < len(_FORCED):
        return dict(_FORCED[k]), False, rng
    return _draw(rng), bool(rng.random() < 0.3), rng


def test_random_shapes_conditions_and_variants_are_bit_exact():
    import m17_sdr_amd as m
    seed = int(os.environ.get("M17_FUZZ_SEED", "5"))
    n_fixed = int(os.environ.get("M17_FUZZ_TRIALS", "60"))
    seconds = float(os.environ.get("M17_FUZZ_SECONDS", "0"))
    only = os.environ.get("M17_FUZZ_ONLY")
    t0, trials, raw = time.time(), 0, 0
    k = int(only) if only is not None else 0
    while True:
        if only is None:
            if seconds > 0:
                if time.time() - t0 >= seconds and k >= len(_FORCED):
                    break
            elif k >= n_fixed or time.time() - t0 > 300.0:
                break
        p, mutilated, rng = _trial(k, seed)
        try:
            if mutilated:
                sig = m.generate_batch(p["C"], p["nblk"], n_stream_frames=p["nsf"], ebn0_db=p["ebn0"],
                                       packet_mode=p["packet_mode"], base_seed=p["seed"])
                _compare_raw(np.ascontiguousarray(_mutilate(rng, sig["iq"].copy())), p["mode"], options=p["options"])
                raw += 1
            else:
                _rx_compare(**p)
        except AssertionError as e:
            raise AssertionError(f"trial {k} of sequence {seed} (re-run: M17_FUZZ_SEED={seed} M17_FUZZ_ONLY={k}) "
                                 f"mutilated={mutilated} {p}: {str(e)[:2000]}") from None
        trials += 1
        k += 1
        if only is not None:
            break
        if trials % 100 == 0:                              # a long run must be seen to be alive
            print(f"fuzz: {trials} trials, {time.time() - t0:.0f} s", flush=True)
    print(f"fuzz: {trials} trials ({raw} on mutilated input) in {time.time() - t0:.1f} s, sequence {seed}")
    assert trials >= (1 if only is not None else min(n_fixed, 3))
