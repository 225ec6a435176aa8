"""The signal source against an ORACLE-side transmitter: oracle/m17_oracle.c restates the reference's
frame builders (m17_tx_routines.cpp:24-255) and 4-FSK RRC modulator (m17_modulate.cpp:22-92); the product's host
generator (m17_txgen.cpp: m17gen_*) must produce the same dibits and the same IQ samples.  CPU only."""
import ctypes as C

import numpy as np
import pytest

from tests import oracle


def _lib():
    import m17_sdr_amd as m
    return m.lib()


def _smix(seed, k):
    M = (1 << 64) - 1
    z = (seed + k * 0x9E3779B97F4A7C15) & M
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & M
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & M
    return z ^ (z >> 31)


def channel_delay(ch):
    """start delay of global channel ch, as the generators' schedule draws it (a property of the test input, not of
    the transmitter: m17gen_batch / m17gpu_gen_batch document `delay (hash of c) % 1920`)"""
    return _smix(0xD1B54A32D192ED03 ^ ((ch * 0x9E3779B97F4A7C15) & ((1 << 64) - 1)), 1) % 1920


def test_lsf_builder_and_frame_builders_match_the_oracle_transmitter():
    lib = _lib()
    rng = np.random.default_rng(21)
    for trial in range(40):
        dst = int(rng.integers(0, 1 << 48)) if trial else 0xFFFFFFFFFFFF
        src = int(rng.integers(0, 1 << 48))
        tw = int(rng.integers(0, 1 << 16))
        meta = rng.integers(0, 256, 14).astype(np.uint8)
        lsf = np.zeros(30, np.uint8)
        assert lib.m17gen_build_lsf(C.c_uint64(dst), C.c_uint64(src), tw, oracle.vp(meta), oracle.vp(lsf)) == 30
        want = oracle.tx_build_lsf(dst, src, tw, meta)
        np.testing.assert_array_equal(lsf, want)
        assert oracle.L().m17o_crc(oracle.vp(want), 30) == 0                      # CRC over the whole LSF is zero
        d = np.zeros(192, np.uint8)
        assert lib.m17gen_lsf_frame_dibits(oracle.vp(lsf), oracle.vp(d)) == 192
        np.testing.assert_array_equal(d, oracle.tx_lsf_frame(want, 0))
        for lich in range(6):
            fn = int(rng.integers(0, 65536))
            pld = rng.integers(0, 256, 16).astype(np.uint8)
            assert lib.m17gen_stream_frame_dibits(oracle.vp(lsf), lich, fn, oracle.vp(pld), oracle.vp(d)) == 192
            np.testing.assert_array_equal(d, oracle.tx_stream_frame(want, lich, fn, pld))
        for length in (0, 1, 7, 24, 25):
            pay = rng.integers(0, 256, 25).astype(np.uint8)
            eof, nf = int(rng.integers(0, 2)), int(rng.integers(0, 32))
            assert lib.m17gen_packet_frame_dibits(oracle.vp(pay), length, eof, nf, oracle.vp(d)) == 192
            np.testing.assert_array_equal(d, oracle.tx_packet_frame(pay[:length], length, eof, nf, 0))
    # more than 25 payload bytes: refused on both sides (m17_tx_routines.cpp:205,221: returns 0)
    assert oracle.tx_packet_frame(np.zeros(26, np.uint8), 26, 0, 0) is None
    assert lib.m17gen_packet_frame_dibits(oracle.vp(np.zeros(32, np.uint8)), 26, 0, 0, oracle.vp(np.zeros(192, np.uint8))) < 0


def test_frames_carry_their_sync_words_and_fixed_patterns():
    """pack_16_to_2 of the sync words (m17_tx_routines.cpp:6-9), the preamble +3 -3 ... and the EOT pattern"""
    lsf = oracle.tx_build_lsf(0xFFFFFFFFFFFF, 0x00102C8DA29F, 5)
    def word(d):
        return sum(int(x) << (14 - 2 * i) for i, x in enumerate(d[:8]))
    assert word(oracle.tx_lsf_frame(lsf)) == 0x55F7
    assert word(oracle.tx_stream_frame(lsf, 0, 0, np.zeros(16, np.uint8))) == 0xFF5D
    assert word(oracle.tx_packet_frame(np.zeros(25, np.uint8), 25, 1, 25)) == 0x75FF
    np.testing.assert_array_equal(oracle.tx_preamble(), np.tile(np.array([1, 3], np.uint8), 96))
    np.testing.assert_array_equal(oracle.tx_eot(), np.tile(np.array([1, 1, 1, 1, 1, 1, 3, 1], np.uint8), 24))


def test_reference_buffer_overrun_in_link_setup_and_packet_frames():
    """`uint8_t tx_bit[2][388]` / `txb[2][388]` (m17_tx_routines.cpp:93,203) are too small for the 488 / 424 coded bits
    of a link-setup / packet frame: the puncturer reads the tail of its input out of the row it is writing.  The oracle
    restates that behind reference_quirks = 1; the generators (and reference_quirks = 0) build the frame as specified.
    What the overrun does: the frame keeps its sync word, differs in a few dozen payload dibits, and no longer decodes
    to the LSF that was sent -- which the reference's own receiver hides behind decode_link_frame's CRC gate."""
    rng = np.random.default_rng(5)
    lvl = np.array([1.0, 3.0, -1.0, -3.0], np.float32)
    ndiff_lsf, ndiff_pkt = [], []
    for trial in range(12):
        lsf = oracle.tx_build_lsf(0xFFFFFFFFFFFF, int(rng.integers(1, 1 << 40)), 5, rng.integers(0, 256, 14).astype(np.uint8))
        good, quirk = oracle.tx_lsf_frame(lsf, 0), oracle.tx_lsf_frame(lsf, 1)
        np.testing.assert_array_equal(good[:8], quirk[:8])
        ndiff_lsf.append(int((good != quirk).sum()))
        for frame, intact in ((good, True), (quirk, False)):
            ch = oracle.Channels(1)
            r = np.zeros(1, oracle.REC_DTYPE)
            oracle.L().m17o_rx_parse(oracle.vp(ch.buf[0]), oracle.vp(lvl[frame & 3] * np.float32(0.3)), 1, oracle.vp(r))
            assert (bytes(r[0]["data"][:30]) == bytes(lsf)) == intact
        pay = rng.integers(0, 256, 25).astype(np.uint8)
        g, q = oracle.tx_packet_frame(pay, 25, 0, trial), oracle.tx_packet_frame(pay, 25, 0, trial, 1)
        np.testing.assert_array_equal(g[:8], q[:8])
        ndiff_pkt.append(int((g != q).sum()))
    assert min(ndiff_lsf) > 0 and max(ndiff_lsf) < 60, ndiff_lsf        # the last 100 of 488 coded bits are wrong: ~half flip
    assert min(ndiff_pkt) > 0 and max(ndiff_pkt) < 30, ndiff_pkt        # the last 32 of 420
    # the stream frame's txb[0] is overrun as well (96 + 296 = 392 > 388) without consequence: nothing to restate
    # (oracle/m17_oracle.c: m17o_stream_frame_dibits)


def test_host_modulator_matches_the_oracle_modulator_sample_for_sample():
    lib = _lib()
    rng = np.random.default_rng(9)
    dib = rng.integers(0, 4, 700).astype(np.uint8)
    dib[100:140] = 255                                                 # a stretch of unmodulated carrier
    iq = np.zeros((7000, 2), np.int16)
    assert lib.m17gen_modulate(oracle.vp(dib), 700, oracle.vp(iq), 1) == 7000
    mod = oracle.Modulator()
    want, sums, ph = mod.modulate(dib, stages=True)
    np.testing.assert_array_equal(iq, want)
    # a second call continues the filter history and the phase
    dib2 = rng.integers(0, 4, 50).astype(np.uint8)
    iq2 = np.zeros((500, 2), np.int16)
    lib.m17gen_modulate(oracle.vp(dib2), 50, oracle.vp(iq2), 0)
    np.testing.assert_array_equal(iq2, mod.modulate(dib2))
    # the modulator is constant-envelope at 0x3FFF and its phase stays within one turn per symbol boundary
    mag = np.hypot(want[:, 0].astype(np.float64), want[:, 1].astype(np.float64))
    assert mag.min() > 16380.5 and mag.max() < 16383.5
    assert np.abs(ph[9::10]).max() < 2 * np.pi + 4.0


@pytest.mark.parametrize("nsf,nblk", [(7, 30), (40, 12)])
def test_host_generator_channel_equals_the_oracle_transmission(nsf, nblk):
    """m17gen_batch (noiseless): every IQ sample of every channel against the oracle transmitter run over the same
    schedule -- start delay, carrier, two preambles, link setup, stream frames, end of transmission, repeating."""
    import m17_sdr_amd as m
    Cn, first = 6, 11
    sig = m.generate_batch(Cn, nblk, n_stream_frames=nsf, first_channel=first, nthreads=2)
    for c in range(Cn):
        delay = channel_delay(first + c)
        sched = oracle.tx_stream_schedule(sig["lsf"][c], sig["payload"][c], nsf, nblk + 1)
        iq = oracle.Modulator().modulate(sched.reshape(-1))
        want = np.empty((nblk * 1920, 2), np.int16)
        want[:delay] = (0x3FFF, 0)
        want[delay:] = iq[:nblk * 1920 - delay]
        np.testing.assert_array_equal(sig["iq"][c].reshape(-1, 2), want)
        # what the generator reports as sent is what the schedule holds
        period = 5 + nsf
        started = sum(1 for g in range(nblk + 1) if 4 <= g % period < 4 + nsf and delay + g * 1920 < nblk * 1920)
        assert sig["nframes"][c] == started


@pytest.mark.parametrize("ebn0,cutoff", [(10.0, 0.0), (8.0, 6250.0)])
def test_host_generator_noise_is_the_specified_awgn_on_top_of_the_oracle_signal(ebn0, cutoff):
    """AWGN is the generators' addition (the reference has none): complex noise of variance
    N0 = Es / (2 Eb/N0) per 48 kHz sample, optionally band-limited by a unity-DC-gain 63-tap low-pass, added to the
    signal the oracle transmitter makes, rounded and clamped to int16."""
    import m17_sdr_amd as m
    Cn, nblk, nsf, first = 12, 14, 9, 2
    sig = m.generate_batch(Cn, nblk, n_stream_frames=nsf, ebn0_db=ebn0, noise_cutoff_hz=cutoff, first_channel=first, nthreads=2)
    sigma = np.sqrt(16383.0 ** 2 * 10 / (2.0 * 2.0 * 10 ** (ebn0 / 10.0)))
    noise = []
    for c in range(Cn):
        delay = channel_delay(first + c)
        iq = oracle.Modulator().modulate(oracle.tx_stream_schedule(sig["lsf"][c], sig["payload"][c], nsf, nblk + 1).reshape(-1))
        clean = np.empty((nblk * 1920, 2), np.float64)
        clean[:delay] = (0x3FFF, 0)
        clean[delay:] = iq[:nblk * 1920 - delay]
        noise.append(sig["iq"][c].reshape(-1, 2).astype(np.float64) - clean)
    noise = np.concatenate(noise)
    assert abs(noise.mean()) < 0.03 * sigma
    frac = noise.var() / sigma ** 2
    if cutoff == 0.0:
        assert abs(frac - 1.0) < 0.04, frac
    else:
        assert 0.8 * (2 * cutoff / 48000.0) < frac < 1.1 * (2 * cutoff / 48000.0), frac
