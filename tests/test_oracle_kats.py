"""CPU tests (no GPU): the oracle against the known-answer values SURVEY.md 8(c)
recorded from the compiled reference, the M17 constants, the committed golden
fixtures, and the host-side tables / signal source of the product library."""
import ctypes as C
import os

import numpy as np
import pytest

from tests import oracle

HERE = os.path.dirname(os.path.abspath(__file__))


def _crc(b):
    return oracle.L().m17o_crc(bytes(b), len(b))


def test_crc_kats():
    # ""->0xFFFF, "A"->0x206E, "123456789"->0x772B, 0..255->0x1C31, 30 zero bytes->0x1B73
    assert _crc(b"") == 0xFFFF
    assert _crc(b"A") == 0x206E
    assert _crc(b"123456789") == 0x772B
    assert _crc(bytes(range(256))) == 0x1C31
    assert _crc(bytes(30)) == 0x1B73


def test_golay_kats():
    L = oracle.L()
    enc = {0xABC: 0xABC23C, 0x001: 0x0018EB, 0x800: 0x800C75, 0xFFF: 0xFFFFFF}
    for d, w in enc.items():
        assert L.m17o_golay_encode(d) == w
    od = C.c_uint16()
    cw = L.m17o_golay_encode(0xABC)
    assert (L.m17o_golay_decode(cw ^ 0x111000, C.byref(od)), od.value) == (3, 0xABC)
    assert (L.m17o_golay_decode(cw ^ 0x111100, C.byref(od)), od.value) == (4, 0x329)
    assert (L.m17o_golay_decode(cw ^ 0x000F00, C.byref(od)), od.value) == (4, 0x2AC)
    et = np.ctypeslib.as_array(L.m17o_tab_golay_err(), (4096,))
    assert list(np.bincount(et >> 12)) == [1, 24, 276, 2024, 1771]
    assert et[1] == 0x1000 and et[0xFFF] == 0x4880
    h = 1469598103934665603
    for v in et:
        h = ((h ^ int(v)) * 1099511628211) & 0xFFFFFFFFFFFFFFFF
    assert h == 0x3A4F6E4090E92D7F          # FNV-1a-64 over the reference's g_errtab
    # every correctable pattern (weight <= 3) decodes to the sent data
    rng = np.random.default_rng(1)
    for _ in range(300):
        d = int(rng.integers(0, 4096))
        e = 0
        for b in rng.choice(24, int(rng.integers(0, 4)), replace=False):
            e |= 1 << int(b)
        n = L.m17o_golay_decode(L.m17o_golay_encode(d) ^ e, C.byref(od))
        assert od.value == d and n == bin(e).count("1")


def test_callsign_interleaver_randomiser_prbs_kats():
    L = oracle.L()
    assert L.m17o_encode_call(b"G4GUO/P  ") == 0x00102C8DA29F
    assert L.m17o_encode_call(b"AB1CD    ") == 0x0000009FDD51
    buf = C.create_string_buffer(12)
    L.m17o_decode_call(C.c_uint64(0x00102C8DA29F), buf)
    assert buf.value == b"G4GUO/P  "
    assert [((i * 45) + (92 * i * i)) % 368 for i in range(10)] == [0, 137, 90, 227, 180, 317, 270, 39, 360, 129]
    src = np.arange(368, dtype=np.float32)
    dst = np.zeros(368, np.float32)
    L.m17o_de_interleave(oracle.vp(src), oracle.vp(dst), 368)
    assert sorted(dst.tolist()) == list(range(368)) and dst[137] == 1
    der = np.ctypeslib.as_array(L.m17o_tab_derand(), (368,))
    assert "".join(map(str, der[:16])) == "1101011010110101"
    p = (C.c_uint8 * 16)()
    L.m17o_prbs9(p, 16)
    assert "".join(map(str, p)) == "0000100011000010"


def test_rrc_tap_kats_and_polyphase_tables():
    L = oracle.L()
    f = (C.c_float * 1240)()
    L.m17o_build_rrc_filter(f, C.c_float(0.5), 1240, 80)
    got = np.array([f[0], f[619], f[620], f[1239]], np.float32)
    want = np.array([-4.53409848e-05, 0.0635405034, 0.0635166764, -5.35539584e-05], np.float32)
    np.testing.assert_array_equal(got, want)
    mf = np.ctypeslib.as_array(L.m17o_tab_mf(), (40, 31))
    md = np.ctypeslib.as_array(L.m17o_tab_md(), (40, 31))
    # every polyphase branch is normalised to unit DC gain, the derivative branch is not
    assert np.allclose(mf.sum(axis=1), 1.0, atol=2e-6)
    g = np.load(os.path.join(HERE, "golden", "tables.npz"))
    np.testing.assert_array_equal(mf.view(np.uint32), g["mf"].view(np.uint32))
    np.testing.assert_array_equal(md.view(np.uint32), g["md"].view(np.uint32))


def test_viterbi_branch_table_and_round_trip():
    L = oracle.L()
    # conv round trip of SURVEY 8c: 18 bytes -> 296 coded -> P2 -> 272 -> depuncture -> Viterbi
    libc = C.CDLL(None)
    libc.srand(7)
    data = bytes(libc.rand() & 0xFF for _ in range(18))
    coded = (C.c_uint8 * 400)()
    punct = (C.c_uint8 * 400)()
    n = L.m17o_conv_encode_8(data, coded, 18)
    assert n == 296
    npn = L.m17o_punc(2, coded, punct, n)
    assert npn == 272
    soft = np.array([1.0 if punct[i] else -1.0 for i in range(npn)], np.float32)
    dep = np.zeros(296, np.float32)
    assert L.m17o_de_punc(2, oracle.vp(soft), oracle.vp(dep), 296) == 296
    bits = oracle.viterbi(dep)
    assert len(bits) == 148 and bits[0] == 0
    out = (C.c_uint8 * 18)()
    L.m17o_pack_1_to_8(oracle.vp(np.ascontiguousarray(bits[1:145])), out, 144)
    assert bytes(out) == data
    # tie-break: all-zero metrics keep choosing the odd predecessor -> all-ones path except out[0]
    z = oracle.viterbi(np.zeros(40, np.float32))
    assert z[0] == 0


def test_sync_templates_match_m17_sync_words():
    L = oracle.L()
    words = {1: 0x55F7, 2: 0xFF5D, 3: 0x75FF, 4: 0xDF55}
    sym = {0: 1.0, 1: 3.0, 2: -1.0, 3: -3.0}
    for t, w in words.items():
        v = np.array([sym[(w >> (14 - 2 * i)) & 3] for i in range(8)], np.float32) * 0.1
        ty, vo, var = C.c_uint8(), C.c_uint8(), C.c_float()
        L.m17o_sync_check(oracle.vp(v), C.byref(ty), C.byref(vo), C.byref(var))
        assert (ty.value, vo.value) == (t, 0) and var.value == 0.0


def test_loopback_behaviour_of_survey_probe():
    """the reference locks 3 blocks after start, parses the LSF and all 40
    stream frames byte-exactly and starts delivering at FN 5 (LICH complete)."""
    import m17_sdr_amd as m
    iq, lsf, pl, n = m.generate_channel(0x4D313700, 50, n_stream_frames=40)
    ref = oracle.Channels(1).rx_blocks(iq[None].copy(), mode=1)
    recs = ref["recs"][0, :ref["counts"][0]]
    assert recs[0]["flags"] == m.F_AOS and recs[0]["block"] == 3 and recs[0]["type"] == 1
    assert recs[1]["type"] == 1 and bytes(recs[1]["data"][:30]) == bytes(lsf)
    stream = [r for r in recs[2:42] if r["type"] == 2]
    assert len(stream) == 40
    for k, r in enumerate(stream):
        assert r["fn"] == k and bytes(r["data"][8:24]) == bytes(pl[k])
        assert bool(r["flags"] & m.F_DELIVERED) == (k >= 5)
    assert recs[42]["flags"] & m.F_EOT


def test_golden_rx_cases():
    g = np.load(os.path.join(HERE, "golden", "rx_cases.npz"))
    for name in ("noiseless_stream", "awgn12_delay777", "packet_burst"):
        iq = g[name + "_iq"]
        ref = oracle.Channels(1).rx_blocks(iq[None].copy(), mode=1)
        k = int(ref["counts"][0])
        assert ref["recs"][0, :k].tobytes() == g[name + "_recs"].tobytes(), name
        np.testing.assert_array_equal(ref["nsyms"][0], g[name + "_nsyms"])
        ns = int(ref["nsyms"][0].sum())
        np.testing.assert_array_equal(ref["syms"][0, :ns].view(np.uint32), g[name + "_syms"].view(np.uint32))


def test_product_host_tables_equal_oracle_tables():
    import m17_sdr_amd as m
    lib = m.lib()
    mf = np.zeros((40, 31), np.float32)
    md = np.zeros((40, 31), np.float32)
    lib.m17gpu_get_taps(oracle.vp(mf), oracle.vp(md))
    genc = np.zeros(4096, np.uint16)
    gerr = np.zeros(4096, np.uint16)
    lib.m17gpu_get_golay_tables(oracle.vp(genc), oracle.vp(gerr))
    g = np.load(os.path.join(HERE, "golden", "tables.npz"))
    np.testing.assert_array_equal(mf.view(np.uint32), g["mf"].view(np.uint32))
    np.testing.assert_array_equal(md.view(np.uint32), g["md"].view(np.uint32))
    np.testing.assert_array_equal(genc, g["golay_enc"])
    np.testing.assert_array_equal(gerr, g["golay_err"])


def test_signal_source_frames_round_trip_through_oracle_decoder():
    """encode -> (hard symbols) -> decode for LSF, stream and packet frames, incl. erasures."""
    import m17_sdr_amd as m
    lib = m.lib()
    rng = np.random.default_rng(3)
    lsf = np.zeros(30, np.uint8)
    meta = np.zeros(14, np.uint8)
    lib.m17gen_build_lsf(0xFFFFFFFFFFFF, lib.m17gen_encode_call(b"N0CALL   "), 5, oracle.vp(meta), oracle.vp(lsf))
    assert _crc(bytes(lsf)) == 0
    lvl = {0: 1.0, 1: 3.0, 2: -1.0, 3: -3.0}
    dib = np.zeros(192, np.uint8)
    for trial in range(20):
        pld = rng.integers(0, 256, 16).astype(np.uint8)
        fn = int(rng.integers(0, 65536))
        lib.m17gen_stream_frame_dibits(oracle.vp(lsf), trial % 6, fn, oracle.vp(pld), oracle.vp(dib))
        s = np.array([lvl[int(d)] for d in dib], np.float32) * 0.02
        s[rng.choice(np.arange(8, 192), 6, replace=False)] = 0.0     # erased symbols
        ch = oracle.Channels(1)
        r = np.zeros(1, oracle.REC_DTYPE)
        oracle.L().m17o_rx_parse(oracle.vp(ch.buf[0]), oracle.vp(s), 2, oracle.vp(r))
        assert r[0]["fn"] == fn and bytes(r[0]["data"][8:24]) == bytes(pld)
        assert bytes(r[0]["data"][:5]) == bytes(lsf[(trial % 6) * 5:(trial % 6) * 5 + 5])
    lib.m17gen_lsf_frame_dibits(oracle.vp(lsf), oracle.vp(dib))
    s = np.array([lvl[int(d)] for d in dib], np.float32) * 0.3
    ch = oracle.Channels(1)
    r = np.zeros(1, oracle.REC_DTYPE)
    oracle.L().m17o_rx_parse(oracle.vp(ch.buf[0]), oracle.vp(s), 1, oracle.vp(r))
    assert bytes(r[0]["data"][:30]) == bytes(lsf)
    pk = rng.integers(0, 256, 25).astype(np.uint8)
    lib.m17gen_packet_frame_dibits(oracle.vp(pk), 25, 0, 3, oracle.vp(dib))
    s = np.array([lvl[int(d)] for d in dib], np.float32)
    oracle.L().m17o_rx_parse(oracle.vp(ch.buf[0]), oracle.vp(s), 3, oracle.vp(r))
    assert bytes(r[0]["data"][:25]) == bytes(pk) and r[0]["fn"] == 3


def test_empty_and_degenerate_inputs():
    # all-zero symbols: sum of |sync| is 0 -> cor = inf -> NaN soft bits; the decoder must not crash
    ch = oracle.Channels(1)
    r = np.zeros(1, oracle.REC_DTYPE)
    oracle.L().m17o_rx_parse(oracle.vp(ch.buf[0]), oracle.vp(np.zeros(192, np.float32)), 2, oracle.vp(r))
    # a (0,0) IQ sample poisons the block's DC estimate with NaN but nothing else breaks
    iq = np.full((1, 2, 1920, 2), 1000, np.int16)
    iq[0, 0, 100] = 0
    ref = oracle.Channels(1).rx_blocks(iq, mode=1)
    assert 0 < ref["nsyms"][0, 1] <= 193


def test_pluto_decimator_oracle_properties():
    """radio.cpp:18-51,157-177 restated: tap design, DC gain, streaming (chunked == whole)."""
    L = oracle.L()
    c = np.zeros(31, np.int16)
    L.m17o_pluto_build_dec_filter(oracle.vp(c))
    assert np.array_equal(c, c[::-1]) and c[15] == c.max() and c[7] == 0 and c[23] == 0   # sinc zeros at +-8
    assert abs(int(c.sum()) - round(0.9 * 32767)) <= 16                                     # truncated Q15, DC gain 0.9
    rng = np.random.default_rng(4)
    x = rng.integers(-32768, 32768, (1920 * 8 * 2, 2)).astype(np.int16)
    whole = np.zeros((len(x) // 8, 2), np.int16)
    h = np.zeros((31, 2), np.int16)
    L.m17o_pluto_decimate(oracle.vp(h), oracle.vp(x), len(x), oracle.vp(whole))
    # the reference consumes 1920 wide-band samples per inner step (radio.cpp:164-172)
    h2 = np.zeros((31, 2), np.int16)
    parts = []
    for k in range(0, len(x), 1920):
        o = np.zeros((240, 2), np.int16)
        L.m17o_pluto_decimate(oracle.vp(h2), oracle.vp(np.ascontiguousarray(x[k:k + 1920])), 1920, oracle.vp(o))
        parts.append(o)
    np.testing.assert_array_equal(np.concatenate(parts), whole)
    # direct evaluation of sub_filter on one output
    i = 777
    win = x[8 * i - 31: 8 * i].astype(np.int64)
    want = (win * c[:, None].astype(np.int64)).sum(axis=0) >> 15
    np.testing.assert_array_equal(whole[i], want.astype(np.int16))
    # constant input settles at gain sum(c)/32768
    k = np.full((4000, 2), 10000, np.int16)
    o = np.zeros((500, 2), np.int16)
    L.m17o_pluto_decimate(oracle.vp(np.zeros((31, 2), np.int16)), oracle.vp(k), 4000, oracle.vp(o))
    assert o[-1, 0] == (10000 * int(c.sum())) >> 15
