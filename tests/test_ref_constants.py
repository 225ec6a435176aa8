"""The literal constants of the path, pinned to the reference's SOURCE TEXT.

tests/golden/ref_constants.json holds the tables regex-extracted from /root/reference/m17gismo by
tests/golden/extract_ref_constants.py (data only, file:line recorded per entry).  Both the CPU
oracle and the product library must be built from exactly these values -- this narrows the
common-mode risk of an oracle and a product restated by the same hand (it is not a substitute for
reference-run vectors, which cannot be produced under the build rules: DESIGN.md section 2)."""
import ctypes as C
import json
import os

import numpy as np

from tests import oracle

HERE = os.path.dirname(os.path.abspath(__file__))
REF = json.load(open(os.path.join(HERE, "golden", "ref_constants.json")))


def _product(name, dtype, count):
    import m17_sdr_amd as m
    buf = np.zeros(count, dtype)
    n = m.lib().m17gpu_get_constant(name.encode(), buf.ctypes.data_as(C.c_void_p), buf.nbytes)
    assert n == buf.nbytes, (name, n)
    return buf


def _oracle(name, dtype, count):
    lib = oracle.L()
    lib.m17o_get_constant.argtypes = [C.c_char_p, C.c_void_p, C.c_int]
    buf = np.zeros(count, dtype)
    n = lib.m17o_get_constant(name.encode(), buf.ctypes.data_as(C.c_void_p), buf.nbytes)
    assert n == buf.nbytes, (name, n)
    return buf


def test_sync_templates():
    want = np.array(REF["sframe"]["value"], np.float32)
    np.testing.assert_array_equal(_oracle("sframe", np.float32, 48).reshape(6, 8), want)
    np.testing.assert_array_equal(_product("sframe", np.float32, 48).reshape(6, 8), want)
    # the transmit sync words are the same templates: dibit 01 -> +3, 11 -> -3 (rows 1..4)
    for k, word in enumerate(REF["sync_words"]["value"]):
        dibits = [(word >> (14 - 2 * i)) & 3 for i in range(8)]
        assert all(d in (1, 3) for d in dibits)
        np.testing.assert_array_equal(want[1 + k], [1.0 if d == 1 else -1.0 for d in dibits])
    np.testing.assert_array_equal(_product("sync_words", np.uint16, 4), REF["sync_words"]["value"])


def test_viterbi_butterfly_rows():
    want = np.array(REF["butterfly"]["value"], np.uint8)
    assert want.shape == (16, 5)
    np.testing.assert_array_equal(_oracle("butterfly", np.uint8, 80).reshape(16, 5), want)
    np.testing.assert_array_equal(_product("butterfly", np.uint8, 80).reshape(16, 5), want)
    assert REF["butterfly_macro"]["strict_greater"] and REF["butterfly_macro"]["else_takes_y"]
    # the rows follow from the code's generator taps (m17_conv.cpp:24-29), as the commented generator says
    t0, t1 = REF["conv_taps"]["value"]
    def out(sr, taps):
        return sum((sr >> t) & 1 for t in taps) & 1
    for v in range(16):
        for p, col in ((0, 2), (1, 4)):
            sr = (v << 1) | p
            assert want[v][col] == (out(sr, t0) << 1) | out(sr, t1)
            assert want[v][col - 1] == (2 * v + p) % 16


def test_randomiser_golay_puncture_crc():
    np.testing.assert_array_equal(_oracle("derand_bytes", np.uint8, 46), REF["derand_bytes"]["value"])
    bits = np.unpackbits(np.array(REF["derand_bytes"]["value"], np.uint8))        # MSB first (m17_correlate.cpp:35-42)
    np.testing.assert_array_equal(_product("derand_bits", np.uint8, 368), bits)
    np.testing.assert_array_equal(np.ctypeslib.as_array(oracle.L().m17o_tab_derand(), (368,)), bits)
    np.testing.assert_array_equal(_oracle("golay_rows", np.uint16, 12), REF["golay_rows"]["value"])
    np.testing.assert_array_equal(_product("golay_rows", np.uint16, 12), REF["golay_rows"]["value"])
    for name in ("punc1", "punc2", "punc3"):
        want = np.array(REF[name]["value"], np.uint8)
        np.testing.assert_array_equal(_oracle(name, np.uint8, len(want)), want)
        np.testing.assert_array_equal(_product(name, np.uint8, len(want)), want)
    assert int(_oracle("crc_poly", np.uint16, 1)[0]) == REF["crc_poly"]["value"]
    assert int(_product("crc_poly", np.uint16, 1)[0]) == REF["crc_poly"]["value"]
    # whole CRC table from the polynomial, MSB first (m17_crc.cpp:8-22)
    poly = REF["crc_poly"]["value"]
    tab = []
    for i in range(256):
        x = i << 8
        for _ in range(8):
            x = ((x << 1) ^ poly) & 0xFFFF if x & 0x8000 else (x << 1) & 0xFFFF
        tab.append(x)
    np.testing.assert_array_equal(np.ctypeslib.as_array(oracle.L().m17o_tab_crc(), (256,)), tab)


def test_modulator_literals_and_geometry():
    lut = np.array(REF["tx_lut"]["value_double"], np.float64).astype(np.float32)
    np.testing.assert_array_equal(_product("tx_lut", np.float32, 4).view(np.uint32), lut.view(np.uint32))
    assert REF["tx_phase_type"]["m_acc_is_float"] and REF["tx_phase_type"]["amplitude"] == 0x3FFF
    import m17_sdr_amd as m
    assert REF["defines"]["value"] == {"N_SAMPLES": m.BLOCK_SAMPLES, "SRATE": 48000, "FRAME_SYM_LENGTH": m.FRAME_SYMS}
    assert REF["sync_filter_geometry"]["value"] == {"NF": 40, "FN": 31}
    a, b = REF["interleave_poly"]["value"]
    perm = [(a * i + b * i * i) % 368 for i in range(368)]
    assert sorted(perm) == list(range(368))
    # the product's fused gather tables are built on this permutation: spot-check through the oracle's
    soft = np.arange(368, dtype=np.float32)
    outv = np.zeros(368, np.float32)
    oracle.L().m17o_de_interleave(oracle.vp(soft), oracle.vp(outv), 368)
    np.testing.assert_array_equal(outv[perm], soft)


def test_streaming_arithmetic_literals():
    """The literals and control constants of the streaming DSP stages -- int16 scale, demapper offset and reference, the
    discriminator's factor and /5 pick, the timing loop's thresholds and initial state, the framer's vote / variance gates,
    N_FERROR, the frame clock after a sync, the Viterbi start metric and traceback mask, the Golay error table's fill -- as
    tests/golden/extract_ref_constants.py found them in the reference's source text, each inside the statement that uses
    it (file:line recorded per entry).  Oracle and product each hold them under ONE name per literal, used by their code
    and returned by their getters: all three must agree, so a mistyped threshold on either side cannot hide behind the
    other.  (It pins constants, not arithmetic: parity of the streaming stages stays "unpinned")"""
    ref = REF["rx_literals"]
    want = np.array(ref["value"], np.float64)
    assert len(ref["order"]) == len(want) == 24 and all(k in ref["source"] for k in ref["order"])
    np.testing.assert_array_equal(_oracle("rx_literals", np.float64, 24), want)
    np.testing.assert_array_equal(_product("rx_literals", np.float64, 24), want)
    named = dict(zip(ref["order"], want))
    # the values the verdicts of rounds 1-5 read out of the reference by eye
    assert named["s16_scale"] == 0.00003 and named["demap_offset"] == 0.6666 and named["thresh_unlocked"] == 10 and \
        named["thresh_locked"] == 80 and named["index_init"] == 10 and named["n_ferror"] == 5 and named["golay_fill_end"] == 0xFFF
    # the oracle's freshly reset channel starts from the extracted initial state
    ch = oracle.Channels(1)
    assert (int(ch.field("m_clk")[0]), int(ch.field("m_thr")[0]), int(ch.field("m_index")[0])) == \
        (int(named["clk_init"]), int(named["thr_init"]), int(named["index_init"]))
