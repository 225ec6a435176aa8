#!/usr/bin/env python3
"""Regenerates the fixtures in tests/golden/.

SOURCE OF THE VECTORS: the repo's own CPU oracle (oracle/m17_oracle.c) and signal
source -- NOT the reference binary, which cannot be built in this image without
writing stand-ins for absent headers/libraries ("Oracle").  The
fixtures therefore pin the oracle and the HIP path against silent drift (a
regression net), while the link to the reference is carried by the known-answer
values of SURVEY.md 8(c) that tests/test_oracle_kats.py checks.

    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

import m17_sdr_amd as m          # noqa: E402
from tests import oracle         # noqa: E402


def main():
    L = oracle.L()
    mf = np.ctypeslib.as_array(L.m17o_tab_mf(), (40, 31)).copy()
    md = np.ctypeslib.as_array(L.m17o_tab_md(), (40, 31)).copy()
    genc = np.ctypeslib.as_array(L.m17o_tab_golay_enc(), (4096,)).copy()
    gerr = np.ctypeslib.as_array(L.m17o_tab_golay_err(), (4096,)).copy()
    np.savez_compressed(os.path.join(HERE, "tables.npz"), mf=mf, md=md, golay_enc=genc, golay_err=gerr)

    cases = {}
    for name, kw in {
        "noiseless_stream": dict(seed=0x4D313700, nblk=14, n_stream_frames=8, delay=0, ebn0_db=200.0),
        "awgn12_delay777": dict(seed=0x4D313777, nblk=14, n_stream_frames=8, delay=777, ebn0_db=12.0),
        "packet_burst": dict(seed=0x4D313799, nblk=12, n_stream_frames=0, delay=123, ebn0_db=200.0, packet_mode=1),
    }.items():
        iq, lsf, pl, n = m.generate_channel(**kw)
        ch = oracle.Channels(1)
        ref = ch.rx_blocks(iq[None].copy(), mode=1)
        k = int(ref["counts"][0])
        cases[name + "_iq"] = iq
        cases[name + "_lsf"] = lsf
        cases[name + "_recs"] = ref["recs"][0, :k].copy().view(np.uint8).reshape(k, 64)
        cases[name + "_nsyms"] = ref["nsyms"][0].copy()
        cases[name + "_syms"] = ref["syms"][0, :int(ref["nsyms"][0].sum())].copy()
    np.savez_compressed(os.path.join(HERE, "rx_cases.npz"), **cases)
    print("wrote", os.listdir(HERE))


if __name__ == "__main__":
    main()
