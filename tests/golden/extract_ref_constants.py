#!/usr/bin/env python3
"""Build-container script: regex-extract the literal constant tables of the reference's receive /
transmit path from its SOURCE TEXT under /root/reference/m17gismo into
tests/golden/ref_constants.json (data only: numbers, with the file:line each came from).

The reference cannot be compiled under this project's rules (m17defines.h needs codec2.h, absent),
so this is the one reference-derived pin of the constants that the oracle (oracle/m17_oracle.c) and
the product tables (m17_sdr_amd/csrc) are otherwise only restated from: tests/test_ref_constants.py
asserts that both equal these values.  Nothing here is executed from the reference; its files are
read as text.  Re-run only in a container that has /root/reference."""
import json
import math
import os
import re
import sys

REF = "/root/reference/m17gismo"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "ref_constants.json")


def text(name):
    with open(os.path.join(REF, name), "r", errors="replace") as f:
        return f.read()


def strip_comments(s):
    s = re.sub(r"/\*.*?\*/", " ", s, flags=re.S)
    return re.sub(r"//[^\n]*", " ", s)


def line_of(src, pos):
    return src.count("\n", 0, pos) + 1


def ints(body):
    return [int(t, 0) for t in re.findall(r"-?(?:0[xX][0-9a-fA-F]+|\d+)", body)]


def find_array(src, pattern):
    """pattern matches up to the opening '{' of an initialiser; returns (body without comments, first line, last line)."""
    m = re.search(pattern, src)
    if not m:
        raise SystemExit(f"pattern not found: {pattern}")
    depth, i = 1, m.end()
    while depth:
        c = src[i]
        depth += (c == "{") - (c == "}")
        i += 1
    return strip_comments(src[m.end():i - 1]), line_of(src, m.start()), line_of(src, i)


def main():
    if not os.path.isdir(REF):
        raise SystemExit(f"{REF} not present: this script only runs in the build container")
    out = {"_note": "extracted from the reference's source text by tests/golden/extract_ref_constants.py; data only"}

    src = text("m17_rx_frame.cpp")
    body, a, b = find_array(src, r"sframe\s*\[\s*6\s*\]\s*\[\s*8\s*\]\s*=\s*\{")
    rows = [ints(r) for r in re.findall(r"\{([^{}]*)\}", body)]
    assert len(rows) == 6 and all(len(r) == 8 for r in rows), rows
    out["sframe"] = {"source": f"m17_rx_frame.cpp:{a}-{b}", "value": rows}

    src = text("m17_conv.cpp")
    bf = [(line_of(src, m.start()), [int(x) for x in m.groups()])
          for m in re.finditer(r"^[ \t]*BF\(\s*(\d+)\s*,\s*(\d+)\s*,\s*(\d+)\s*,\s*(\d+)\s*,\s*(\d+)\s*\)\s*$", src, flags=re.M)]
    assert len(bf) == 16, len(bf)
    out["butterfly"] = {"source": f"m17_conv.cpp:{bf[0][0]}-{bf[-1][0]}", "columns": "v,w,x,y,z of BF(v,w,x,y,z)",
                        "value": [r for _, r in bf]}
    m = re.search(r"#define\s+BF\(v,w,x,y,z\)\s+(.*)", src)
    out["butterfly_macro"] = {"source": f"m17_conv.cpp:{line_of(src, m.start())}",
                              "strict_greater": "if(tempa>tempb)" in m.group(1).replace(" ", ""),
                              "else_takes_y": "else{tm[v]=tempb;m_path[v][m_hp]=y;}" in m.group(1).replace(" ", "")}
    m = re.search(r"clut\[i\]\[0\]\s*=\s*([^;]*);\s*clut\[i\]\[1\]\s*=\s*([^;]*);", src)
    taps = []
    for expr in m.groups():
        taps.append(sorted(int(t) if t else 0 for t in re.findall(r"\(i>>(\d+)\)&1", expr)) + ([0] if re.search(r"\^\(i&1\)", expr) else []))
    out["conv_taps"] = {"source": f"m17_conv.cpp:{line_of(src, m.start())}-{line_of(src, m.end())}",
                        "note": "bit positions of the 5-bit register XORed into output 0 and output 1",
                        "value": [sorted(t) for t in taps]}

    src = text("m17_correlate.cpp")
    body, a, b = find_array(src, r"ctab\s*\[\s*46\s*\]\s*=\s*\{")
    v = ints(body)
    assert len(v) == 46
    out["derand_bytes"] = {"source": f"m17_correlate.cpp:{a}-{b}", "value": v}

    src = text("m17_golay.cpp")
    body, a, b = find_array(src, r"gtab\s*\[\s*12\s*\]\s*=\s*\{")
    v = ints(body)
    assert len(v) == 12
    out["golay_rows"] = {"source": f"m17_golay.cpp:{a}", "value": v}

    src = text("m17_puncture.cpp")
    for name, n in (("P1", 61), ("P2", 12), ("P3", 8)):
        body, a, b = find_array(src, name + r"\s*\[\s*%d\s*\]\s*=\s*\{" % n)
        v = ints(body)
        assert len(v) == n, (name, len(v))
        out["punc" + name[1]] = {"source": f"m17_puncture.cpp:{a}-{b}", "value": v}

    src = text("m17_modulate.cpp")
    body, a, b = find_array(src, r"m_tx_lu\s*\[\s*4\s*\]\s*=\s*\{")
    exprs = [e.strip() for e in body.split(",")]
    assert len(exprs) == 4 and all(re.fullmatch(r"-?M_PI/[0-9.]+", e) for e in exprs), exprs
    vals = [(-1.0 if e.startswith("-") else 1.0) * math.pi / float(e.split("/")[1]) for e in exprs]
    out["tx_lut"] = {"source": f"m17_modulate.cpp:{a}", "expressions": exprs, "value_double": vals}
    m = re.search(r"=\s*cos\(m_acc\)\*0x3FFF;", src)
    m2 = re.search(r"static\s+float\s+m_acc\s*;", src)
    out["tx_phase_type"] = {"source": f"m17_modulate.cpp:{line_of(src, m.start())}",
                            "m_acc_is_float": bool(m2), "amplitude": 0x3FFF}

    src = text("m17_tx_routines.cpp")
    words = {}
    for nm in ("SYNC_LINK_SETUP", "SYNC_STREAM", "SYNC_PACKET", "SYNC_BERT"):
        m = re.search(r"#define\s+%s\s+(0[xX][0-9a-fA-F]+)" % nm, src)
        words[nm] = (int(m.group(1), 16), line_of(src, m.start()))
    out["sync_words"] = {"source": "m17_tx_routines.cpp:%d-%d" % (min(w[1] for w in words.values()), max(w[1] for w in words.values())),
                         "order": list(words), "value": [w[0] for w in words.values()]}

    src = text("m17_crc.cpp")
    m = re.search(r"#define\s+CRC_POLY\s+(0[xX][0-9a-fA-F]+)", src)
    out["crc_poly"] = {"source": f"m17_crc.cpp:{line_of(src, m.start())}", "value": int(m.group(1), 16)}

    src = text("m17_interleave.cpp")
    m = re.search(r"\(\(i\s*\*\s*(\d+)\)\s*\+\s*\(\s*(\d+)\s*\*\s*i\s*\*\s*i\s*\)\)\s*%\s*(\w+)", src) or \
        re.search(r"(\d+)\s*\*\s*i\s*\+\s*(\d+)\s*\*\s*i\s*\*\s*i", src)
    if m:
        out["interleave_poly"] = {"source": f"m17_interleave.cpp:{line_of(src, m.start())}",
                                  "value": [int(m.group(1)), int(m.group(2))]}

    src = text("m17defines.h")
    defs = {}
    for nm in ("N_SAMPLES", "SRATE", "FRAME_SYM_LENGTH"):
        m = re.search(r"#define\s+%s\s+(\d+)" % nm, src)
        if m:
            defs[nm] = int(m.group(1))
    out["defines"] = {"source": "m17defines.h", "value": defs}

    src = text("m17_rx_sync.cpp")
    m_nf = re.search(r"#define\s+NF\s+(\d+)", src)
    m_fn = re.search(r"#define\s+FN\s+(\d+)", src)
    if m_nf and m_fn:
        out["sync_filter_geometry"] = {"source": f"m17_rx_sync.cpp:{line_of(src, m_nf.start())}-{line_of(src, m_fn.start())}",
                                       "value": {"NF": int(m_nf.group(1)), "FN": int(m_fn.group(1))}}

    with open(OUT, "w") as f:
        json.dump(out, f, indent=1)
        f.write("\n")
    print("wrote", OUT, "with", len(out) - 1, "entries")


if __name__ == "__main__":
    sys.exit(main())
