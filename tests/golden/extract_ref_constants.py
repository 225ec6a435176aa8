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

    # ---- the literals and control constants of the STREAMING arithmetic (round 6): each found in place, in the statement that
    # uses it, so that a changed statement shape fails the extraction instead of silently matching something else
    lit, where = {}, {}

    def grab(fname, src, name, pattern, conv=float, group=1, flags=0):
        m = re.search(pattern, src, flags)
        if not m:
            raise SystemExit(f"{fname}: pattern for {name} not found: {pattern}")
        lit[name] = conv(m.group(group))
        where[name] = f"{fname}:{line_of(src, m.start(group))}"
        return m

    def fn_body(src, header_pattern):
        m = re.search(header_pattern, src)
        if not m:
            raise SystemExit(f"function not found: {header_pattern}")
        depth, i = 1, src.index("{", m.end() - 1) + 1
        start = i
        while depth:
            depth += (src[i] == "{") - (src[i] == "}")
            i += 1
        return start, i

    hexint = lambda t: int(t, 0)
    src = text("m17_dsp.cpp")
    grab("m17_dsp.cpp", src, "s16_scale", r"out\[i\]\.re\s*=\s*in\[i\]\.re\s*\*\s*([0-9.eE+-]+)\s*;")
    m = re.search(r"out\[i\]\.im\s*=\s*in\[i\]\.im\s*\*\s*([0-9.eE+-]+)\s*;", src)
    assert m and float(m.group(1)) == lit["s16_scale"], "re and im are scaled alike"
    a0, b0 = fn_body(src, r"void\s+m17_dsp_demap_symbol\s*\(\s*float in\s*,\s*float mag\s*,\s*float \*out\s*\)\s*\{")
    body = strip_comments(src[a0:b0])
    assert re.search(r"m\s*=\s*in\s*\*\s*mag\s*;", body) and re.search(r"out\[0\]\s*=\s*-m\s*;", body), "demap: m = in*mag; out[0] = -m"
    mm = re.search(r"out\[1\]\s*=\s*\(\s*fabs\(m\)\s*-\s*([0-9.]+)\s*\)\s*;", src[a0:b0])
    lit["demap_offset"] = float(mm.group(1)); where["demap_offset"] = f"m17_dsp.cpp:{line_of(src, a0 + mm.start(1))}"
    a0, b0 = fn_body(src, r"void\s+m17_dsp_demap_frame\s*\(\s*float \*in\s*,\s*float \*out\s*\)\s*\{")
    mm = re.search(r"float\s+cor\s*=\s*([0-9.]+)\s*/\s*sum\s*;", src[a0:b0])
    lit["demap_cor_num"] = float(mm.group(1)); where["demap_cor_num"] = f"m17_dsp.cpp:{line_of(src, a0 + mm.start(1))}"
    mm = re.search(r"for\(\s*int i = 0; i < (\d+); i\+\+\)\{\s*sum \+= fabs\(in\[i\]\);", src[a0:b0])
    lit["demap_sync_symbols"] = int(mm.group(1)); where["demap_sync_symbols"] = f"m17_dsp.cpp:{line_of(src, a0 + mm.start(1))}"
    a0, b0 = fn_body(src, r"static\s+int\s+dsp_arctan_disc2\s*\([^)]*\)\s*\{")
    body = src[a0:b0]
    mm = re.search(r"\bc\s*=\s*([0-9.]+)\s*;", body)
    lit["disc_c"] = float(mm.group(1)); where["disc_c"] = f"m17_dsp.cpp:{line_of(src, a0 + mm.start(1))}"
    mm = re.search(r"count\s*=\s*\(count\+1\)\s*%\s*(\d+)\s*;\s*if\(count == 0\)", body)
    lit["disc_decim"] = int(mm.group(1)); where["disc_decim"] = f"m17_dsp.cpp:{line_of(src, a0 + mm.start(1))}"
    assert re.search(r"out\[idx\+\+\]\s*=\s*u\*c\s*;", body) and re.search(r"offset\s*\+=\s*u\*c\s*;", body) and \
        re.search(r"offset\s*=\s*offset/len\s*;", body), "discriminator: pick u*c, sum u*c, offset/len"
    a0, b0 = fn_body(src, r"static\s+void\s+dsp_limit\s*\([^)]*\)\s*\{")
    body = src[a0:b0]
    mm = re.search(r"float\s+g\s*=\s*([0-9.]+)\s*/\s*m\s*;", body)
    lit["limit_num"] = float(mm.group(1)); where["limit_num"] = f"m17_dsp.cpp:{line_of(src, a0 + mm.start(1))}"
    assert re.search(r"float\s+m\s*=\s*sqrt\(in\[i\]\.re\*in\[i\]\.re \+ in\[i\]\.im\*in\[i\]\.im\)\s*;", body), "limiter: m = sqrt(re*re + im*im)"

    src = text("m17_rx_sync.cpp")
    a0, b0 = fn_body(src, r"int\s+m17_rx_sync_samples\s*\([^)]*\)\s*\{")
    body = src[a0:b0]
    mm = re.search(r"if\(m17_rx_lock\(\) == false\)\s*m17_sync_adjust\((\d+), out\);\s*else\s*m17_sync_adjust\((\d+), out\);", body)
    lit["thresh_unlocked"] = int(mm.group(1)); where["thresh_unlocked"] = f"m17_rx_sync.cpp:{line_of(src, a0 + mm.start(1))}"
    lit["thresh_locked"] = int(mm.group(2)); where["thresh_locked"] = f"m17_rx_sync.cpp:{line_of(src, a0 + mm.start(2))}"
    mm = re.search(r"m_clk\s*=\s*\(m_clk\+1\)\s*%\s*(\d+)\s*;", body)
    lit["clk_modulus"] = int(mm.group(1)); where["clk_modulus"] = f"m17_rx_sync.cpp:{line_of(src, a0 + mm.start(1))}"
    a0, b0 = fn_body(src, r"void\s+m17_rx_sync_init\s*\([^)]*\)\s*\{")
    body = src[a0:b0]
    for nm, var in (("clk_init", "m_clk"), ("thr_init", "m_thr"), ("index_init", "m_index")):
        mm = re.search(r"\b%s\s*=\s*(\d+)\s*;" % var, body)
        lit[nm] = int(mm.group(1)); where[nm] = f"m17_rx_sync.cpp:{line_of(src, a0 + mm.start(1))}"

    src = text("m17_rx_frame.cpp")
    for nm, fn in (("unlocked", "m17_unlocked_sync_check"), ("locked", "m17_locked_sync_check")):
        a0, b0 = fn_body(src, r"bool\s+%s\s*\([^)]*\)\s*\{" % fn)
        body = src[a0:b0]
        mm = re.search(r"if\(sync->votes > (\d+) \)\{\s*return false;", body)
        lit["votes_" + nm + "_max"] = int(mm.group(1)); where["votes_" + nm + "_max"] = f"m17_rx_frame.cpp:{line_of(src, a0 + mm.start(1))}"
        mm = re.search(r"if\( sync->variance < ([0-9.]+)\)\{\s*return true;", body)
        lit["var_" + nm] = float(mm.group(1)); where["var_" + nm] = f"m17_rx_frame.cpp:{line_of(src, a0 + mm.start(1))}"
        types = [int(t) for t in re.findall(r"sync->type == (\d+)", body)]
        assert types == [1, 2, 3, 4], (fn, types)
    grab("m17_rx_frame.cpp", src, "n_ferror", r"#define\s+N_FERROR\s+(\d+)", int)
    assert re.search(r"if\(\s*m_frame_errors > N_FERROR\s*\)", src), "m_frame_errors > N_FERROR"
    a0, b0 = fn_body(src, r"void\s+m17_rx_sym\s*\(\s*float sym\s*\)\s*\{")
    mm = re.search(r"copy_sync\(\);\s*m_fclk\s*=\s*(\d+)\s*;", src[a0:b0])
    lit["fclk_after_sync"] = int(mm.group(1)); where["fclk_after_sync"] = f"m17_rx_frame.cpp:{line_of(src, a0 + mm.start(1))}"

    src = text("m17_conv.cpp")
    a0, b0 = fn_body(src, r"int\s+m17_viterbi_decode\s*\([^)]*\)\s*\{")
    body = src[a0:b0]
    mm = re.search(r"\n\s*m_acm\[0\]\s*=\s*([0-9.]+)\s*;", body)
    lit["acm0"] = float(mm.group(1)); where["acm0"] = f"m17_conv.cpp:{line_of(src, a0 + mm.start(1))}"
    mm = re.search(r"out\[i\]\s*=\s*state&(0x[0-9a-fA-F]+)\?1:0;", body)
    lit["traceback_mask"] = int(mm.group(1), 16); where["traceback_mask"] = f"m17_conv.cpp:{line_of(src, a0 + mm.start(1))}"
    assert re.search(r"uint8_t\s+state\s*=\s*0\s*;", body), "traceback starts in state 0"

    src = text("m17_golay.cpp")
    a0, b0 = fn_body(src, r"static\s+void\s+golay_build_error_table\s*\(\s*void\s*\)\s*\{")
    body = src[a0:b0]
    mm = re.search(r"for\(int i = 0 ; i < (0x[0-9a-fA-F]+); i\+\+\)\{\s*g_errtab\[i\]\s*=\s*(0x[0-9a-fA-F]+);", body)
    lit["golay_fill_end"] = int(mm.group(1), 16); where["golay_fill_end"] = f"m17_golay.cpp:{line_of(src, a0 + mm.start(1))}"
    lit["golay_unrecoverable"] = int(mm.group(2), 16); where["golay_unrecoverable"] = f"m17_golay.cpp:{line_of(src, a0 + mm.start(2))}"
    mm = re.search(r"if\(\s*bits < (\d+)\)\{", body)
    lit["golay_max_bits"] = int(mm.group(1)); where["golay_max_bits"] = f"m17_golay.cpp:{line_of(src, a0 + mm.start(1))}"

    order = ["s16_scale", "demap_offset", "demap_cor_num", "demap_sync_symbols", "disc_c", "disc_decim", "limit_num",
             "thresh_unlocked", "thresh_locked", "clk_modulus", "clk_init", "thr_init", "index_init",
             "votes_unlocked_max", "var_unlocked", "votes_locked_max", "var_locked", "n_ferror", "fclk_after_sync",
             "acm0", "traceback_mask", "golay_fill_end", "golay_unrecoverable", "golay_max_bits"]
    assert sorted(order) == sorted(lit), (set(order) ^ set(lit))
    out["rx_literals"] = {"note": "literals and control constants of the streaming arithmetic, each matched inside the statement that uses it; "
                                  "value = the literals as doubles in `order` (what m17o_get_constant / m17gpu_get_constant(\"rx_literals\") return)",
                          "order": order, "value": [float(lit[k]) for k in order], "source": {k: where[k] for k in order}}

    with open(OUT, "w") as f:
        json.dump(out, f, indent=1)
        f.write("\n")
    print("wrote", OUT, "with", len(out) - 1, "entries")


if __name__ == "__main__":
    sys.exit(main())
