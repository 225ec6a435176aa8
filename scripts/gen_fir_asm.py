#!/usr/bin/env python3
"""Writes m17_sdr_amd/csrc/m17_fir_sgpr.inc: the two inline-asm bodies of fir_window_s (m17_sync_wave.hip).

One round of the timing stage evaluates, per lane, the matched and the derivative 31-tap filter over the lane's
window of the delay line (rx_sync_filter, m17_rx_sync.cpp:25-31): acc = x0*t0, then acc += xk*tk for k = 1..30,
ascending, separate multiply and add, as one packed (matched, derivative) chain.  Tap pair k is the SGPR pair
s[36+2k : 37+2k]; the branch's 62 tap dwords are loaded by scalar loads at the head of the body (operand %[row] = the
table row), so the registers are live only inside the statement.  The window is read from LDS as sixteen aligned 8-byte pairs (ds_read_b64: 32 lanes cover 64
consecutive dwords, conflict-free), so there are two bodies:
  even: the window starts on an even dword: sample k = pair k>>1, element k&1
  odd : the window starts on an odd dword and the read starts one dword earlier: sample k = pair (k+1)>>1, element (k+1)&1
Every product is formed two instructions before the add that consumes it (the packed multiply's op_sel forwarding
needs a wait state on gfx950).  Scalar loads return out of order, so there is one lgkmcnt(0) behind the tap loads and
the window reads (the body without tap loads, with counted waits per group of four samples, is body(odd, False)).
"""
import os

BASE = 36        # first SGPR of the 62 tap registers: s[36:97] (4-aligned for the wide scalar loads; below the compiler's reserved top pair)

def body(odd, load_taps=False):
    o = 1 if odd else 0
    L = []
    if load_taps:
        # the branch's 62 tap dwords from the table row %[row], requested in front of the window reads and waited for
        # together with them (scalar loads return out of order: one lgkmcnt(0) for everything)
        for reg, n, off in ((BASE, 16, 0x0), (BASE + 16, 16, 0x40), (BASE + 32, 16, 0x80), (BASE + 48, 8, 0xc0), (BASE + 56, 4, 0xe0), (BASE + 60, 2, 0xf0)):
            L.append(f"s_load_dwordx{n} s[{reg}:{reg + n - 1}], %[row], {hex(off)}")
    for q in range(16):
        L.append(f"ds_read_b64 %[x{q}], %[a] offset:{8 * q}")
    def mul(dst, k):
        pair, half = (k + o) >> 1, (k + o) & 1
        sel = "op_sel:[1,0]" if half else "op_sel_hi:[0,1]"
        return f"v_pk_mul_f32 %[{dst}], %[x{pair}], s[{BASE + 2 * k}:{BASE + 1 + 2 * k}] {sel}"
    def add(src):
        return f"v_pk_add_f32 %[acc], %[acc], %[{src}]"
    def need(k):                                   # pairs that must have arrived before sample k is multiplied
        return ((k + o) >> 1) + 1
    def wait(k):
        if load_taps:
            return "s_waitcnt lgkmcnt(0)" if k == 3 else None
        return f"s_waitcnt lgkmcnt({16 - need(k)})"
    # samples 0..3
    L += [wait(3), mul("acc", 0), mul("q", 1), mul("p", 2), add("q"), mul("q", 3), add("p")]
    for j in range(1, 7):
        k = 4 * j
        L += [wait(k + 3), mul("p", k), add("q"), mul("q", k + 1), add("p"), mul("p", k + 2), add("q"), mul("q", k + 3), add("p")]
    L += [wait(30), mul("p", 28), add("q"), mul("q", 29), add("p"), mul("p", 30), add("q"), add("p")]
    return [ln for ln in L if ln is not None]

def body_half(odd):
    """The same filter for EIGHT waves per SIMD: 64 VGPRs and 80 SGPRs per wave (MI355X_MICROARCH.md: a 256-thread block is
    admitted 8 times per CU only with .sgpr_count <= 80; the full body's 62 tap registers alone allow six).  Window AND
    taps go through half the registers, twice: pairs 0..7 of the window in x0..x7 and tap pairs 0..15 in s[36:67]; then
    pairs 8..15 in the same VGPRs and tap pairs 16..30 in s[36:65] (s[66:67], tap pair 15, stays: with an odd window
    start sample 15 is the first of the second batch).  The second batch is requested behind the last instruction that
    reads the first; what latency that exposes is other waves' time."""
    o = 1 if odd else 0
    L = []
    for reg, n, off in ((BASE, 16, 0x0), (BASE + 16, 16, 0x40)):
        L.append(f"s_load_dwordx{n} s[{reg}:{reg + n - 1}], %[row], {hex(off)}")
    for q in range(8):
        L.append(f"ds_read_b64 %[x{q}], %[a] offset:{8 * q}")
    L.append("s_waitcnt lgkmcnt(0)")
    def tapreg(k):
        return BASE + 2 * k if k < 16 else BASE + 2 * (k - 16)
    def mul(dst, k):
        pair, half = (k + o) >> 1, (k + o) & 1
        sel = "op_sel:[1,0]" if half else "op_sel_hi:[0,1]"
        return f"v_pk_mul_f32 %[{dst}], %[x{pair & 7}], s[{tapreg(k)}:{tapreg(k) + 1}] {sel}"
    def add(src):
        return f"v_pk_add_f32 %[acc], %[acc], %[{src}]"
    # the chain in program order: product k is formed two instructions ahead of the add that consumes it
    seq = [("mul", "acc", 0), ("mul", "q", 1), ("mul", "p", 2), ("add", "q", None)]
    cur = "q"
    for k in range(3, 31):
        seq.append(("mul", cur, k)); seq.append(("add", "p" if cur == "q" else "q", None))
        cur = "p" if cur == "q" else "q"
    seq.append(("add", "p" if cur == "q" else "q", None))
    last1 = 15 - o                                   # last sample held by window pairs 0..7
    reloaded = False
    for kind, reg, k in seq:
        if kind == "mul" and k > last1 and not reloaded:
            # every register of the first batch has been read by now (tap pair 15 excepted, which is not overwritten)
            for reg2, n, off in ((BASE, 16, 0x80), (BASE + 16, 8, 0xc0), (BASE + 24, 4, 0xe0), (BASE + 28, 2, 0xf0)):
                L.append(f"s_load_dwordx{n} s[{reg2}:{reg2 + n - 1}], %[row], {hex(off)}")
            for q in range(8):
                L.append(f"ds_read_b64 %[x{q}], %[a] offset:{8 * (q + 8)}")
            L.append("s_waitcnt lgkmcnt(0)")
            reloaded = True
        L.append(mul(reg, k) if kind == "mul" else add(reg))
    return L

def emit_lines(name, lines):
    s = f"#define {name} \\\n"
    s += " \\\n".join(f'    "{ln}\\n\\t"' for ln in lines[:-1])
    s += f' \\\n    "{lines[-1]}"\n'
    return s

def emit(name, odd, load_taps=False):
    lines = body(odd, load_taps)
    s = f"#define {name} \\\n"
    s += " \\\n".join(f'    "{ln}\\n\\t"' for ln in lines[:-1])
    s += f' \\\n    "{lines[-1]}"\n'
    return s

out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "m17_sdr_amd", "csrc", "m17_fir_sgpr.inc")
with open(out, "w") as f:
    f.write("// generated by scripts/gen_fir_asm.py -- do not edit; see that script for the layout\n")
    f.write(emit("M17_FIR_SGPR_EVEN", False, True))
    f.write(emit("M17_FIR_SGPR_ODD", True, True))
    f.write(emit_lines("M17_FIR_SGPR_EVEN_H", body_half(False)))
    f.write(emit_lines("M17_FIR_SGPR_ODD_H", body_half(True)))
print("wrote", out)
