/*
 * m17_oracle.c -- CPU ORACLE (test infrastructure, NOT product code).
 *
 * Plain-C restatement of the G4GUO/m17_sdr receive hot path; see
 * m17_oracle.h for the pin status.  Build with
 *     gcc -O2 -ffp-contract=off -fno-fast-math (no -march=native)
 * so that every float expression is evaluated as separate IEEE binary32
 * operations exactly as the reference's `g++ -O3` x86-64 (SSE2, no FMA) build
 * does.
 *
 * Citations are relative to /root/reference/m17gismo/.
 */
#include <math.h>
#include <string.h>
#include <stdlib.h>
#include "m17_oracle.h"

/* The literals and control constants of the reference's streaming arithmetic, each under ONE name: the code below uses
 * these names where the reference has the literal, and m17o_get_constant("rx_literals") returns them in this order --
 * tests/test_ref_constants.py holds that array against the values found in the reference's source text
 * (tests/golden/extract_ref_constants.py) and against the product's. */
#define M17O_LIT_S16_SCALE           0.00003   /* m17_dsp.cpp:138-139 */
#define M17O_LIT_DEMAP_OFFSET        0.6666    /* m17_dsp.cpp:41 */
#define M17O_LIT_DEMAP_COR_NUM       8.0       /* m17_dsp.cpp:88 */
#define M17O_LIT_DEMAP_SYNC_SYMBOLS  8         /* m17_dsp.cpp:85 */
#define M17O_LIT_DISC_C              0.5f      /* m17_dsp.cpp:199 */
#define M17O_LIT_DISC_DECIM          5         /* m17_dsp.cpp:207 */
#define M17O_LIT_LIMIT_NUM           1.0       /* m17_dsp.cpp:415 */
#define M17O_LIT_THRESH_UNLOCKED     10        /* m17_rx_sync.cpp:93 */
#define M17O_LIT_THRESH_LOCKED       80        /* m17_rx_sync.cpp:95 */
#define M17O_LIT_CLK_MODULUS         2         /* m17_rx_sync.cpp:82 */
#define M17O_LIT_CLK_INIT            1         /* m17_rx_sync.cpp:123 */
#define M17O_LIT_THR_INIT            0         /* m17_rx_sync.cpp:125 */
#define M17O_LIT_INDEX_INIT          10        /* m17_rx_sync.cpp:126 */
#define M17O_LIT_VOTES_UNLOCKED_MAX  0         /* m17_rx_frame.cpp:83 */
#define M17O_LIT_VAR_UNLOCKED        0.3       /* m17_rx_frame.cpp:87 */
#define M17O_LIT_VOTES_LOCKED_MAX    1         /* m17_rx_frame.cpp:94 */
#define M17O_LIT_VAR_LOCKED          0.5       /* m17_rx_frame.cpp:98 */
#define M17O_LIT_N_FERROR            5         /* m17_rx_frame.cpp:122 */
#define M17O_LIT_FCLK_AFTER_SYNC     8         /* m17_rx_frame.cpp:166 */
#define M17O_LIT_ACM0                1.0f      /* m17_conv.cpp:153 */
#define M17O_LIT_TRACEBACK_MASK      0x08      /* m17_conv.cpp:165 */
#define M17O_LIT_GOLAY_FILL_END      0xFFF     /* m17_golay.cpp:53 */
#define M17O_LIT_GOLAY_UNRECOVERABLE 0x400     /* m17_golay.cpp:54 */
#define M17O_LIT_GOLAY_MAX_BITS      5         /* m17_golay.cpp:61 */
static const double t_rx_literals[24] = {
    M17O_LIT_S16_SCALE, M17O_LIT_DEMAP_OFFSET, M17O_LIT_DEMAP_COR_NUM, M17O_LIT_DEMAP_SYNC_SYMBOLS, M17O_LIT_DISC_C,
    M17O_LIT_DISC_DECIM, M17O_LIT_LIMIT_NUM, M17O_LIT_THRESH_UNLOCKED, M17O_LIT_THRESH_LOCKED, M17O_LIT_CLK_MODULUS,
    M17O_LIT_CLK_INIT, M17O_LIT_THR_INIT, M17O_LIT_INDEX_INIT, M17O_LIT_VOTES_UNLOCKED_MAX, M17O_LIT_VAR_UNLOCKED,
    M17O_LIT_VOTES_LOCKED_MAX, M17O_LIT_VAR_LOCKED, M17O_LIT_N_FERROR, M17O_LIT_FCLK_AFTER_SYNC, M17O_LIT_ACM0,
    M17O_LIT_TRACEBACK_MASK, M17O_LIT_GOLAY_FILL_END, M17O_LIT_GOLAY_UNRECOVERABLE, M17O_LIT_GOLAY_MAX_BITS };


/* ------------------------------------------------------------------ */
/* tables                                                             */
/* ------------------------------------------------------------------ */
static uint16_t t_crc[256];            /* m17_crc.cpp:6 */
static uint8_t  t_clut[32][2];         /* m17_conv.cpp:22 */
static uint8_t  t_bf[16][4];           /* m17_conv.cpp:93-108 as {w,x,y,z} */
static uint8_t  t_derand[368];         /* m17_correlate.cpp:9 */
static uint16_t t_genc[4096];          /* m17_golay.cpp:19 */
static uint16_t t_gerr[4096];          /* m17_golay.cpp:28 */
static float    t_mf[M17O_NF][M17O_FN];/* m17_rx_sync.cpp:12 */
static float    t_md[M17O_NF][M17O_FN];/* m17_rx_sync.cpp:13 */
static int      t_init_done;

/* M17 sync templates, +-1 per symbol (m17_rx_frame.cpp:5-12) */
static const float t_sframe[6][8] = {
    { 1,-1, 1,-1, 1,-1, 1,-1},   /* 0 preamble            */
    { 1, 1, 1, 1,-1,-1, 1,-1},   /* 1 link setup  0x55F7  */
    {-1,-1,-1,-1, 1, 1,-1, 1},   /* 2 stream      0xFF5D  */
    { 1,-1, 1, 1,-1,-1,-1,-1},   /* 3 packet      0x75FF  */
    {-1, 1,-1,-1, 1, 1, 1, 1},   /* 4 BERT        0xDF55  */
    { 1, 1, 1, 1, 1, 1,-1, 1}    /* 5 EOT                 */
};

/* de-randomiser bytes: M17 spec randomising sequence (m17_correlate.cpp:3-7) */
static const uint8_t t_ctab[46] = {
    0xD6,0xB5,0xE2,0x30,0x82,0xFF,0x84,0x62,0xBA,0x4E,0x96,0x90,0xD8,0x98,0xDD,0x5D,
    0x0C,0xC8,0x52,0x43,0x91,0x1D,0xF8,0x6E,0x68,0x2F,0x35,0xDA,0x14,0xEA,0xCD,0x76,
    0x19,0x8D,0xD5,0x80,0xD1,0x33,0x87,0x13,0x57,0x18,0x2D,0x29,0x78,0xC3 };

/* Golay(24,12) parity generator rows, M17 spec (m17_golay.cpp:11) */
static const uint16_t t_gtab[12] = {
    0xC75,0x63B,0xF68,0x7B4,0x3DA,0xD99,0x6CD,0x367,0xDC6,0xA97,0x93E,0x8EB };

/* puncture patterns, M17 spec (m17_puncture.cpp:4-10) */
static const uint8_t t_p1[61] = {
    1,1,0,1,1,1,0,1,1,1,0,1,1,1,0,1,1,1,0,1,1,1,0,1,1,1,0,1,1,1,0,1,1,1,0,1,1,1,0,1,
    1,1,0,1,1,1,0,1,1,1,0,1,1,1,0,1,1,1,0,1,1 };
static const uint8_t t_p2[12] = {1,1,1,1,1,1,1,1,1,1,1,0};
static const uint8_t t_p3[8]  = {1,1,1,1,1,1,1,0};

/* m17_crc.cpp:8-22 : poly 0x5935, MSB first */
static void build_crc(void)
{
    for (int i = 0; i < 256; i++) {
        uint16_t x = (uint16_t)(i << 8);
        for (int n = 0; n < 8; n++)
            x = (x & 0x8000) ? (uint16_t)((x << 1) ^ 0x5935) : (uint16_t)(x << 1);
        t_crc[i] = x;
    }
}

/* m17_conv.cpp:24-29 (G1 = 0x19, G2 = 0x17 over a 5-bit register) and the
 * butterfly table :93-108, derived the way the commented generator :117-144 did */
static void build_conv(void)
{
    for (int i = 0; i < 32; i++) {
        t_clut[i][0] = ((i >> 4) ^ (i >> 1) ^ i) & 1;
        t_clut[i][1] = ((i >> 4) ^ (i >> 3) ^ (i >> 2) ^ i) & 1;
    }
    for (int v = 0; v < 16; v++) {
        int s0 = (v << 1), s1 = (v << 1) + 1;
        t_bf[v][0] = (uint8_t)(s0 & 0xF);
        t_bf[v][1] = (uint8_t)((t_clut[s0][0] << 1) + t_clut[s0][1]);
        t_bf[v][2] = (uint8_t)(s1 & 0xF);
        t_bf[v][3] = (uint8_t)((t_clut[s1][0] << 1) + t_clut[s1][1]);
    }
}

/* m17_correlate.cpp:35-42 */
static void build_derand(void)
{
    int idx = 0;
    for (int i = 0; i < 46; i++)
        for (int n = 0x80; n; n >>= 1)
            t_derand[idx++] = (t_ctab[i] & n) ? 1 : 0;
}

/* m17_golay.cpp:31-72.  The error table is filled by visiting all 2^24 words
 * in ascending order and letting the last writer win, exactly like :57-71, so
 * weight-4 syndromes resolve to the numerically largest pattern. */
static void build_golay(void)
{
    for (int i = 0; i < 0x1000; i++) {
        uint16_t p = 0; int m = 0x800;
        for (int n = 0; n < 12; n++) { if (i & m) p ^= t_gtab[n]; m >>= 1; }
        t_genc[i] = p;
    }
    memset(t_gerr, 0, sizeof t_gerr);
    for (int i = 0; i < M17O_LIT_GOLAY_FILL_END; i++) t_gerr[i] = M17O_LIT_GOLAY_UNRECOVERABLE;       /* :53-55 as written */
    for (uint32_t w = 0; w < 0x1000000u; w++) {
        int bits = __builtin_popcount(w);
        if (bits < M17O_LIT_GOLAY_MAX_BITS) {
            uint16_t data = (uint16_t)(w >> 12), par = (uint16_t)(w & 0xFFF);
            uint16_t syn = par ^ t_genc[data];
            t_gerr[syn] = (uint16_t)((bits << 12) | data);
        }
    }
}

/* m17_dsp.cpp:295-315 */
void m17o_build_rrc_filter(float *filter, float rolloff, int ntaps, int sps)
{
    double a, b, c, d;
    double B = (rolloff + 0.0001);
    double t = -(ntaps - 1) / 2;          /* integer division on purpose (:298) */
    double Ts = sps;
    for (int i = 0; i < ntaps; i++) {
        a = 2.0 * B / (M_PI * sqrt(Ts));
        b = cos((1.0 + B) * M_PI * t / Ts);
        if (t == 0)
            c = (1.0 - B) * M_PI / (4 * B);
        else
            c = sin((1.0 - B) * M_PI * t / Ts) / (4.0 * B * t / Ts);
        d = (1.0 - (4.0 * B * t / Ts) * (4.0 * B * t / Ts));
        filter[i] = (float)(a * (b + c) / d);
        t = t + 1.0;
    }
}

/* m17_dsp.cpp:420-429 */
void m17o_set_filter_gain(float *f, float gain, int stride, int ntaps)
{
    float sum = 0;
    for (int i = 0; i < ntaps; i++) sum += f[i * stride];
    gain = gain / sum;
    for (int i = 0; i < ntaps; i++) f[i * stride] = f[i * stride] * gain;
}

/* m17_rx_sync.cpp:101-122 */
static void build_sync_filters(void)
{
    enum { N = M17O_NF * M17O_FN };
    static float mf[N], md[N];
    m17o_build_rrc_filter(mf, 0.5f, N, M17O_NF * 2);
    for (int i = 0; i < N; i++)
        md[i] = mf[(i + 1) % N] - mf[(i + N - 1) % N];
    for (int i = 0; i < M17O_NF; i++)
        for (int j = 0; j < M17O_FN; j++) {
            t_mf[i][j] = mf[i + j * M17O_NF];
            t_md[i][j] = md[i + j * M17O_NF];
        }
    for (int i = 0; i < M17O_NF; i++)
        m17o_set_filter_gain(t_mf[i], 1.0f, 1, M17O_FN);
}

void m17o_init(void)
{
    if (t_init_done) return;
    build_crc();
    build_conv();
    build_derand();
    build_golay();
    build_sync_filters();
    t_init_done = 1;
}

const float *m17o_tab_mf(void) { return &t_mf[0][0]; }
const float *m17o_tab_md(void) { return &t_md[0][0]; }
const uint16_t *m17o_tab_golay_enc(void) { return t_genc; }
const uint16_t *m17o_tab_golay_err(void) { return t_gerr; }
const uint8_t *m17o_tab_derand(void) { return t_derand; }
const uint16_t *m17o_tab_crc(void) { return t_crc; }

/* The literal constants this restatement is built from, by name, for the test that holds them
 * against the values extracted from the reference's source text (tests/golden/ref_constants.json).
 * Returns the number of bytes written, or -1. */
int m17o_get_constant(const char *name, void *out, int cap)
{
    uint8_t buf[256];
    int n = 0;
    m17o_init();
    if (!strcmp(name, "rx_literals")) { memcpy(buf, t_rx_literals, sizeof t_rx_literals); n = sizeof t_rx_literals; } else
    if (!strcmp(name, "sframe")) { memcpy(buf, t_sframe, sizeof t_sframe); n = sizeof t_sframe; }
    else if (!strcmp(name, "derand_bytes")) { memcpy(buf, t_ctab, 46); n = 46; }
    else if (!strcmp(name, "golay_rows")) { memcpy(buf, t_gtab, sizeof t_gtab); n = sizeof t_gtab; }
    else if (!strcmp(name, "punc1")) { memcpy(buf, t_p1, 61); n = 61; }
    else if (!strcmp(name, "punc2")) { memcpy(buf, t_p2, 12); n = 12; }
    else if (!strcmp(name, "punc3")) { memcpy(buf, t_p3, 8); n = 8; }
    else if (!strcmp(name, "butterfly")) {          /* BF(v,w,x,y,z) rows */
        for (int v = 0; v < 16; v++) {
            buf[n++] = (uint8_t)v;
            for (int k = 0; k < 4; k++) buf[n++] = t_bf[v][k];
        }
    }
    else if (!strcmp(name, "crc_poly")) { memcpy(buf, &t_crc[1], 2); n = 2; }  /* MSB-first table: entry 1 is the polynomial */
    else return -1;
    if (n > cap) return -1;
    memcpy(out, buf, (size_t)n);
    return n;
}
int m17o_sizeof_chan(void) { return (int)sizeof(m17o_chan); }

/* zero-initialised statics + m17_rx_sync.cpp:123-126 */
void m17o_chan_reset(m17o_chan *st)
{
    memset(st, 0, sizeof *st);
    st->m_clk = M17O_LIT_CLK_INIT;
    st->m_thr = M17O_LIT_THR_INIT;
    st->m_index = M17O_LIT_INDEX_INIT;
}

/* ------------------------------------------------------------------ */
/* codec primitives                                                   */
/* ------------------------------------------------------------------ */
/* m17_crc.cpp:26-35 */
uint16_t m17o_crc(const uint8_t *in, int len)
{
    uint16_t crc = 0xFFFF;
    for (int i = 0; i < len; i++) {
        uint8_t pos = (uint8_t)((crc >> 8) ^ in[i]);
        crc = (uint16_t)((crc << 8) ^ t_crc[pos]);
    }
    return crc;
}

/* m17_golay.cpp:94-102 */
uint32_t m17o_golay_encode(uint16_t data)
{
    return ((uint32_t)data << 12) | t_genc[data & 0xFFF];
}

/* m17_golay.cpp:103-116 */
int m17o_golay_decode(uint32_t word, uint16_t *odata)
{
    uint16_t data = (word >> 12) & 0xFFF, par = word & 0xFFF;
    uint16_t syn = par ^ t_genc[data];
    *odata = data ^ (t_gerr[syn] & 0xFFF);
    return (t_gerr[syn] & 0xF000) >> 12;
}

/* m17_conv.cpp:53-71 */
int m17o_conv_encode_8(const uint8_t *in, uint8_t *out, int len)
{
    int idx = 0; uint8_t sr = 0;
    for (int i = 0; i < len; i++)
        for (int n = 0x80; n; n >>= 1) {
            if (in[i] & n) sr |= 0x10;
            out[idx++] = t_clut[sr][0];
            out[idx++] = t_clut[sr][1];
            sr >>= 1;
        }
    for (int i = 0; i < 4; i++) {
        out[idx++] = t_clut[sr][0];
        out[idx++] = t_clut[sr][1];
        sr >>= 1;
    }
    return idx;
}

/* m17_conv.cpp:33-49 */
int m17o_conv_encode_1(const uint8_t *in, uint8_t *out, int len)
{
    int idx = 0; uint8_t sr = 0;
    for (int i = 0; i < len; i++) {
        if (in[i]) sr |= 0x10;
        out[idx++] = t_clut[sr][0];
        out[idx++] = t_clut[sr][1];
        sr >>= 1;
    }
    for (int i = 0; i < 4; i++) {
        out[idx++] = t_clut[sr][0];
        out[idx++] = t_clut[sr][1];
        sr >>= 1;
    }
    return idx;
}

static const uint8_t *punc_tab(int which, int *period)
{
    switch (which) {
    case 1: *period = 61; return t_p1;
    case 2: *period = 12; return t_p2;
    default: *period = 8; return t_p3;
    }
}

/* m17_puncture.cpp:12-41; len = input length */
int m17o_punc(int which, const uint8_t *in, uint8_t *out, int len)
{
    int per; const uint8_t *p = punc_tab(which, &per);
    int idx = 0;
    for (int i = 0; i < len; i++)
        if (p[i % per]) out[idx++] = in[i];
    return idx;
}

/* m17_puncture.cpp:47-79; len = OUTPUT length, erasures become 0.0f */
int m17o_de_punc(int which, const float *in, float *out, int len)
{
    int per; const uint8_t *p = punc_tab(which, &per);
    int odx = 0, idx = 0;
    for (int i = 0; i < len; i++) {
        if (p[i % per]) out[odx++] = in[idx++];
        else            out[odx++] = 0.0f;
    }
    return odx;
}

/* m17_interleave.cpp:3-7 */
void m17o_interleave_u8(const uint8_t *in, uint8_t *out, int len)
{
    for (int i = 0; i < len; i++) out[((i * 45) + (92 * i * i)) % 368] = in[i];
}

/* m17_interleave.cpp:8-12 */
void m17o_de_interleave(const float *in, float *out, int len)
{
    for (int i = 0; i < len; i++) out[((i * 45) + (92 * i * i)) % 368] = in[i];
}

/* m17_correlate.cpp:16-20 */
void m17o_de_correlate_u8(const uint8_t *in, uint8_t *out, int len)
{
    for (int i = 0; i < len; i++) out[i] = (in[i] ^ t_derand[i % 368]) & 1;
}

/* m17_correlate.cpp:27-31 (in == out allowed) */
void m17o_de_correlate_f(const float *in, float *out, int len)
{
    for (int i = 0; i < len; i++) out[i] = t_derand[i % 368] ? -in[i] : in[i];
}

/* m17_conv.cpp:73-113 + :148-168.  Correlation metric, strict '>' keeps the
 * even predecessor, ties go to the odd one; traceback from state 0 emitting the
 * MSB of the predecessor (so out[0] is always 0). */
int m17o_viterbi_decode(const float *in, uint8_t *out, int len)
{
    float acm[16], tm[16];
    static __thread uint8_t path[16][1024];
    int hp = 0;
    memset(acm, 0, sizeof acm);
    acm[0] = M17O_LIT_ACM0;
    for (int i = 0; i < len; i += 2) {
        float m1 = in[i], m2 = in[i + 1];
        float metric1X = m1, metric0X = -m1, metricX1 = m2, metricX0 = -m2;
        float metric[4];
        metric[0] = (metric0X + metricX0);
        metric[1] = (metric0X + metricX1);
        metric[2] = (metric1X + metricX0);
        metric[3] = (metric1X + metricX1);
        for (int v = 0; v < 16; v++) {
            int w = t_bf[v][0], x = t_bf[v][1], y = t_bf[v][2], z = t_bf[v][3];
            float ta = acm[w] + metric[x];
            float tb = acm[y] + metric[z];
            if (ta > tb) { tm[v] = ta; path[v][hp] = (uint8_t)w; }
            else         { tm[v] = tb; path[v][hp] = (uint8_t)y; }
        }
        for (int v = 0; v < 16; v++) acm[v] = tm[v];
        hp++;
    }
    uint8_t state = 0;
    for (int i = hp - 1; i >= 0; i--) {
        state = path[state][i];
        out[i] = (state & M17O_LIT_TRACEBACK_MASK) ? 1 : 0;
    }
    return hp;
}

/* m17_bit_utils.cpp:180-187 */
uint32_t m17o_hard_decode_24(const float *in)
{
    uint32_t w = 0;
    for (int i = 0; i < 24; i++) { w <<= 1; w |= (in[i] >= 0) ? 1 : 0; }
    return w;
}

/* m17_bit_utils.cpp:26-32 */
int m17o_pack_1_to_8(const uint8_t *in, uint8_t *out, int len)
{
    int idx = 0;
    for (int i = 0; i < len; i += 8)
        out[idx++] = (uint8_t)((in[i] << 7) | (in[i+1] << 6) | (in[i+2] << 5) | (in[i+3] << 4) |
                               (in[i+4] << 3) | (in[i+5] << 2) | (in[i+6] << 1) | in[i+7]);
    return idx;
}

/* m17_dsp.cpp:35-42 + :82-95.  fabs() on a float picks the float overload in
 * the reference's C++; `fabs(m) - 0.6666` and `8.0/sum` are double. */
void m17o_demap_frame(const float *in, float *out)
{
    float sum = 0;
    for (int i = 0; i < M17O_LIT_DEMAP_SYNC_SYMBOLS; i++) sum += fabsf(in[i]);
    float cor = (float)(M17O_LIT_DEMAP_COR_NUM / (double)sum);
    int idx = 0;
    for (int i = 8; i < M17O_FRAME_SYMS; i++) {
        float m = in[i] * cor;
        out[idx]     = -m;
        out[idx + 1] = (float)((double)fabsf(m) - M17O_LIT_DEMAP_OFFSET);
        idx += 2;
    }
}

/* m17_bit_utils.cpp:191-208 */
uint64_t m17o_encode_call(const char *call)
{
    uint64_t word = 0;
    for (int i = 8; i >= 0; i--) {
        word *= 40;
        char ch = call[i];
        if (ch >= 'A' && ch <= 'Z') word += (uint64_t)(ch - 'A' + 1);
        else if (ch >= '0' && ch <= '9') word += (uint64_t)(ch - '0' + 27);
        else {
            if (ch == '-') word += 37;
            if (ch == '/') word += 38;
            if (ch == '.') word += 39;
        }
    }
    return word;
}

/* m17_bit_utils.cpp:209-226 */
void m17o_decode_call(uint64_t word, char *call)
{
    if (word == 0xFFFFFFFFFFFFull) { strcpy(call, "BROADCAST"); return; }
    for (int i = 8; i >= 0; i--) {
        int ch = (int)(word % 40);
        if (ch == 0) call[8 - i] = ' ';
        if (ch == 37) call[8 - i] = '-';
        if (ch == 38) call[8 - i] = '/';
        if (ch == 39) call[8 - i] = '.';
        if (ch >= 1 && ch <= 26) call[8 - i] = (char)(ch + 'A' - 1);
        if (ch >= 27 && ch <= 36) call[8 - i] = (char)(ch + '0' - 27);
        word /= 40;
    }
    call[9] = 0;
}

/* parse_lsf (m17_rx_parse.cpp:52-70): the fields the reference hands to valid_lsf_received / gui_save_dest_address /
 * gui_save_src_address -- destination and source by pack_8_to_48 (m17_bit_utils.cpp:100-114, big endian), the type
 * word by pack_8_to_16 (:125-131) through m17_upack_type (:245-254), the 14 meta bytes; the callsign texts are what
 * the GUI makes of the addresses with m17_decode_call (:209-226).  crc / crc_ok: lsf[28..29] and the test
 * update_lich applies before it calls parse_lsf (m17_rx_parse.cpp:79-82: CRC over all 30 bytes is 0).  The struct is
 * zeroed first so that two parsers can be compared byte for byte. */
void m17o_parse_lsf(const uint8_t *in, m17o_lsf_fields *out)
{
    memset(out, 0, sizeof *out);
    uint64_t dst_add = 0, src_add = 0;
    for (int i = 0; i < 6; i++) { dst_add <<= 8; dst_add |= in[i]; }
    for (int i = 0; i < 6; i++) { src_add <<= 8; src_add |= in[6 + i]; }
    uint16_t tw = in[12];
    tw = (uint16_t)((tw << 8) | in[13]);
    out->dst = dst_add;
    out->src = src_add;
    m17o_decode_call(dst_add, out->dst_call);
    m17o_decode_call(src_add, out->src_call);
    out->reserved = (tw >> 11) & 0x1F;
    out->can = (tw >> 7) & 0xF;
    out->est = (tw >> 5) & 0x3;
    out->et = (tw >> 3) & 0x3;
    out->dt = (tw >> 1) & 0x3;
    out->p_s = tw & 0x1;
    memcpy(out->meta, &in[14], 14);
    out->crc = (uint16_t)((in[28] << 8) | in[29]);
    out->crc_ok = m17o_crc(in, 30) == 0;
}

/* m17_prbs9.cpp:16-32 : x^9 + x^5 + 1, start 0x001 */
void m17o_prbs9(uint8_t *out, int len)
{
    uint16_t sr = 1;
    for (int i = 0; i < len; i++) {
        uint8_t bit = ((sr >> 8) ^ (sr >> 4)) & 1;
        sr = (uint16_t)(((sr << 1) | bit) & 0x1FF);
        out[i] = bit;
    }
}

/* ------------------------------------------------------------------ */
/* streaming stages                                                   */
/* ------------------------------------------------------------------ */
/* m17_dsp.cpp:136-141 (int16 * 0.00003 in double, rounded to float),
 * :412-419 (limiter: float sqrt, reciprocal in double), :194-222 (discriminator,
 * /5 pick, sequential DC sum over all 1920 samples, DC removal). */
void m17o_frontend(m17o_chan *st, const int16_t *iq, float *d, float *d_raw, float *offset_out)
{
    float offset = 0;
    int idx = 0;
    float z0re = st->z0re, z0im = st->z0im, z1re = st->z1re, z1im = st->z1im;
    int count = st->disc_count;
    /* AFC, only when switched on (m17_dsp.cpp:468): radio_get_afc_delta() radio.cpp:201-208 -- outside a
     * frame the correction is dropped; dsp_nco_mixer m17_dsp.cpp:390-408 with its static double phase */
    float delta = 0;
    double acc = st->afc_acc;
    if (st->afc_on) {
        if (st->in_frame) delta = st->afc_delta;
        else st->afc_delta = 0;
    }
    for (int i = 0; i < M17O_BLOCK_SAMPLES; i++) {
        float re = (float)((double)iq[2 * i] * M17O_LIT_S16_SCALE);
        float im = (float)((double)iq[2 * i + 1] * M17O_LIT_S16_SCALE);
        if (st->afc_on) {
            float c = (float)cos(acc);
            float s = (float)sin(acc);
            acc += delta;
            float mre = (re * c) - (im * s);
            float mim = (re * s) + (im * c);
            re = mre; im = mim;
        }
        float m = sqrtf(re * re + im * im);
        float g = (float)(M17O_LIT_LIMIT_NUM / (double)m);
        re = re * g;
        im = im * g;
        float a = z0im * (re - z1re);
        float b = z0re * (im - z1im);
        float u = b - a;
        z1re = z0re; z1im = z0im;
        z0re = re;   z0im = im;
        count = (count + 1) % M17O_LIT_DISC_DECIM;
        if (count == 0) d[idx++] = u * M17O_LIT_DISC_C;
        offset += u * M17O_LIT_DISC_C;
    }
    if (st->afc_on) {
        double ip;
        acc = acc / (2.0 * M_PI);               /* :402-407 */
        acc = modf(acc, &ip);
        acc = acc * 2.0 * M_PI;
        if (acc != acc) acc = 0;
        st->afc_acc = acc;
    }
    st->z0re = z0re; st->z0im = z0im; st->z1re = z1re; st->z1im = z1im;
    st->disc_count = count;
    if (d_raw) memcpy(d_raw, d, sizeof(float) * (size_t)idx);
    offset = offset / M17O_BLOCK_SAMPLES;
    /* radio_afc(offset), radio.cpp:196-200: float m_afc_delta, double arithmetic */
    if (st->afc_on && st->in_frame) st->afc_delta = (float)((double)st->afc_delta - (double)offset * 0.1);
    for (int i = 0; i < idx; i++) d[i] = d[i] - offset;
    if (offset_out) *offset_out = offset;
}

void m17o_set_afc(m17o_chan *st, int on) { st->afc_on = on ? 1 : 0; if (!on) { st->afc_delta = 0; } }
void m17o_get_afc(const m17o_chan *st, float *delta, double *acc) { if (delta) *delta = st->afc_delta; if (acc) *acc = st->afc_acc; }

/* m17_rx_sync.cpp:25-31 */
static float sync_filter(const float *in, const float *c)
{
    float sum = in[0] * c[0];
    for (int i = 1; i < M17O_FN; i++) sum += in[i] * c[i];
    return sum;
}

/* m17_rx_sync.cpp:77-99 with :32-42 and :45-72.
 * `out[-1]` (m_idx-- at m_idx==0, :69) is undefined behaviour in the reference;
 * here the symbol written to index -1 is discarded. */
int m17o_rx_sync_samples(m17o_chan *st, const float *in, float *out, int len)
{
    int m_idx = 0;
    for (int i = 0; i < len; i++) {
        for (int k = 0; k < M17O_FN - 1; k++) st->m_buff[k] = st->m_buff[k + 1];
        st->m_buff[M17O_FN - 1] = in[i];
        st->m_clk = (st->m_clk + 1) % M17O_LIT_CLK_MODULUS;
        if (st->m_clk) {
            st->sum = sync_filter(st->m_buff, t_mf[st->m_index]);
            st->dif = sync_filter(st->m_buff, t_md[st->m_index]);
            if (m_idx >= 0) out[m_idx] = st->sum;
            m_idx++;
        } else {
            float sum = st->sum, dif = st->dif;
            if (sum < 0) dif = -dif;
            if (dif > 0) st->m_thr++;
            if (dif < 0) st->m_thr--;
            int thresh = st->m_flock ? M17O_LIT_THRESH_LOCKED : M17O_LIT_THRESH_UNLOCKED;
            if (st->m_thr > thresh) {
                st->m_index = (st->m_index + 1) % M17O_NF;
                st->m_thr = 0;
                if (st->m_index == 0) {
                    st->m_clk = 1;
                    if (m_idx >= 0) out[m_idx] = 0;
                    m_idx++;
                }
            }
            if (st->m_thr < -thresh) {
                st->m_thr = 0;
                st->m_index = (st->m_index + M17O_NF - 1) % M17O_NF;
                if (st->m_index == (M17O_NF - 1)) {
                    st->m_clk = 1;
                    m_idx--;
                }
            }
        }
    }
    return m_idx;
}

/* m17_rx_frame.cpp:22-43 */
static float find_variance(const float *in, int len)
{
    float v, mmin, mmax;
    mmin = fabsf(in[0]);
    mmax = mmin;
    for (int i = 1; i < len; i++) {
        v = fabsf(in[i]);
        if (v > mmax) mmax = v;
        else if (v < mmin) mmin = v;
    }
    v = (mmax - mmin) / mmax;
    if (v != v) v = 1.0f;
    return v;
}

/* m17_rx_frame.cpp:47-81 */
void m17o_sync_check(const float *vect, uint8_t *type, uint8_t *votes, float *variance)
{
    float sums[6];
    for (int k = 0; k < 6; k++) sums[k] = vect[0] * t_sframe[k][0];
    for (int i = 1; i < 8; i++)
        for (int k = 0; k < 6; k++) sums[k] += vect[i] * t_sframe[k][i];
    *variance = find_variance(vect, 8);
    float mmax = 0; int nmax = 0;
    for (int i = 0; i < 6; i++)
        if (sums[i] > mmax) { mmax = sums[i]; nmax = i; }
    *type = (uint8_t)nmax;
    int v = 0;
    for (int i = 0; i < 8; i++)
        if (vect[i] * t_sframe[nmax][i] < 0) v++;
    *votes = (uint8_t)v;
}

/* m17_rx_frame.cpp:82-103; the literals 0.3 / 0.5 are doubles there */
static int sync_ok(uint8_t type, uint8_t votes, float variance, int locked)
{
    if (votes > (locked ? M17O_LIT_VOTES_LOCKED_MAX : M17O_LIT_VOTES_UNLOCKED_MAX)) return 0;
    if (type == 1 || type == 2 || type == 3 || type == 4)
        if ((double)variance < (locked ? M17O_LIT_VAR_LOCKED : M17O_LIT_VAR_UNLOCKED)) return 1;
    return 0;
}

/* m17_rx_parse.cpp:71-85 */
static int update_lich(m17o_chan *st, const uint8_t *in)
{
    int seq = in[5] >> 5;
    if (seq < 6) {
        memcpy(&st->m_lsf[0][seq * 5], in, 5);
        if (m17o_crc(st->m_lsf[0], 30) == 0) {
            memcpy(st->m_lsf[1], st->m_lsf[0], 30);
            return 1;
        }
    }
    return 0;
}

/* m17_rx_parse.cpp:34-51.  The reference can run 6 bytes past m_packet[800]
 * for an (invalid) fn > 25 at EOF; the copy is clamped here. */
static int parse_packet(m17o_chan *st, const uint8_t *data, uint8_t eof, uint8_t fn)
{
    int valid = 0;
    if (eof) {
        int n = fn;
        if (st->m_packet_idx + n > 800) n = 800 - st->m_packet_idx;
        memcpy(&st->m_packet[st->m_packet_idx], data, (size_t)n);
        st->m_packet_idx += n;
        if (m17o_crc(st->m_packet, st->m_packet_idx) == 0) valid = 1;
        st->m_packet_idx = 0;
    } else {
        memcpy(&st->m_packet[fn * 25], data, 25);
        st->m_packet_idx = fn * 25;
    }
    return valid;
}

/* The network sink of decode_stream_frame (m17_rx_parse.cpp:151-154): m17_net_new_rx_data(m_frame_id, m_lsf[1], fn,
 * data) builds the 54-byte M17-over-IP frame "M17 " | stream id | LSF bytes 0..27 | FN | 16 payload bytes | CRC-16 of
 * the first 52 bytes (net_add_* m17_net.cpp:25-49, :53-74; the destination callsign optionally overwritten as :60-62
 * does with the reflector's).  Not mirrored: the reference copies 54 bytes out of the 30-byte m_lsf[1] (:58), of which
 * only the first 28 reach the frame.  m_frame_id is rand() there; here it is stream_id_base + frame_id_epoch, the
 * event counter that replaces it.  The batch driver below points the sink at the channel's rows. */
typedef struct { uint8_t *net; const m17o_rec *recs; int cap; uint16_t sid_base; uint64_t dst_override; } net_sink;
static __thread net_sink t_sink;

void m17o_format_net_frame(uint16_t stream_id, const uint8_t *lsf, uint16_t fn, const uint8_t *pld,
                           uint64_t dst_override, uint8_t *out)
{
    out[0] = 0x4D; out[1] = 0x31; out[2] = 0x37; out[3] = 0x20;
    out[4] = (uint8_t)(stream_id >> 8); out[5] = (uint8_t)(stream_id & 0xFF);
    memcpy(&out[6], lsf, 28);
    if (dst_override)
        for (int i = 0; i < 6; i++) out[6 + i] = (uint8_t)(dst_override >> (40 - 8 * i));     /* pack_48_to_8 */
    out[34] = (uint8_t)(fn >> 8); out[35] = (uint8_t)(fn & 0xFF);
    memcpy(&out[36], pld, 16);
    uint16_t crc = m17o_crc(out, 52);
    out[52] = (uint8_t)(crc >> 8); out[53] = (uint8_t)(crc & 0xFF);
}

/* m17_rx_parse.cpp:185-226 and the three decoders :86-177 */
void m17o_rx_parse(m17o_chan *st, const float *s, uint8_t type, m17o_rec *r)
{
    float sb[384];
    float so0[488], so1[488];
    uint8_t bits[244];
    r->flags |= M17O_F_PARSED;
    switch (type) {
    case 0: case 5:
        st->frame_id_epoch++;                       /* generate_new_frame_id() */
        break;
    case 1: {                                       /* decode_link_frame :86-101 */
        m17o_demap_frame(s, sb);
        m17o_de_correlate_f(sb, sb, 368);
        m17o_de_interleave(sb, so0, 368);
        m17o_de_punc(1, so0, so1, 488);
        m17o_viterbi_decode(so1, bits, 488);
        m17o_pack_1_to_8(&bits[1], r->data, 240);
        if (m17o_crc(st->m_packet, 30) == 0) r->flags |= M17O_F_LSF_GATE;   /* :98 quirk */
        break; }
    case 2: {                                       /* decode_stream_frame :105-160 */
        uint16_t w[4]; int e = 0;
        m17o_demap_frame(s, sb);
        m17o_de_correlate_f(sb, sb, 368);
        m17o_de_interleave(sb, so0, 368);
        for (int k = 0; k < 4; k++)
            e += m17o_golay_decode(m17o_hard_decode_24(&so0[24 * k]), &w[k]);
        st->g_errors += (uint32_t)e;                /* m17_dbase.cpp:79-82 */
        st->n_frames++;
        r->golay_errs = (uint8_t)e;
        /* pack_12_to_8_x4x6, m17_bit_utils.cpp:152-172 */
        uint32_t ww = ((uint32_t)w[0] << 12) | w[1];
        r->data[0] = (ww >> 16) & 0xFF; r->data[1] = (ww >> 8) & 0xFF; r->data[2] = ww & 0xFF;
        ww = ((uint32_t)w[2] << 12) | w[3];
        r->data[3] = (ww >> 16) & 0xFF; r->data[4] = (ww >> 8) & 0xFF; r->data[5] = ww & 0xFF;
        if (update_lich(st, r->data)) r->flags |= M17O_F_LICH_OK;
        m17o_de_punc(2, &so0[96], so1, 296);
        m17o_viterbi_decode(so1, bits, 296);
        m17o_pack_1_to_8(&bits[1], &r->data[6], 144);
        r->fn = (uint16_t)((r->data[6] << 8) | r->data[7]);
        if (m17o_crc(st->m_lsf[1], 30) == 0) {
            r->flags |= M17O_F_DELIVERED;
            if (t_sink.net && r >= t_sink.recs && r < t_sink.recs + t_sink.cap)
                m17o_format_net_frame((uint16_t)(t_sink.sid_base + st->frame_id_epoch), st->m_lsf[1], r->fn, &r->data[8],
                                      t_sink.dst_override, t_sink.net + (size_t)(r - t_sink.recs) * 56);
        }
        break; }
    case 3: {                                       /* decode_packet_frame :161-177 */
        m17o_demap_frame(s, sb);
        m17o_de_correlate_f(sb, sb, 368);
        m17o_de_interleave(sb, so0, 368);
        m17o_de_punc(3, so0, so1, 420);
        m17o_viterbi_decode(so1, bits, 420);
        m17o_pack_1_to_8(&bits[1], r->data, 208);
        uint8_t eof = r->data[25] >> 7;
        uint8_t fn = (r->data[25] >> 2) & 0x1F;
        r->fn = (uint16_t)((eof << 8) | fn);
        if (parse_packet(st, r->data, eof, fn)) r->flags |= M17O_F_PKT_VALID;
        break; }
    case 4:                                         /* decode_bert_frame: empty */
    default:
        break;
    }
}

static void reset_sync(m17o_chan *st) { for (int i = 0; i < 8; i++) st->m_sync[i] = 0; }

/* m17_dbase.cpp:60-75 */
static void do_aos(m17o_chan *st) { st->g_errors = 0; st->n_frames = 0; st->in_frame = 1; st->frame_id_epoch++; }
static void do_los(m17o_chan *st) { st->in_frame = 0; st->frame_id_epoch++; }

static m17o_rec *new_rec(m17o_rec *recs, int cap, int *n, m17o_rec *scratch)
{
    m17o_rec *r = (*n < cap) ? &recs[*n] : scratch;
    memset(r, 0, sizeof *r);
    (*n)++;
    return r;
}

/* m17_rx_frame.cpp:126-177 */
static int rx_symbols(m17o_chan *st, const float *sym, int len, m17o_rec *recs, int cap, int parse)
{
    int n = 0;
    m17o_rec scratch;
    for (int i = 0; i < len; i++) {
        uint8_t type, votes; float var;
        if (st->m_flock) {
            st->m_f_sym[st->m_fclk] = sym[i];
            st->m_fclk = (st->m_fclk + 1) % M17O_FRAME_SYMS;
            if (st->m_fclk == 0) {
                m17o_sync_check(st->m_f_sym, &type, &votes, &var);
                m17o_rec *r = new_rec(recs, cap, &n, &scratch);
                r->type = type; r->votes = votes; r->variance = var;
                r->block = st->block_count; r->sym_pos = (uint16_t)i;
                if (type == 5) {
                    st->m_flock = 0; reset_sync(st); do_los(st);
                    r->flags |= M17O_F_EOT;
                } else if (sync_ok(type, votes, var, 1)) {
                    r->flags |= M17O_F_SYNC_OK;
                    if (parse) m17o_rx_parse(st, st->m_f_sym, type, r);
                    st->m_frame_errors = 0;
                } else {
                    st->m_frame_errors++;
                    if (st->m_frame_errors > M17O_LIT_N_FERROR) {
                        st->m_flock = 0; reset_sync(st); do_los(st);
                        r->flags |= M17O_F_LOST;
                    } else if (parse) {
                        m17o_rx_parse(st, st->m_f_sym, type, r);
                    }
                }
                r->frame_errors = (uint8_t)st->m_frame_errors;
            }
        } else {
            for (int k = 0; k < 7; k++) st->m_sync[k] = st->m_sync[k + 1];
            st->m_sync[7] = sym[i];
            m17o_sync_check(st->m_sync, &type, &votes, &var);
            if (sync_ok(type, votes, var, 0)) {
                for (int k = 0; k < 8; k++) st->m_f_sym[k] = st->m_sync[k];
                st->m_fclk = M17O_LIT_FCLK_AFTER_SYNC;
                st->m_frame_errors = 0;
                st->m_flock = 1;
                do_aos(st);
                m17o_rec *r = new_rec(recs, cap, &n, &scratch);
                r->type = type; r->votes = votes; r->variance = var;
                r->block = st->block_count; r->sym_pos = (uint16_t)i;
                r->flags = M17O_F_AOS;
            }
        }
    }
    return n;
}

int m17o_rx_symbols(m17o_chan *st, const float *sym, int n, m17o_rec *recs, int cap)
{
    return rx_symbols(st, sym, n, recs, cap, 1);
}

/* m17_dsp.cpp:461-476 (AFC branch :468 not taken: radio.cpp:8 default off) */
static int dsp_rx(m17o_chan *st, const int16_t *iq, m17o_rec *recs, int cap,
                  float *syms, int *nsym, int parse)
{
    float tempd[M17O_DISC_OUT];
    float tempc[M17O_BLOCK_SAMPLES / 2];
    m17o_frontend(st, iq, tempd, NULL, NULL);
    int n = m17o_rx_sync_samples(st, tempd, tempc, M17O_DISC_OUT);
    if (syms) memcpy(syms, tempc, sizeof(float) * (size_t)(n > 0 ? n : 0));
    if (nsym) *nsym = n;
    int k = rx_symbols(st, tempc, n, recs, cap, parse);
    st->block_count++;
    return k;
}

int m17o_dsp_rx(m17o_chan *st, const int16_t *iq, m17o_rec *recs, int cap, float *syms, int *nsym)
{
    return dsp_rx(st, iq, recs, cap, syms, nsym, 1);
}

/* ------------------------------------------------------------------ */
/* Pluto-style wide-band ingest: 384 kHz -> 48 kHz        */
/* ------------------------------------------------------------------ */
/* m17_dsp_build_lpf_filter (m17_dsp.cpp:347-360) */
static void build_lpf(float *filter, float bw, int ntaps)
{
    double a, B = bw, t = -(ntaps - 1) / 2;
    for (int i = 0; i < ntaps; i++) {
        if (t == 0) a = 2.0 * B;
        else a = 2.0 * B * sin(M_PI * t * B) / (M_PI * t * B);
        filter[i] = (float)a;
        t = t + 1.0;
    }
}

/* build_pluto_rx_dec_filter (radio.cpp:45-51): rectangular-window LPF 0.125, DC gain 0.9,
 * m17_dsp_float_to_short (m17_dsp.cpp:380-384): (int16_t)(f * 0x7FFF), truncating */
void m17o_pluto_build_dec_filter(int16_t *coffs)
{
    float f[31];
    build_lpf(f, 0.125f, 31);
    m17o_set_filter_gain(f, 0.9f, 1, 31);
    for (int i = 0; i < 31; i++) coffs[i] = (int16_t)(f[i] * 0x7FFF);
}

/* rx_decimate_filter + sub_filter (radio.cpp:18-40) run as radio_receive_samples does
 * (:157-177): 31-sample history in front of each chunk, output i = taps over buffer[8i..8i+30],
 * symmetric form, int32 accumulate, arithmetic >> 15, truncation to int16.
 * hist: 31 complex int16 (zero-initialised static in the reference); n_in multiple of 8, >= 31. */
void m17o_pluto_decimate(int16_t *hist, const int16_t *in, int n_in, int16_t *out)
{
    int16_t c[31];
    m17o_pluto_build_dec_filter(c);
    for (int i = 0; i < n_in / 8; i++) {
        int32_t acc[2];
        for (int q = 0; q < 2; q++) {
            /* window sample j is stream sample 8i - 31 + j (negative: history) */
#define XS(j) (((8 * i - 31 + (j)) < 0) ? hist[2 * (31 + 8 * i - 31 + (j)) + q] : in[2 * (8 * i - 31 + (j)) + q])
            int32_t real = XS(15) * c[15];
            for (int k = 0; k < 15; k++) real += c[k] * (XS(k) + XS(30 - k));
#undef XS
            acc[q] = real >> 15;
        }
        out[2 * i] = (int16_t)acc[0];
        out[2 * i + 1] = (int16_t)acc[1];
    }
    for (int k = 0; k < 31; k++) {
        hist[2 * k] = in[2 * (n_in - 31 + k)];
        hist[2 * k + 1] = in[2 * (n_in - 31 + k) + 1];
    }
}

/* ------------------------------------------------------------------ */
/* Transmit side: frame builders + 4-FSK RRC modulator.   */
/* The checker of the product's signal source (m17_txgen.cpp host,      */
/* m17_gen.hip device).  Not on the receive hot path.                   */
/* ------------------------------------------------------------------ */
/* pack_16_to_2 (m17_bit_utils.cpp:74-85) */
static int tx_pack_16_to_2(uint16_t in, uint8_t *out)
{
    for (int i = 0; i < 8; i++) out[i] = (uint8_t)((in >> (14 - 2 * i)) & 0x03);
    return 8;
}
/* pack_1_to_2 (m17_bit_utils.cpp:19-25) */
static int tx_pack_1_to_2(const uint8_t *in, uint8_t *out, int len)
{
    int idx = 0;
    for (int i = 0; i < len; i += 2) out[idx++] = (uint8_t)((in[i] << 1) | in[i + 1]);
    return idx;
}

/* build_lich (m17_tx_routines.cpp:38-54): pack_48_to_8 dest, src (m17_bit_utils.cpp:33-41), the packed type word
 * (m17_pack_type, :230-244, passed in packed), 14 meta bytes, CRC-16 big-endian.  Returns 30. */
int m17o_build_lsf(uint64_t dst, uint64_t src, uint16_t type_word, const uint8_t *meta, uint8_t *lsf)
{
    int idx = 0;
    for (int i = 0; i < 6; i++) lsf[idx++] = (uint8_t)((dst >> (40 - 8 * i)) & 0xFF);
    for (int i = 0; i < 6; i++) lsf[idx++] = (uint8_t)((src >> (40 - 8 * i)) & 0xFF);
    lsf[idx++] = (uint8_t)((type_word >> 8) & 0xFF);
    lsf[idx++] = (uint8_t)(type_word & 0xFF);
    for (int i = 0; i < 14; i++) lsf[idx++] = meta[i];
    uint16_t crc = m17o_crc(lsf, idx);
    lsf[idx++] = (uint8_t)(crc >> 8);
    lsf[idx++] = (uint8_t)(crc & 0xFF);
    return idx;
}

/* m17_fmt_add_tx_preamble (m17_tx_routines.cpp:24-31) */
int m17o_preamble_dibits(uint8_t *dibits)
{
    int idx = 0;
    for (int i = 0; i < 192 / 2; i++) { dibits[idx++] = 0x01; dibits[idx++] = 0x03; }
    return idx;
}

/* m17_fmt_add_eot (m17_tx_routines.cpp:242-255) */
int m17o_eot_dibits(uint8_t *dibits)
{
    int idx = 0;
    for (int i = 0; i < 24; i++) {
        for (int k = 0; k < 6; k++) dibits[idx++] = 0x01;
        dibits[idx++] = 0x03;
        dibits[idx++] = 0x01;
    }
    return idx;
}

/* m17_fmt_add_link_setup_frame (m17_tx_routines.cpp:92-117) behind build_lich: conv_encode_8 of the 30 LSF bytes
 * (488 bits), P1 puncture (368), interleave, de-correlate, sync word 0x55F7 + 184 dibits.
 * reference_quirks = 0 (default of every caller in tests/): the frame the code sets out to build, each stage in a
 *   buffer of its own -- what the M17 specification describes.
 * reference_quirks = 1: the reference's `uint8_t tx_bit[2][388]` (:93) as it lies in memory: 488 coded bits run
 *   100 bytes into tx_bit[1], which m17_punc_p1 (m17_puncture.cpp:12-21) is overwriting with its own output while it
 *   still reads tx_bit[0][388..487] from there -- the last 100 coded bits it punctures are its own early output.
 *   The reference's receiver never notices: decode_link_frame's CRC gate drops the LSF content. */
int m17o_lsf_frame_dibits(const uint8_t *lsf, uint8_t *dibits, int reference_quirks)
{
    uint8_t flat[2 * 388 + 128], wide0[512], wide1[512];
    uint8_t *tx0 = reference_quirks ? flat : wide0;
    uint8_t *tx1 = reference_quirks ? flat + 388 : wide1;
    int len = m17o_conv_encode_8(lsf, tx0, 30);
    len = m17o_punc(1, tx0, tx1, len);
    m17o_interleave_u8(tx1, tx0, len);
    m17o_de_correlate_u8(tx0, tx1, len);
    int idx = tx_pack_16_to_2(0x55F7, dibits);
    idx += tx_pack_1_to_2(tx1, &dibits[idx], len);
    return idx;
}

/* m17_fmt_add_stream_frame (m17_tx_routines.cpp:143-187): five LICH bytes + counter byte as four Golay words (96
 * bits), FN + 16 payload bytes conv-encoded (296) and P2-punctured in place (272), interleave, de-correlate, sync word
 * 0xFF5D.  lich_count and fn are the caller's m_lich_count / m_fn (the reference advances them itself, :155,:170).
 * (The reference's txb[0] is overrun here too -- 96 + 296 = 392 > 388 -- but the in-place puncture reads every bit
 * before anything is written over it, so its output is the specified one: no quirk to restate.) */
int m17o_stream_frame_dibits(const uint8_t *lsf, int lich_count, uint16_t fn, const uint8_t *payload, uint8_t *dibits)
{
    uint8_t tmp[80], txb0[512], txb1[512];
    int idx = 0;
    for (int i = 0; i < 5; i++) tmp[idx++] = lsf[lich_count * 5 + i];
    tmp[idx++] = (uint8_t)((lich_count & 0x07) << 5);
    /* pack_8_to_12_x4 (m17_bit_utils.cpp:132-149) */
    uint16_t dw[4];
    dw[0] = (uint16_t)((tmp[0] << 4) | ((tmp[1] >> 4) & 0x0F));
    dw[1] = (uint16_t)(((tmp[1] & 0x0F) << 8) | tmp[2]);
    dw[2] = (uint16_t)((tmp[3] << 4) | ((tmp[4] >> 4) & 0x0F));
    dw[3] = (uint16_t)(((tmp[4] & 0x0F) << 8) | tmp[5]);
    int len = 0;
    for (int i = 0; i < 4; i++) {
        uint32_t dpw = m17o_golay_encode(dw[i]);
        for (uint32_t m = 0x800000; m; m >>= 1) txb0[len++] = (dpw & m) ? 1 : 0;      /* pack_24_to_1 :63-69 */
    }
    const int fn_start = len;
    tmp[0] = (uint8_t)(fn >> 8); tmp[1] = (uint8_t)(fn & 0xFF);
    for (int i = 0; i < 16; i++) tmp[2 + i] = payload[i];
    len = m17o_conv_encode_8(tmp, &txb0[fn_start], 18);
    len = m17o_punc(2, &txb0[fn_start], &txb0[fn_start], len);
    m17o_interleave_u8(txb0, txb1, len + fn_start);
    m17o_de_correlate_u8(txb1, txb0, len + fn_start);
    idx = tx_pack_16_to_2(0xFF5D, dibits);
    idx += tx_pack_1_to_2(txb0, &dibits[idx], len + fn_start);
    return idx;
}

/* m17_fmt_add_packet (m17_tx_routines.cpp:201-222): up to 25 payload bytes, byte 25 = EOF flag | nf << 2 (nf is an
 * uint8_t in the reference, not masked), conv_encode_8 of 26 bytes (424 bits), P3 puncture of the first 420 (368),
 * interleave, de-correlate, sync word 0x75FF.  reference_quirks as for the link-setup frame: `txb[2][388]` (:203)
 * takes 424 coded bits, and m17_punc_p3 reads txb[0][388..419] out of the row it is writing. */
int m17o_packet_frame_dibits(const uint8_t *payload, int len, int eof, int nf, uint8_t *dibits, int reference_quirks)
{
    uint8_t tmp[80], flat[2 * 388 + 128], wide0[512], wide1[512];
    if (len > 25 || len < 0) return 0;
    uint8_t *t0 = reference_quirks ? flat : wide0;
    uint8_t *t1 = reference_quirks ? flat + 388 : wide1;
    memset(tmp, 0, 25);
    memcpy(tmp, payload, (size_t)len);
    tmp[25] = eof ? 0x80 : 0x00;
    tmp[25] |= (uint8_t)(nf << 2);
    m17o_conv_encode_8(tmp, t0, 26);
    m17o_punc(3, t0, t1, 420);
    m17o_interleave_u8(t1, t0, 368);
    m17o_de_correlate_u8(t0, t1, 368);
    tx_pack_16_to_2(0x75FF, dibits);
    tx_pack_1_to_2(t1, &dibits[8], 368);
    return M17O_FRAME_SYMS;
}

/* m17_mod_init (m17_modulate.cpp:65-77) at radio_get_oversample() == 10 (radio.cpp:207-215, LimeSDR) */
void m17o_mod_init(m17o_mod *m)
{
    memset(m, 0, sizeof *m);
    m17o_build_rrc_filter(m->c, 0.5f, M17O_TX_FN * M17O_TX_OS, M17O_TX_OS);
    m17o_set_filter_gain(m->c, 10, 1, M17O_TX_FN * M17O_TX_OS);
    /* m_tx_lu (m17_modulate.cpp:9): float initialisers from double expressions */
    m->lu[0] = (float)(M_PI / 30.0); m->lu[1] = (float)(M_PI / 10.0);
    m->lu[2] = (float)(-M_PI / 30); m->lu[3] = (float)(-M_PI / 10.0);
}

/* one symbol: mod_filter (m17_modulate.cpp:49-61) + sub_filter (:42-48) + mod_fsk (:22-38).  cos / sin of the float
 * m_acc under <math.h> in C++ are the float overloads, and float * int (0x3FFF) stays float: cosf / sinf here.
 * sums / phases (optional, 10 floats each): the filter outputs and m_acc after each sample, before the wrap. */
static void mod_symbol(m17o_mod *m, float sample, int16_t *iq, float *sums, float *phases)
{
    const int os = M17O_TX_OS;
    float sum[M17O_TX_OS];
    for (int i = 0; i < M17O_TX_FN - 1; i++) m->s[i] = m->s[i + 1];
    m->s[M17O_TX_FN - 1] = sample;
    for (int i = 0, n = os - 1; i < os; i++, n--) {
        const float *c = &m->c[n];
        float a = m->s[0] * c[0];
        for (int j = 1; j < M17O_TX_FN; j++) a += m->s[j] * c[j * os];
        sum[i] = a;
    }
    for (int i = 0; i < os; i++) {
        m->acc += sum[i];
        iq[2 * i]     = (int16_t)(cosf(m->acc) * 0x3FFF);
        iq[2 * i + 1] = (int16_t)(sinf(m->acc) * 0x3FFF);
        if (sums) sums[i] = sum[i];
        if (phases) phases[i] = m->acc;
    }
    /* phase accumulator wrap (:33-37): float / double -> float, modf on the promoted value, float * double -> float */
    m->acc = (float)(m->acc / (2.0 * M_PI));
    double ip;
    m->acc = (float)modf((double)m->acc, &ip);
    m->acc = (float)(m->acc * 2.0 * M_PI);
}

/* m17_mod_dibits (m17_modulate.cpp:80-84) / m17_mod_carrier (:88-92: a dibit value of 255 here): n symbols in,
 * 10 n IQ samples out */
int m17o_modulate(m17o_mod *m, const uint8_t *dibits, int n, int16_t *iq, float *sums, float *phases)
{
    for (int i = 0; i < n; i++)
        mod_symbol(m, dibits[i] == 255 ? 0.0f : m->lu[dibits[i] & 3], iq + (size_t)i * 2 * M17O_TX_OS,
                   sums ? sums + (size_t)i * M17O_TX_OS : NULL, phases ? phases + (size_t)i * M17O_TX_OS : NULL);
    return n * M17O_TX_OS;
}
int m17o_sizeof_mod(void) { return (int)sizeof(m17o_mod); }

int m17o_rx_blocks_net(m17o_chan *st, int C, int nblk, const int16_t *iq,
                       m17o_rec *recs, int cap, int32_t *counts,
                       float *syms, int32_t *nsyms, int mode, int nthreads,
                       uint8_t *net, const uint16_t *stream_ids, uint64_t dst_override)
{
    const size_t blk = (size_t)M17O_BLOCK_SAMPLES * 2;
    const size_t symstride = (size_t)nblk * 193 + 8;
    if (nthreads < 1) nthreads = 1;
#pragma omp parallel for schedule(static) num_threads(nthreads)
    for (int c = 0; c < C; c++) {
        int n = 0, ns = 0;
        t_sink.net = (net && recs) ? net + (size_t)c * cap * 56 : NULL;
        t_sink.recs = recs ? recs + (size_t)c * cap : NULL;
        t_sink.cap = cap;
        t_sink.sid_base = stream_ids ? stream_ids[c] : 0;
        t_sink.dst_override = dst_override;
        for (int b = 0; b < nblk; b++) {
            int k = 0;
            int at = n < cap ? n : cap;
            n += dsp_rx(&st[c], iq + ((size_t)c * nblk + b) * blk,
                        recs ? recs + (size_t)c * cap + at : NULL, recs ? cap - at : 0,
                        syms ? syms + (size_t)c * symstride + ns : NULL, &k, mode);
            if (k < 0) k = 0;
            if (nsyms) nsyms[(size_t)c * nblk + b] = k;
            ns += k;
        }
        t_sink.net = NULL;
        if (counts) counts[c] = n;
    }
    return 0;
}

int m17o_rx_blocks(m17o_chan *st, int C, int nblk, const int16_t *iq,
                   m17o_rec *recs, int cap, int32_t *counts,
                   float *syms, int32_t *nsyms, int mode, int nthreads)
{
    return m17o_rx_blocks_net(st, C, nblk, iq, recs, cap, counts, syms, nsyms, mode, nthreads, NULL, NULL, 0);
}
