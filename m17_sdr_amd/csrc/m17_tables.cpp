// m17_tables.cpp -- host construction of the constant tables of the M17
// receive chain (what the reference builds in main.cpp:110-118).  Compiled with
// -ffp-contract=off: the tap design must round exactly like the reference's
// double-precision libm code.
#include "m17_host.h"
#include "m17_dev.h"
#include "../../include/m17gpu.h"
#include <cmath>
#include <cstring>
#include <cstdlib>
#include <mutex>
#include <vector>
#include <string>

namespace m17 {

// M17 randomising sequence, 46 bytes (m17_correlate.cpp:3-7)
static const uint8_t kRandSeq[46] = {
    0xD6,0xB5,0xE2,0x30,0x82,0xFF,0x84,0x62,0xBA,0x4E,0x96,0x90,0xD8,0x98,0xDD,0x5D,
    0x0C,0xC8,0x52,0x43,0x91,0x1D,0xF8,0x6E,0x68,0x2F,0x35,0xDA,0x14,0xEA,0xCD,0x76,
    0x19,0x8D,0xD5,0x80,0xD1,0x33,0x87,0x13,0x57,0x18,0x2D,0x29,0x78,0xC3 };

// Golay(24,12) generator rows (m17_golay.cpp:11)
static const uint16_t kGolayRows[12] = {
    0xC75,0x63B,0xF68,0x7B4,0x3DA,0xD99,0x6CD,0x367,0xDC6,0xA97,0x93E,0x8EB };

// Root-raised-cosine design in double, cast to float (m17_dsp.cpp:295-315).
// Keeps the reference's quirks: roll-off + 1e-4, integer-division start time.
void build_rrc(float *f, float rolloff, int ntaps, int sps)
{
    const double beta = rolloff + 0.0001;
    const double ts = sps;
    double t = -(ntaps - 1) / 2;
    for (int i = 0; i < ntaps; ++i, t = t + 1.0) {
        const double a = 2.0 * beta / (M_PI * std::sqrt(ts));
        const double b = std::cos((1.0 + beta) * M_PI * t / ts);
        double c;
        if (t == 0)
            c = (1.0 - beta) * M_PI / (4 * beta);
        else
            c = std::sin((1.0 - beta) * M_PI * t / ts) / (4.0 * beta * t / ts);
        const double d = (1.0 - (4.0 * beta * t / ts) * (4.0 * beta * t / ts));
        f[i] = (float)(a * (b + c) / d);
    }
}

// m17_dsp.cpp:420-429 : sequential float sum, one float divide, scale
void set_filter_gain(float *f, float gain, int stride, int ntaps)
{
    float acc = 0;
    for (int i = 0; i < ntaps; ++i) acc += f[i * stride];
    const float g = gain / acc;
    for (int i = 0; i < ntaps; ++i) f[i * stride] = f[i * stride] * g;
}

uint16_t crc16(const uint8_t *p, int n)
{
    const Tables &T = tables();
    uint16_t crc = 0xFFFF;
    for (int i = 0; i < n; ++i)
        crc = (uint16_t)((crc << 8) ^ T.crc[(uint8_t)((crc >> 8) ^ p[i])]);
    return crc;
}

int puncture_keep(int type, int k)
{
    switch (type) {
    case 1: return (k % 61) % 4 != 2;        // P1: 1,1,0,1 repeating, 61 long (m17_puncture.cpp:4-6)
    case 2: return (k % 12) != 11;           // P2 (m17_puncture.cpp:8)
    default: return (k % 8) != 7;            // P3 (m17_puncture.cpp:10)
    }
}

static void build(Tables &T)
{
    std::memset(&T, 0, sizeof T);
    // CRC-16/M17: poly 0x5935, MSB first
    for (int i = 0; i < 256; ++i) {
        uint16_t x = (uint16_t)(i << 8);
        for (int b = 0; b < 8; ++b)
            x = (uint16_t)((x & 0x8000) ? ((x << 1) ^ kCrcPoly) : (x << 1));
        T.crc[i] = x;
    }
    // de-randomiser bits, MSB of byte 0 first
    for (int i = 0; i < kSoftBits; ++i)
        T.derand[i] = (kRandSeq[i >> 3] >> (7 - (i & 7))) & 1;
    // quadratic permutation
    std::vector<int> inv(kSoftBits, -1);
    for (int i = 0; i < kSoftBits; ++i) {
        int d = ((i * 45) + (92 * i * i)) % kSoftBits;
        T.interleave[i] = (uint16_t)d;
        if (inv[d] != -1) std::abort();      // must be a bijection
        inv[d] = i;
    }
    // Golay tables; error table filled in ascending word order, last writer
    // wins, initial fill as the reference writes it (m17_golay.cpp:53-71)
    for (int d = 0; d < 4096; ++d) {
        uint16_t p = 0;
        for (int b = 0; b < 12; ++b)
            if (d & (0x800 >> b)) p ^= kGolayRows[b];
        T.golay_enc[d] = p;
    }
    for (int i = 0; i < M17_LIT_GOLAY_FILL_END; ++i) T.golay_err[i] = M17_LIT_GOLAY_UNRECOVERABLE;
    for (uint32_t w = 0; w < (1u << 24); ++w) {
        const int wt = __builtin_popcount(w);
        if (!(wt < M17_LIT_GOLAY_MAX_BITS)) continue;
        const uint16_t data = (uint16_t)(w >> 12);
        const uint16_t syn = (uint16_t)((w & 0xFFF) ^ T.golay_enc[data]);
        T.golay_err[syn] = (uint16_t)((wt << 12) | data);
    }
    // polyphase matched / derivative filters (m17_rx_sync.cpp:101-122)
    {
        constexpr int N = kPhases * kTaps;
        std::vector<float> mother(N), deriv(N);
        build_rrc(mother.data(), 0.5f, N, kPhases * 2);
        for (int i = 0; i < N; ++i)
            deriv[i] = mother[(i + 1) % N] - mother[(i + N - 1) % N];
        for (int ph = 0; ph < kPhases; ++ph)
            for (int j = 0; j < kTaps; ++j) {
                T.mf[ph][j] = mother[ph + j * kPhases];
                T.md[ph][j] = deriv[ph + j * kPhases];
            }
        for (int ph = 0; ph < kPhases; ++ph)
            set_filter_gain(T.mf[ph], 1.0f, 1, kTaps);
    }
    // fused de-randomise / de-interleave / de-puncture gather tables
    const int olen[4] = {0, 488, 296, 420};
    for (int type = 1; type <= 3; ++type) {
        T.glen[type] = (int16_t)olen[type];
        int j = (type == 2) ? 96 : 0;        // stream: conv part starts at so[96]
        for (int k = 0; k < olen[type]; ++k) {
            if (puncture_keep(type, k)) {
                const int src = inv[j++];
                T.gather[type][k] = (int16_t)src;
                T.gsign[type][k] = T.derand[src] ? -1 : 1;
            } else {
                T.gather[type][k] = -1;
                T.gsign[type][k] = 1;
            }
        }
        if (j != kSoftBits) std::abort();
    }
    for (int j = 0; j < 96; ++j) {
        T.lich_src[j] = (int16_t)inv[j];
        T.lich_sign[j] = T.derand[inv[j]] ? -1 : 1;
    }
    // convolutional code K=5, G1=0x19 G2=0x17 (m17_conv.cpp:24-29) and the
    // branch-metric selectors of the add-compare-select table (:93-108)
    for (int i = 0; i < 32; ++i) {
        T.clut[i][0] = (uint8_t)(((i >> 4) ^ (i >> 1) ^ i) & 1);
        T.clut[i][1] = (uint8_t)(((i >> 4) ^ (i >> 3) ^ (i >> 2) ^ i) & 1);
    }
    for (int v = 0; v < 16; ++v) {
        T.bm_even[v] = (uint8_t)((T.clut[2 * v][0] << 1) | T.clut[2 * v][1]);
        T.bm_odd[v]  = (uint8_t)((T.clut[2 * v + 1][0] << 1) | T.clut[2 * v + 1][1]);
    }
    // the quad decoder's branch metrics (m17_decode_quad.hip, DQ_STEP) rest on this separation of bm_even
    for (int v = 0; v < 16; ++v) {
        const int j = v >> 2, i = v & 3;
        if (T.bm_even[v] != (((((j >> 1) ^ i) & 1) << 1) | (((j >> 1) ^ j ^ (i >> 1)) & 1))) std::abort();
    }
}

// Pluto receive decimator taps (radio.cpp:45-51): rectangular-window low-pass of
// bandwidth 0.125 (m17_dsp.cpp:347-360), DC gain 0.9, Q15 by truncation (:380-384)
void build_pluto_dec_filter(int16_t *coffs)
{
    float f[31];
    const double bw = 0.125f;
    double t = -(31 - 1) / 2;
    for (int i = 0; i < 31; ++i, t = t + 1.0)
        f[i] = (float)((t == 0) ? 2.0 * bw : 2.0 * bw * std::sin(M_PI * t * bw) / (M_PI * t * bw));
    set_filter_gain(f, 0.9f, 1, 31);
    for (int i = 0; i < 31; ++i) coffs[i] = (int16_t)(f[i] * 0x7FFF);
}

const Tables &tables()
{
    static Tables T;
    static std::once_flag once;
    std::call_once(once, [] { build(T); });
    return T;
}

} // namespace m17

// Host view of the literal constants the product is built from, by name -- so that a test can hold
// them against the values extracted from the reference's source text (tests/golden/ref_constants.json).
// Every row is produced from the SAME definition the kernels / tables / signal sources use.
extern "C" int m17gpu_get_constant(const char *name, void *out, int cap_bytes)
{
    if (!name || !out) return M17GPU_ERR_ARG;
    const m17::Tables &T = m17::tables();
    std::vector<uint8_t> buf;
    auto put = [&](const void *p, size_t n) { const uint8_t *b = (const uint8_t *)p; buf.insert(buf.end(), b, b + n); };
    const std::string nm(name);
    if (nm == "sframe") {                       // float [6][8], +1 / -1
        for (int k = 0; k < 6; ++k)
            for (int i = 0; i < 8; ++i) { const float v = (m17dev::sync_neg_mask(k) >> i & 1u) ? -1.0f : 1.0f; put(&v, 4); }
    } else if (nm == "derand_bits") {           // uint8 [368]
        put(T.derand, sizeof T.derand);
    } else if (nm == "golay_rows") {            // uint16 [12]: parity of the data word with only bit (11-k) set
        for (int k = 0; k < 12; ++k) { const uint16_t v = T.golay_enc[0x800 >> k]; put(&v, 2); }
    } else if (nm == "punc1" || nm == "punc2" || nm == "punc3") {     // uint8 [61] / [12] / [8]
        const int ty = nm[4] - '0', per = ty == 1 ? 61 : (ty == 2 ? 12 : 8);
        for (int k = 0; k < per; ++k) { const uint8_t v = (uint8_t)m17::puncture_keep(ty, k); put(&v, 1); }
    } else if (nm == "butterfly") {             // uint8 [16][5]: BF(v, w, x, y, z) rows of m17_conv.cpp:93-108
        for (int v = 0; v < 16; ++v) {
            const uint8_t row[5] = {(uint8_t)v, (uint8_t)((2 * v) & 15), T.bm_even[v], (uint8_t)((2 * v + 1) & 15), T.bm_odd[v]};
            put(row, 5);
        }
    } else if (nm == "crc_poly") {              // uint16: table entry of byte 0x01 of an MSB-first table is the polynomial
        put(&T.crc[1], 2);
    } else if (nm == "tx_lut") {                // float [4]
        float lut[4]; m17::tx_deviation_lut(lut); put(lut, sizeof lut);
    } else if (nm == "rx_literals") {           // double [24]: the streaming arithmetic's literals, in M17_RX_LITERALS' order (m17_dev.h)
        const double v[] = M17_RX_LITERALS;
        put(v, sizeof v);
    } else if (nm == "sync_words") {            // uint16 [4]: link setup, stream, packet, BERT
        const uint16_t w[4] = {m17::kSyncLinkSetup, m17::kSyncStream, m17::kSyncPacket, m17::kSyncBert};
        put(w, sizeof w);
    } else
        return M17GPU_ERR_ARG;
    if ((int)buf.size() > cap_bytes) return M17GPU_ERR_ARG;
    std::memcpy(out, buf.data(), buf.size());
    return (int)buf.size();
}
