// m17_compat.cpp -- the reference's C++ function signatures (include/
// m17defines_compat.h) on top of the C-ABI core, for ONE hidden channel, so that
// G4GUO/m17_sdr's m17_rx_frame.cpp and m17_rx_parse.cpp link against this library
// unchanged.  Compute stages (front end, timing recovery, demap,
// Viterbi, Golay) run on the GPU through libm17gpu.so with batch size 1; the
// byte/bit movers of m17_bit_utils.cpp, the permutations and the CRC are host
// code, as they are host code inside the reference's own dispatcher.
// The reference reports no errors on this path; a HIP failure aborts.
#include "../../include/m17defines_compat.h"
#include "../../include/m17gpu.h"
#include "m17_host.h"
#include <hip/hip_runtime_api.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>

// supplied by the unchanged reference m17_rx_frame.cpp when it is linked in
extern "C++" __attribute__((weak)) void m17_rx_symbols(float *sym, int len);
extern "C++" __attribute__((weak)) bool m17_rx_lock(void);

namespace {

struct Chan0 {
    m17gpu_ctx *ctx = nullptr;
    void *d_a = nullptr, *d_b = nullptr;      // scratch device buffers
    Chan0() {
        if (m17gpu_create(&ctx, 1, 1, 0) != M17GPU_OK) die("m17gpu_create");
        if (hipMalloc(&d_a, 16384) != hipSuccess || hipMalloc(&d_b, 16384) != hipSuccess) die("hipMalloc");
    }
    [[noreturn]] static void die(const char *what) {
        std::fprintf(stderr, "m17 compat shim: %s failed: %s\n", what, m17gpu_last_error());
        std::abort();
    }
    void up(void *d, const void *h, size_t n) { if (hipMemcpy(d, h, n, hipMemcpyHostToDevice) != hipSuccess) die("H2D"); }
    void down(void *h, const void *d, size_t n) { if (hipMemcpy(h, d, n, hipMemcpyDeviceToHost) != hipSuccess) die("D2H"); }
    void ok(int rc, const char *what) { if (rc != M17GPU_OK) die(what); }
};

Chan0 &ch() { static Chan0 c; return c; }

int depuncture(int type, const float *in, float *out, int len)
{
    const m17::Tables &T = m17::tables();
    (void)T;
    int idx = 0;
    for (int k = 0; k < len; ++k) {
        const bool keep = (type == 1) ? ((k % 61) % 4 != 2) : (type == 2) ? (k % 12 != 11) : (k % 8 != 7);
        out[k] = keep ? in[idx++] : 0.0f;                 // m17_puncture.cpp:47-79
    }
    return len;
}

} // namespace

// ---- init (main.cpp:110-118): tables are built when the context is created
void m17_crc_init(void) { (void)m17::tables(); }
void m17_init_conv(void) { (void)m17::tables(); }
void m17_init_de_correlate(void) { (void)m17::tables(); }
void m17_dsp_init(void) {}
void m17_golay_init(void) { (void)m17::tables(); }
void m17_rx_sync_init(void) { ch().ok(m17gpu_reset(ch().ctx, nullptr), "m17gpu_reset"); }

// ---- GPU stages
void m17_dsp_demap_frame(float *in, float *out)            // m17_dsp.cpp:82-95
{
    Chan0 &c = ch();
    c.up(c.d_a, in, sizeof(float) * 192);
    c.ok(m17gpu_demap_frame(c.ctx, (const float *)c.d_a, (float *)c.d_b, 1, nullptr), "m17gpu_demap_frame");
    c.down(out, c.d_b, sizeof(float) * 368);
}

int m17_viterbi_decode(float *in, uint8_t *out, int len)   // m17_conv.cpp:148-168
{
    Chan0 &c = ch();
    c.up(c.d_a, in, sizeof(float) * (size_t)len);
    c.ok(m17gpu_viterbi_decode(c.ctx, (const float *)c.d_a, (uint8_t *)c.d_b, len, 1, nullptr), "m17gpu_viterbi_decode");
    c.down(out, c.d_b, (size_t)len / 2);
    return len / 2;
}

int m_17_golay_decode(uint24_t word, uint12_t &odata)      // m17_golay.cpp:103-116
{
    Chan0 &c = ch();
    uint16_t r;
    c.up(c.d_a, &word, 4);
    c.ok(m17gpu_golay_decode(c.ctx, (const uint32_t *)c.d_a, (uint16_t *)c.d_b, 1, nullptr), "m17gpu_golay_decode");
    c.down(&r, c.d_b, 2);
    odata = r & 0xFFF;
    return r >> 12;
}

int m17_rx_sync_samples(float *in, float *out, int len)    // m17_rx_sync.cpp:77-99
{
    Chan0 &c = ch();
    if (len != M17GPU_DISC_OUT) {       // the batched core works in whole 40 ms blocks; no abort: the reference has no error path
        std::fprintf(stderr, "m17 compat shim: m17_rx_sync_samples handles len == 384 only (got %d); call ignored\n", len);
        return 0;
    }
    const int lock = (&m17_rx_lock != nullptr) ? (int)m17_rx_lock() : 0;
    c.up(c.d_a, in, sizeof(float) * M17GPU_DISC_OUT);
    int32_t *d_n = (int32_t *)((char *)c.d_b + 8192);
    c.ok(m17gpu_sync_samples(c.ctx, (const float *)c.d_a, 1, lock, (float *)c.d_b, d_n, nullptr), "m17gpu_sync_samples");
    int32_t n = 0;
    c.down(&n, d_n, 4);
    c.down(out, c.d_b, sizeof(float) * (size_t)n);
    return n;
}

void m17_dsp_rx(scmplx *in, int len)                       // m17_dsp.cpp:461-476
{
    Chan0 &c = ch();
    if (len != M17GPU_BLOCK_SAMPLES) {  // every caller in the reference passes N_SAMPLES (m17_tx_rx.cpp:37,147,165)
        std::fprintf(stderr, "m17 compat shim: m17_dsp_rx handles len == 1920 only (got %d); block ignored\n", len);
        return;
    }
    float tempd[M17GPU_DISC_OUT], tempc[M17GPU_BLOCK_SAMPLES / 2];
    c.up(c.d_a, in, sizeof(scmplx) * M17GPU_BLOCK_SAMPLES);
    c.ok(m17gpu_frontend(c.ctx, (const int16_t *)c.d_a, 1, (float *)c.d_b, nullptr, nullptr), "m17gpu_frontend");
    c.down(tempd, c.d_b, sizeof tempd);
    const int n = m17_rx_sync_samples(tempd, tempc, M17GPU_DISC_OUT);
    if (&m17_rx_symbols != nullptr) m17_rx_symbols(tempc, n);      // the reference's framer + parser
}

// ---- host glue (bit movers, permutations, CRC)
void m17_de_correlate_1(float *in, float *out, int len)    // m17_correlate.cpp:27-31
{
    const m17::Tables &T = m17::tables();
    for (int i = 0; i < len; ++i) out[i] = T.derand[i % 368] ? -in[i] : in[i];
}
void m17_de_interleave(float *in, float *out, int len)     // m17_interleave.cpp:8-12
{
    const m17::Tables &T = m17::tables();
    for (int i = 0; i < len; ++i) out[T.interleave[i % 368]] = in[i];
}
int m17_de_punc_p1(float *in, float *out, int len) { return depuncture(1, in, out, len); }
int m17_de_punc_p2(float *in, float *out, int len) { return depuncture(2, in, out, len); }
int m17_de_punc_p3(float *in, float *out, int len) { return depuncture(3, in, out, len); }

uint24_t hard_decode_24_bits(float *in)                    // m17_bit_utils.cpp:180-187
{
    uint24_t w = 0;
    for (int i = 0; i < 24; ++i) w = (w << 1) | (in[i] >= 0 ? 1u : 0u);
    return w;
}
int pack_1_to_8(uint8_t *in, uint8_t *out, int len)        // m17_bit_utils.cpp:26-32
{
    int n = 0;
    for (int i = 0; i < len; i += 8) {
        unsigned v = 0;
        for (int k = 0; k < 8; ++k) v = (v << 1) | in[i + k];
        out[n++] = (uint8_t)v;
    }
    return n;
}
int pack_12_to_8_x4x6(uint12_t *in, uint8_t *out)          // m17_bit_utils.cpp:152-172
{
    for (int h = 0; h < 2; ++h) {
        const uint32_t w = ((uint32_t)in[2 * h] << 12) | in[2 * h + 1];
        out[3 * h] = (uint8_t)(w >> 16); out[3 * h + 1] = (uint8_t)(w >> 8); out[3 * h + 2] = (uint8_t)w;
    }
    return 6;
}
uint48_t pack_8_to_48(uint8_t *in)                         // m17_bit_utils.cpp:100-114
{
    uint48_t v = 0;
    for (int i = 0; i < 6; ++i) v = (v << 8) | in[i];
    return v;
}
uint16_t pack_8_to_16(uint8_t *in) { return (uint16_t)((in[0] << 8) | in[1]); }   // :125-131
uint16_t m17_crc_array_encode(uint8_t *in, int len) { return m17::crc16(in, len); } // m17_crc.cpp:26-35
M17Type m17_upack_type(uint16_t word)                      // m17_bit_utils.cpp:245-254
{
    M17Type t;
    t.reserved = (word >> 11) & 0x1F; t.can = (word >> 7) & 0xF; t.est = (word >> 5) & 0x3;
    t.et = (word >> 3) & 0x3; t.dt = (word >> 1) & 0x3; t.p_s = word & 0x1;
    return t;
}
