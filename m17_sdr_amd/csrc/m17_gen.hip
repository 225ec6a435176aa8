// m17_gen.hip -- GPU-side synthetic M17 signal source: the host generator of
// m17_txgen.cpp (itself a restatement of m17_tx_routines.cpp:24-255 and m17_modulate.cpp:22-86)
// as four kernels, so that 16,384-channel sweeps and multi-GPU runs need no host-made IQ.
// Stream mode only (packet bursts stay on the host generator).
//
//   k_gen_symbols  thread per (channel, 192-symbol slot): carrier / preamble / LSF / stream
//                  frame n / EOT as dibit codes.  SplitMix64 is counter based, so the payload of
//                  frame n is draw 2n -- no sequential RNG walk.
//   k_gen_sum      thread per (channel, sample): the 31-tap polyphase RRC of the modulator,
//                  fp32, ascending order, bare first product (mod_filter, m17_modulate.cpp:49-61)
//   k_gen_phase    lane per channel: the running phase `acc += sum` with its per-symbol wrap
//                  (mod_fsk :22-38) is a strict fp32 chain -- the only sequential part
//   k_gen_iq       thread per (channel, sample): cos/sin in fp64, x 0x3FFF, truncation; AWGN by
//                  Box-Muller from counter-based draws, optional 63-tap band limit, rint, clamp
//
// Parity with the host generator: dibits, FIR sums and phases are bit-identical; cos/sin/log
// come from the device math library instead of glibc, so an IQ sample can differ by one LSB
// when a product lands within ~1e-12 of an integer (tests allow |diff| <= 1 on < 1e-6 of the
// samples and require identical decoded records).
#pragma clang fp contract(off)

namespace m17dev {

__device__ __forceinline__ uint64_t smix(uint64_t seed, uint64_t k)       // k-th draw, k >= 1
{
    uint64_t z = seed + k * 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
__device__ __forceinline__ double smix_uniform(uint64_t seed, uint64_t k)
{
    return (double)((smix(seed, k) >> 11)) * (1.0 / 9007199254740992.0) + 0.5 * (1.0 / 9007199254740992.0);
}

// M17 randomising sequence (m17_correlate.cpp:3-7)
__constant__ uint8_t c_rand_seq[46] = {
    0xD6,0xB5,0xE2,0x30,0x82,0xFF,0x84,0x62,0xBA,0x4E,0x96,0x90,0xD8,0x98,0xDD,0x5D,
    0x0C,0xC8,0x52,0x43,0x91,0x1D,0xF8,0x6E,0x68,0x2F,0x35,0xDA,0x14,0xEA,0xCD,0x76,
    0x19,0x8D,0xD5,0x80,0xD1,0x33,0x87,0x13,0x57,0x18,0x2D,0x29,0x78,0xC3 };

struct GenArgs {
    uint64_t base_seed;
    int first_channel, C, nblk, n_stream_frames, nslots;
    float lut[4];
};

__device__ __forceinline__ int gen_delay(uint64_t ch)
{
    return (int)(smix(0xD1B54A32D192ED03ull ^ (ch * 0x9E3779B97F4A7C15ull), 1) % 1920);
}

// conv encoder output pair for shift register sr (m17_conv.cpp:24-29)
__device__ __forceinline__ int clut0(unsigned i) { return ((i >> 4) ^ (i >> 1) ^ i) & 1; }
__device__ __forceinline__ int clut1(unsigned i) { return ((i >> 4) ^ (i >> 3) ^ (i >> 2) ^ i) & 1; }

// conv_encode (m17_conv.cpp:53-71) + puncture (m17_puncture.cpp:12-41) of nbytes into bits[at...]
__device__ int gen_encode_punctured(const uint8_t *in, int nbytes, int type, uint8_t *bits, int at)
{
    unsigned sr = 0; int k = 0;
    for (int i = 0; i <= nbytes; ++i) {
        const int nb = (i < nbytes) ? 8 : 4;                     // 4 flush steps
        for (int b = 0; b < nb; ++b) {
            if (i < nbytes && (in[i] & (0x80 >> b))) sr |= 0x10;
            const int o[2] = {clut0(sr), clut1(sr)};
            for (int h = 0; h < 2; ++h, ++k) {
                const bool keep = (type == 1) ? ((k % 61) % 4 != 2) : (k % 12 != 11);
                if (keep) bits[at++] = (uint8_t)o[h];
            }
            sr >>= 1;
        }
    }
    return at;
}

// interleave, randomise, dibits behind the sync word (finish_frame of m17_txgen.cpp)
__device__ void gen_finish(uint16_t sync, const uint8_t *bits, uint8_t *dst)
{
    uint8_t il[368];
    for (int i = 0; i < 368; ++i) il[((i * 45) + (92 * i * i)) % 368] = bits[i];
    for (int i = 0; i < 8; ++i) dst[i] = (uint8_t)((sync >> (14 - 2 * i)) & 3);
    for (int i = 0; i < 184; ++i) {
        const int a = 2 * i, b = 2 * i + 1;
        const int ra = (c_rand_seq[a >> 3] >> (7 - (a & 7))) & 1, rb = (c_rand_seq[b >> 3] >> (7 - (b & 7))) & 1;
        dst[8 + i] = (uint8_t)((((il[a] ^ ra) & 1) << 1) | ((il[b] ^ rb) & 1));
    }
}

__global__ __launch_bounds__(64)
void k_gen_symbols(GenArgs A, const uint16_t *__restrict__ genc, uint8_t *__restrict__ sym,
                   uint8_t *__restrict__ d_lsf, uint8_t *__restrict__ d_payload, int max_payload_frames,
                   int32_t *__restrict__ d_nframes)
{
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (long long)A.C * A.nslots) return;
    const int c = (int)(t / A.nslots), g = (int)(t - (long long)c * A.nslots);
    const uint64_t ch = (uint64_t)(A.first_channel + c), seed = A.base_seed + ch;
    uint8_t *dst = sym + ((size_t)c * A.nslots + g) * 192;
    const int N = A.n_stream_frames, period = 5 + N;
    const int slot = g % period, cyc = g / period;
    if (slot == 0) { for (int i = 0; i < 192; ++i) dst[i] = 255; return; }               // m17_mod_carrier
    if (slot <= 2) { for (int i = 0; i < 96; ++i) { dst[2 * i] = 1; dst[2 * i + 1] = 3; } return; }
    if (slot == 4 + N) {                                                                  // EOT
        const uint8_t pat[8] = {1, 1, 1, 1, 1, 1, 3, 1};
        for (int i = 0; i < 192; ++i) dst[i] = pat[i & 7];
        return;
    }
    // station identity (random_call + m17gen_build_lsf): broadcast destination, voice stream
    uint8_t lsf[30];
    {
        const int n = 4 + (int)(smix(seed, 1) % 3);
        uint64_t w = 0;
        for (int i = 8; i >= 0; --i) {                       // base-40, first character least significant
            w *= 40;
            if (i < n) {
                const int a = (int)(smix(seed, 2 + (uint64_t)i) % 36);     // "A..Z0..9"
                w += (uint64_t)(a < 26 ? a + 1 : a - 26 + 27);
            }
        }
        for (int i = 0; i < 6; ++i) lsf[i] = 0xFF;
        for (int i = 0; i < 6; ++i) lsf[6 + i] = (uint8_t)(w >> (40 - 8 * i));
        lsf[12] = 0x00; lsf[13] = 0x05;
        for (int i = 14; i < 28; ++i) lsf[i] = 0;
        uint32_t crc = 0xFFFF;
        for (int i = 0; i < 28; ++i) crc = ((crc << 8) ^ c_tab.crc[((crc >> 8) ^ lsf[i]) & 0xFF]) & 0xFFFF;
        lsf[28] = (uint8_t)(crc >> 8); lsf[29] = (uint8_t)crc;
    }
    const uint64_t d0 = 1 + (uint64_t)(4 + (int)(smix(seed, 1) % 3));      // draws used by the callsign
    uint8_t bits[368];
    if (slot == 3) {                                                        // link setup frame, P1
        gen_encode_punctured(lsf, 30, 1, bits, 0);
        gen_finish(m17::kSyncLinkSetup, bits, dst);
        if (d_lsf && cyc == 0) for (int i = 0; i < 30; ++i) d_lsf[(size_t)c * 30 + i] = lsf[i];
        return;
    }
    // stream frame f of cycle cyc (m17_fmt_add_stream_frame, m17_tx_routines.cpp:143-187)
    const int f = slot - 4;
    const long long sent = (long long)cyc * N + f;
    uint8_t body[18];
    body[0] = (uint8_t)(f >> 8); body[1] = (uint8_t)f;
    for (int h = 0; h < 2; ++h) {
        const uint64_t r = smix(seed, d0 + 2 * (uint64_t)sent + h + 1);
        for (int i = 0; i < 8; ++i) body[2 + 8 * h + i] = (uint8_t)(r >> (8 * i));
    }
    const int lich = f % 6;
    uint8_t chunk[6];
    for (int i = 0; i < 5; ++i) chunk[i] = lsf[lich * 5 + i];
    chunk[5] = (uint8_t)((lich & 7) << 5);
    const uint16_t w[4] = {
        (uint16_t)((chunk[0] << 4) | (chunk[1] >> 4)), (uint16_t)(((chunk[1] & 0xF) << 8) | chunk[2]),
        (uint16_t)((chunk[3] << 4) | (chunk[4] >> 4)), (uint16_t)(((chunk[4] & 0xF) << 8) | chunk[5]) };
    int n = 0;
    for (int k = 0; k < 4; ++k) {
        const uint32_t cw = ((uint32_t)w[k] << 12) | genc[w[k]];
        for (int b = 23; b >= 0; --b) bits[n++] = (uint8_t)((cw >> b) & 1);
    }
    gen_encode_punctured(body, 18, 2, bits, n);
    gen_finish(m17::kSyncStream, bits, dst);
    // the host loop sends frame (cyc, f) only while the buffer is not yet full at its start
    const long long start = (long long)gen_delay(ch) + (long long)g * 1920;
    if (start < (long long)A.nblk * kBlockSamples) {
        if (d_nframes) atomicAdd(&d_nframes[c], 1);
        if (d_payload && sent < max_payload_frames)
            for (int i = 0; i < 16; ++i) d_payload[((size_t)c * max_payload_frames + (size_t)sent) * 16 + i] = body[2 + i];
    }
}

// mod_filter (m17_modulate.cpp:49-61): sample m of the modulated stream = symbol m/10, branch 9 - m%10
__global__ __launch_bounds__(256)
void k_gen_sum(GenArgs A, const uint8_t *__restrict__ sym, const float *__restrict__ taps /* [310] */,
               float *__restrict__ sumv)
{
    const long long want = (long long)A.nblk * kBlockSamples;
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (long long)A.C * want) return;
    const int c = (int)(t / want);
    const long long smp = t - (long long)c * want;
    const long long m = smp - gen_delay((uint64_t)(A.first_channel + c));
    if (m < 0) { sumv[t] = 0.0f; return; }
    const long long s = m / 10;
    const int n = 9 - (int)(m - s * 10);
    const uint8_t *row = sym + (size_t)c * A.nslots * 192;
    float acc = 0.0f;
#pragma unroll 1
    for (int j = 0; j < 31; ++j) {
        const long long si = s - 30 + j;
        float dev = 0.0f;
        if (si >= 0) { const int code = row[si]; dev = (code == 255) ? 0.0f : A.lut[code & 3]; }
        const float p = dev * taps[n + 10 * j];
        acc = (j == 0) ? p : acc + p;
    }
    sumv[t] = acc;
}

// mod_fsk (m17_modulate.cpp:22-38): acc += sum per sample, wrapped into one turn after every symbol
__global__ __launch_bounds__(64)
void k_gen_phase(GenArgs A, float *__restrict__ sumv)
{
    const int c = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (c >= A.C) return;
    const long long want = (long long)A.nblk * kBlockSamples;
    const int delay = gen_delay((uint64_t)(A.first_channel + c));
    float *row = sumv + (size_t)c * want;
    float acc = 0.0f;
    for (long long i = 0; i < delay && i < want; ++i) row[i] = 0.0f;
    auto wrap = [](float a) {
        a = (float)((double)a / (2.0 * M_PI));
        double ip;
        a = (float)modf((double)a, &ip);
        return (float)((double)a * 2.0 * M_PI);
    };
    // four symbols per trip: 40 loads in flight, then the chain, then 40 stores
    constexpr int U = 40;
    long long i = delay;
    for (; i + U <= want; i += U) {
        float v[U];
#pragma unroll
        for (int k = 0; k < U; ++k) v[k] = row[i + k];
#pragma unroll
        for (int k = 0; k < U; ++k) {
            acc += v[k];
            v[k] = acc;
            if (k % 10 == 9) acc = wrap(acc);
        }
#pragma unroll
        for (int k = 0; k < U; ++k) row[i + k] = v[k];
    }
    int k = 0;
    for (; i < want; ++i) {
        acc += row[i];
        row[i] = acc;
        if (++k == 10) { k = 0; acc = wrap(acc); }
    }
}

struct NoiseArgs {
    int on, taps;                // taps = 63 (band limit) or 0 (white)
    double sigma;
    double h[63];
};

constexpr int GEN_SEG = 256;

__global__ __launch_bounds__(GEN_SEG)
void k_gen_iq(GenArgs A, NoiseArgs NZ, const float *__restrict__ phase, int16_t *__restrict__ iq)
{
    __shared__ double gr[GEN_SEG + 62], gi[GEN_SEG + 62];
    const long long want = (long long)A.nblk * kBlockSamples;
    const int segs = (int)((want + GEN_SEG - 1) / GEN_SEG);
    const int c = (int)(blockIdx.x / segs), seg = (int)(blockIdx.x - (long long)c * segs);
    const long long i0 = (long long)seg * GEN_SEG, i = i0 + threadIdx.x;
    const uint64_t ch = (uint64_t)(A.first_channel + c);
    const uint64_t nseed = (A.base_seed + ch) ^ 0xA36E0000A36E0000ull;
    if (NZ.on) {
        const int halo = NZ.taps ? 31 : 0;
        for (int q = (int)threadIdx.x; q < GEN_SEG + 2 * halo; q += GEN_SEG) {
            const long long j = i0 - halo + q;
            double a = 0.0, b = 0.0;
            if (j >= 0 && j < want) {                                    // Box-Muller on draws 2j+1, 2j+2
                const double u = smix_uniform(nseed, 2 * (uint64_t)j + 1), v = smix_uniform(nseed, 2 * (uint64_t)j + 2);
                const double r = sqrt(-2.0 * log(u));
                a = r * cos(2.0 * M_PI * v);
                b = r * sin(2.0 * M_PI * v);
            }
            gr[q] = a; gi[q] = b;
        }
        __syncthreads();
    }
    if (i >= want) return;
    const float ph = phase[(size_t)c * want + i];
    // single precision as the reference's overload resolution selects (m17_modulate.cpp:25-26)
    int re = (int)(int16_t)(cosf(ph) * (float)0x3FFF);
    int im = (int)(int16_t)(sinf(ph) * (float)0x3FFF);
    if (NZ.on) {
        double a, b;
        if (NZ.taps) {
            a = 0.0; b = 0.0;
            for (int k = 0; k < 63; ++k) {
                const long long j = i + k - 31;
                if (j >= 0 && j < want) { a += NZ.h[k] * gr[threadIdx.x + k]; b += NZ.h[k] * gi[threadIdx.x + k]; }
            }
        } else { a = gr[threadIdx.x]; b = gi[threadIdx.x]; }
        long long r2 = llrint((double)re + NZ.sigma * a), i2 = llrint((double)im + NZ.sigma * b);
        r2 = r2 > 32767 ? 32767 : (r2 < -32767 ? -32767 : r2);
        i2 = i2 > 32767 ? 32767 : (i2 < -32767 ? -32767 : i2);
        if (r2 == 0 && i2 == 0) r2 = 1;                      // the limiter divides by |z|
        re = (int)r2; im = (int)i2;
    }
    reinterpret_cast<uint32_t *>(iq)[(size_t)c * want + i] = (uint32_t)(uint16_t)re | ((uint32_t)(uint16_t)im << 16);
}

} // namespace m17dev
