// m17_pluto.hip -- k_pluto_decimate: wide-band ingest ahead of the hot path.
//
// Reference: rx_decimate_filter / sub_filter (radio.cpp:18-40) as driven by
// radio_receive_samples (:157-177): int16 IQ at 384 kHz -> 31-tap symmetric low-pass,
// keep every 8th -> int16 IQ at 48 kHz; int32 accumulate, arithmetic >> 15.  Integer
// arithmetic: bit-exact by construction.  This stage is HBM-bound (36 algorithmic bytes
// per output sample, ~70 integer ops): output i needs stream samples 8i-31 .. 8i-1.
//
// Mapping: lane l of a wave owns the 8-sample group S_m = x[8m-8 .. 8m-1] (two aligned
// 16-byte loads, lanes contiguous: 2 KB per wave, fully coalesced) for m = base-3+l; the
// other 23 window samples come from lanes l-1, l-2, l-3 by lane shuffles, so every input
// byte is fetched once per wave (3 of 64 lanes are halo).  Re and Im share a 32-bit
// register; tap pairs (k, 30-k) are evaluated with v_dot2_i32_i16 on re-packed halves.
namespace m17dev {

__constant__ int16_t c_dec[32];               // 31 taps (radio.cpp:45-51)

__device__ __forceinline__ int pack_lo(uint32_t a, uint32_t b) { return (int)((a & 0xFFFFu) | (b << 16)); }
__device__ __forceinline__ int pack_hi(uint32_t a, uint32_t b) { return (int)((a >> 16) | (b & 0xFFFF0000u)); }

typedef short v2s __attribute__((ext_vector_type(2)));

__device__ __forceinline__ int dot2(int ab, int cc, int acc)
{
    // acc + ab.lo*cc.lo + ab.hi*cc.hi in int32 (wraps like the reference's int arithmetic)
    return __builtin_amdgcn_sdot2(__builtin_bit_cast(v2s, ab), __builtin_bit_cast(v2s, cc), acc, false);
}

__global__ __launch_bounds__(256)
void k_pluto_decimate(const uint32_t *__restrict__ in,     // [C][n_in] packed (re | im << 16)
                      uint32_t *__restrict__ hist,         // [C][32]: x[-32 .. -1] of the stream (x[-32] unused)
                      uint32_t *__restrict__ out,          // [C][n_in/8]
                      int n_in, int waves_per_chan)
{
    const int lane = lane_id();
    const int gw = (int)(blockIdx.x * 4 + (threadIdx.x >> 6));
    const int chan = gw / waves_per_chan, wv = gw - chan * waves_per_chan;
    const int M = n_in >> 3;
    const int m = 61 * wv - 3 + lane;                      // group index owned by this lane
    if (61 * wv >= M) return;
    const uint32_t *x = in + (size_t)chan * n_in;
    const uint32_t *hx = hist + (size_t)chan * 32;

    // S_m = x[8m-8 .. 8m-1]; groups with m <= 0 lie in the history
    uint32_t s[8];
    if (m <= M) {
        const uint4 *p = (m >= 1) ? reinterpret_cast<const uint4 *>(x + 8 * m - 8)
                                  : reinterpret_cast<const uint4 *>(hx + 8 * (m + 3));
        const uint4 a = p[0], b = p[1];
        s[0] = a.x; s[1] = a.y; s[2] = a.z; s[3] = a.w; s[4] = b.x; s[5] = b.y; s[6] = b.z; s[7] = b.w;
    } else {
#pragma unroll
        for (int k = 0; k < 8; ++k) s[k] = 0;
    }
    // window w[j] = x[8i-31+j], i = m: w[0..6] = S_{m-3}[1..7], w[7..14] = S_{m-2}, w[15..22] = S_{m-1}, w[23..30] = S_m
    uint32_t w[31];
#pragma unroll
    for (int k = 0; k < 7; ++k) w[k] = (uint32_t)__shfl_up((int)s[k + 1], 3, 64);
#pragma unroll
    for (int k = 0; k < 8; ++k) w[7 + k] = (uint32_t)__shfl_up((int)s[k], 2, 64);
#pragma unroll
    for (int k = 0; k < 8; ++k) w[15 + k] = (uint32_t)__shfl_up((int)s[k], 1, 64);
#pragma unroll
    for (int k = 0; k < 8; ++k) w[23 + k] = s[k];

    // sub_filter (radio.cpp:18-33): centre tap first, then the 15 symmetric pairs in order
    const int cc15 = (int)c_dec[15];
    int re = (int)(short)(w[15] & 0xFFFF) * cc15;
    int im = ((int)w[15] >> 16) * cc15;
#pragma unroll
    for (int k = 0; k < 15; ++k) {
        const int ck = (int)(unsigned short)c_dec[k];
        const int cpair = ck | (ck << 16);
        re = dot2(pack_lo(w[k], w[30 - k]), cpair, re);    // c[k] * (in[k].re + in[30-k].re)
        im = dot2(pack_hi(w[k], w[30 - k]), cpair, im);
    }
    const uint32_t y = ((uint32_t)(re >> 15) & 0xFFFFu) | ((uint32_t)(im >> 15) << 16);
    const int i = m;                                       // output index
    if (lane >= 3 && i < M) out[(size_t)chan * M + i] = y;
}

// history for the next call: the last 31 samples of this call's input (m_rx_buff copy, radio.cpp:168)
__global__ void k_pluto_hist(const uint32_t *__restrict__ in, uint32_t *__restrict__ hist, int n_in, int C)
{
    const int t = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    const int chan = t >> 5, k = t & 31;
    if (chan >= C) return;
    hist[(size_t)chan * 32 + k] = in[(size_t)chan * n_in + n_in - 32 + k];
}

} // namespace m17dev
