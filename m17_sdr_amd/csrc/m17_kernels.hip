// m17_kernels.hip -- CDNA4 (gfx950) kernels of the batched M17 receive chain, first of the
// translation unit's parts (m17gpu_capi.hip includes them in order, one namespace).
//
// Data-parallel restatement of the reference's single-channel, file-static receive path
// (SURVEY.md section 8a).  What m17gpu_rx_blocks launches by default:
//
//   k_frontend_q      a3+a5+a6   int16 IQ -> limiter -> discriminator -> /5 -> DC sum      (this file)
//                     16 (channel, block) rows per wave, 4 lanes per row; the 1920-term DC sum is a
//                     strict sequential fp32 chain, one per row.
//   k_sync_frame_*    a9+a11+a12 timing recovery + sync correlator + framer                (m17_sync_*.hip)
//   k_worklist, k_decode_quad    a14..a24 demap / gather / Viterbi / Golay / packers       (m17_decode_quad.hip)
//   k_book_chan       a25 etc.   per-channel in-order LICH/LSF/packet bookkeeping          (m17_book.hip)
//
// Also here: the exact-arithmetic helpers with their exhaustive self tests, the sync correlator
// (SyncResult, sync_accept), the one-state-per-lane Viterbi (viterbi16) behind the stage entry
// points k_viterbi / k_demap / k_golay, and k_reset.
//
// Numeric contract: IEEE binary32, no FMA contraction, no
// re-association, correctly rounded sqrt/divide; the double-promoted
// expressions of the reference are evaluated in fp64.  Build with
// -ffp-contract=off and WITHOUT -ffast-math.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <type_traits>
#include "m17_dev.h"

#pragma clang fp contract(off)

namespace m17dev {

__constant__ DevTables c_tab;

typedef float v2f __attribute__((ext_vector_type(2)));

// ---------------------------------------------------------------------------
// small helpers
// ---------------------------------------------------------------------------
__device__ __forceinline__ int lane_id() { return (int)(threadIdx.x & 63); }

__device__ __forceinline__ float bcast_lane(float v, int src)
{
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), src));
}
__device__ __forceinline__ int bcast_lane_i(int v, int src) { return __builtin_amdgcn_readlane(v, src); }
__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ float unif(float v) { return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v))); }

// ---- exact arithmetic of the front end -------------------------------------
// Two implementations of each step: *_ref follows the reference expression
// literally (fp64 where the reference promotes, compiler-generated IEEE sqrt and
// divide); the default one is a shorter instruction sequence that returns the
// SAME bits.  k_selftest_* compares them exhaustively over the whole input
// domain on the device (m17gpu_selftest; run by tests/test_gpu_parity.py).

// dsp_short_to_float (m17_dsp.cpp:136-141): (float)((double)x * 0.00003)
__device__ __forceinline__ float s16_to_float_ref(int x) { return (float)((double)x * M17_LIT_S16_SCALE); }
// 0.00003 = CHI + CLO + 2.4e-20 with CHI = (float)0.00003.  fma(x, CHI, RN(x*CLO))
// equals the double-rounded reference for every int16 x (65,536 cases, verified
// on the host at build-test time and on the device by the self test).
__device__ __forceinline__ float s16_to_float(int x)
{
    const float xf = (float)x;
    return __builtin_fmaf(xf, 0x1.f75104p-16f, xf * 0x1.aaa3aep-41f);
}

// correctly rounded sqrt for normal, finite a (here 9e-10 <= a <= 2): Markstein's
// final step on v_rsq_f32 -- y0 = a*q, exact residual r = a - y0^2, y = y0 + r*(q/2).
// Equal to IEEE sqrt for every float in [8e-10, 2] (262,412,546 values, checked on the
// device by m17gpu_selftest; search program: scripts/micro/limit_seq.hip).  Four VALU
// slots shorter than v_sqrt_f32 + the two-sided residual select used before.
// a == 0 gives NaN here and 0 in the reference; both limit to NaN outputs.
__device__ __forceinline__ float sqrt_rn_normal(float a)
{
    const float q = __builtin_amdgcn_rsqf(a);
    const float y0 = a * q;
    const float r = __builtin_fmaf(-y0, y0, a);
    return __builtin_fmaf(r, q * 0.5f, y0);
}

// correctly rounded 1/m for normal m away from the exponent limits: one
// Newton-Raphson step on v_rcp_f32 (1 ulp), both operations fused.
__device__ __forceinline__ float rcp_rn_normal(float m)
{
    const float r0 = __builtin_amdgcn_rcpf(m);
    const float e = __builtin_fmaf(-m, r0, 1.0f);
    return __builtin_fmaf(e, r0, r0);
}

// dsp_limit (m17_dsp.cpp:412-419): m = sqrtf(re^2+im^2); g = (float)(1.0/m).
// (float)(1.0/(double)m) == correctly rounded 1.0f/m (double rounding is
// innocuous for division at 53 >= 2*24+2 bits), so fp32 IEEE divide is used.
__device__ __forceinline__ void limit_ref(float &re, float &im)
{
    const float m = __builtin_sqrtf(re * re + im * im);   // IEEE-correct under -fhip-fp32-correctly-rounded-divide-sqrt
    const float g = 1.0f / m;
    re = re * g;
    im = im * g;
}
__device__ __forceinline__ void limit(float &re, float &im)
{
#ifdef M17_REF_ARITH
    limit_ref(re, im);
#else
#ifdef M17_LIMIT_NO_RCP
    // EXPERIMENT (round 6, profiles/r06_limiter_three_newton_steps.txt; build: make norcp): 1 / m without v_rcp_f32 -- three
    // Newton steps from the v_rsq_f32 value the square root already has.  Two steps leave 432 of the 2^32 int16 pairs on the
    // wrong side of a rounding boundary (r05_limiter_without_rcp.txt); the third is the residual fix-up of the round-5 review.
    const float a = re * re + im * im;
    const float q = __builtin_amdgcn_rsqf(a);
    const float y0 = a * q;
    const float r = __builtin_fmaf(-y0, y0, a);
    const float m = __builtin_fmaf(r, q * 0.5f, y0);
    float g = q;
    g = __builtin_fmaf(__builtin_fmaf(-m, g, 1.0f), g, g);
    g = __builtin_fmaf(__builtin_fmaf(-m, g, 1.0f), g, g);
    g = __builtin_fmaf(__builtin_fmaf(-m, g, 1.0f), g, g);
    re = re * g;
    im = im * g;
#else
    const float m = sqrt_rn_normal(re * re + im * im);
    const float g = rcp_rn_normal(m);
    re = re * g;
    im = im * g;
#endif
#endif
}

// exhaustive device-side equivalence checks of the sequences above
__global__ void k_selftest_scale(unsigned *bad)
{
    const int x = (int)(blockIdx.x * blockDim.x + threadIdx.x) - 32768;
    if (x > 32767) return;
    const float a = s16_to_float(x), b = s16_to_float_ref(x);
    if (__float_as_uint(a) != __float_as_uint(b)) atomicAdd(bad, 1u);
}
__global__ void k_selftest_sqrt(unsigned lo, unsigned hi, unsigned *bad)
{
    for (unsigned long long u = lo + (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; u <= hi;
         u += (unsigned long long)gridDim.x * blockDim.x) {
        const float a = __uint_as_float((unsigned)u);
        const float f = sqrt_rn_normal(a), r = __builtin_sqrtf(a);
        if (__float_as_uint(f) != __float_as_uint(r) && !(f != f && r != r)) atomicAdd(bad, 1u);
    }
}
__global__ void k_selftest_rcp(unsigned lo, unsigned hi, unsigned *bad)
{
    for (unsigned long long u = lo + (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; u <= hi;
         u += (unsigned long long)gridDim.x * blockDim.x) {
        const float m = __uint_as_float((unsigned)u);
        const float f = rcp_rn_normal(m), r = 1.0f / m;
        if (__float_as_uint(f) != __float_as_uint(r) && !(f != f && r != r)) atomicAdd(bad, 1u);
    }
}
// the composed conversion + limiter on EVERY int16 pair (2^32 cases; (0,0) gives NaN both ways)
__global__ void k_selftest_limit(unsigned *bad)
{
    for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < (1ull << 32);
         i += (unsigned long long)gridDim.x * blockDim.x) {
        const int xr = (int)(i & 0xFFFF) - 32768, xi = (int)(i >> 16) - 32768;
        float ar = s16_to_float(xr), ai = s16_to_float(xi), br = s16_to_float_ref(xr), bi = s16_to_float_ref(xi);
        limit(ar, ai);
        limit_ref(br, bi);
        const bool same = (__float_as_uint(ar) == __float_as_uint(br) || (ar != ar && br != br)) &&
                          (__float_as_uint(ai) == __float_as_uint(bi) || (ai != ai && bi != bi));
        if (!same) atomicAdd(bad, 1u);
    }
}

// ---------------------------------------------------------------------------
// front end
// ---------------------------------------------------------------------------
__device__ __forceinline__ void wave_lds_sync()
{
    // LDS traffic of one wave executes in order; this only pins the compiler
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// z[0], z[1] of dsp_arctan_disc2 (m17_dsp.cpp:196,205-206) after the last block of the call of the channel whose
// block 0 is item cb: the limited last two samples of the channel's nblk blocks
__device__ __forceinline__ void fe_next_z(const uint4 *iq, int cb, int nblk, float &z0re, float &z0im, float &z1re, float &z1im)
{
    const uint32_t *pe = reinterpret_cast<const uint32_t *>(iq) + (size_t)(cb + nblk) * kBlockSamples;
    const uint32_t a = pe[-2], b = pe[-1];
    z1re = s16_to_float((int)(short)(a & 0xFFFF)); z1im = s16_to_float((int)a >> 16);
    z0re = s16_to_float((int)(short)(b & 0xFFFF)); z0im = s16_to_float((int)b >> 16);
    limit(z1re, z1im);
    limit(z0re, z0im);
}

// ---------------------------------------------------------------------------
// k_frontend_q: the stage with FOUR lanes per (channel, block), 16 per wave (one lane per row, round 1's first kernel,
// left 51,200 channel-blocks at 0.8 waves per SIMD and was slower at every size; removed in round 6).
//   load   : the wave fetches a [16 rows][64 samples] tile with 16 lanes per row,
//            i.e. 256 contiguous bytes per row per load instruction (measured:
//            6.4 TB/s for this pattern vs 3.6 TB/s for 16-byte pieces), one chunk
//            ahead in registers, staged through LDS;
//   compute: lane (row, sub) converts, limits and discriminates its 16 consecutive
//            samples; the two preceding limited samples come from the neighbour
//            lane by DPP (quad carry for sub 0);
//   sum    : only the DC sum is a chain: the u*0.5 values go back into the same LDS
//            tile and lane sub==0 of each quad adds them in sample order, picking
//            every 5th into an output tile that the quad stores as 256-byte rows.
// ---------------------------------------------------------------------------
constexpr int FQ_CHUNK  = 64;
constexpr int FQ_STRIDE = 68;              // 64 + 4 dwords (68 = 4*17): conflict-free b128 for all three access shapes
constexpr int FQ_WAVES  = 1;               // single-wave workgroups: finer placement over 256 CUs (measured 0.115 -> 0.107 ms at 51,200 rows)
constexpr int FQ_NCHUNK = kBlockSamples / FQ_CHUNK;   // 30

__device__ __forceinline__ float dpp_row_shr1(float v)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x111, 0xF, 0xF, true));
}
__device__ __forceinline__ float dpp_quad_b3(float v)      // broadcast lane 3 of every quad
{
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xFF, 0xF, 0xF, true));
}

template <int C5>
__device__ __forceinline__ void fq_sum_chunk(const float *row, float &offset, float *orow)
{
#pragma unroll
    for (int q = 0; q < FQ_CHUNK / 4; ++q) {
        const float4 v = reinterpret_cast<const float4 *>(row)[q];
        const float e[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int pos = C5 * FQ_CHUNK + q * 4 + k;      // position inside the 320-sample period
            offset += e[k];                                  // strictly sequential DC sum (m17_dsp.cpp:211)
            if (pos % 5 == 4) orow[pos / 5] = e[k];          // count%5==0 pick (m17_dsp.cpp:207-210)
        }
    }
}

#ifdef M17_STAMPS
__device__ unsigned long long g_fe_stamps[8];
__device__ unsigned long long g_fe_span[16384][2];        // start / end s_memtime of the first 16,384 workgroups
#define FSTAMP(i) do { __builtin_amdgcn_sched_barrier(0); const unsigned long long now_ = __builtin_amdgcn_s_memtime(); \
    __builtin_amdgcn_sched_barrier(0); facc_[i] += now_ - flast_; flast_ = now_; } while (0)
#else
#define FSTAMP(i) do {} while (0)
#endif
__global__ __launch_bounds__(64 * FQ_WAVES, 4)
void k_frontend_q(const uint4 *__restrict__ iq, ChanState *__restrict__ st,
                  float *__restrict__ disc_raw, float *__restrict__ offs,
                  int nblk, int total, int update_state)
{
    // total = C * nblk items, channel-major
    __shared__ __attribute__((aligned(16))) uint32_t tile[FQ_WAVES][16 * FQ_STRIDE];  // raw IQ, then u*0.5
    __shared__ __attribute__((aligned(16))) float otile[FQ_WAVES][16 * FQ_STRIDE];    // 64 picked outputs per row
    const int wave = (int)(threadIdx.x >> 6), lane = lane_id();
    const int cbl = lane >> 2, sub = lane & 3;
    const int cb0 = ((int)blockIdx.x * FQ_WAVES + wave) * 16;
    if (cb0 >= total) return;
    const bool valid = (cb0 + cbl) < total;
    const int cb = valid ? cb0 + cbl : total - 1;
    const int chan = cb / nblk, blk = cb - chan * nblk;
    uint32_t *my = tile[wave];
    float *myf = reinterpret_cast<float *>(tile[wave]);
    float *myo = otile[wave];

    // z[0], z[1] at the start of the block, identical in the four lanes of a quad
    // The state for the NEXT call -- the limited last two samples of the channel's last block -- is written by the quad of
    // block 0 itself, behind its own read (fe_next_z): written by the lane of the last block it could land before the
    // block-0 lane of another wave had read the old one (workgroups of different XCDs run far apart in a large grid;
    // seen as run-to-run differences above 65,536 channels).
    float c0re, c0im, c1re, c1im;
    float n0re = 0.0f, n0im = 0.0f, n1re = 0.0f, n1im = 0.0f;
    if (blk == 0) {
        c0re = st[chan].z0re; c0im = st[chan].z0im; c1re = st[chan].z1re; c1im = st[chan].z1im;
        fe_next_z(iq, cb, nblk, n0re, n0im, n1re, n1im);
    } else {
        const uint32_t *p = reinterpret_cast<const uint32_t *>(iq) + (size_t)cb * kBlockSamples;
        const uint32_t a = p[-2], b = p[-1];
        c1re = s16_to_float((int)(short)(a & 0xFFFF)); c1im = s16_to_float((int)a >> 16);
        c0re = s16_to_float((int)(short)(b & 0xFFFF)); c0im = s16_to_float((int)b >> 16);
        limit(c1re, c1im);
        limit(c0re, c0im);
    }

    // cooperative tile load: instruction j covers rows 4j..4j+3, 16 lanes x 16 B = 256 B per row
    // (named scalars, not arrays: the chunk body is a lambda and captured arrays end up in scratch)
    const int lr = lane >> 4, c16 = lane & 15;
#ifdef M17_STAMPS
    // instrumented build only (scripts/exp_fe_bound.py): bit 1 = read a 31 MB cache-resident window of the input instead of
    // the whole stream, bit 2 = do not store the discriminator stream -- what the stage costs without its HBM traffic
    const int dbg = update_state >> 1;
    update_state &= 1;
#endif
    auto row_ptr = [&](int j) {
        int row = cb0 + j * 4 + lr; row = row < total ? row : total - 1;
#ifdef M17_STAMPS
        if (dbg & 1) row &= 4095;
#endif
        return iq + (size_t)row * (kBlockSamples / 4) + c16;
    };
    const uint4 *g0 = row_ptr(0), *g1 = row_ptr(1), *g2 = row_ptr(2), *g3 = row_ptr(3);
    const int l0 = lr * FQ_STRIDE + c16 * 4, l1 = l0 + 4 * FQ_STRIDE, l2 = l0 + 8 * FQ_STRIDE, l3 = l0 + 12 * FQ_STRIDE;
    // the IQ stream is read exactly once: non-temporal, so that it does not evict the discriminator stream the
    // timing stage is about to re-read from the cache levels below
    typedef uint32_t u4v __attribute__((ext_vector_type(4)));
    auto ld = [](const uint4 *p) {
        const u4v v = __builtin_nontemporal_load(reinterpret_cast<const u4v *>(p));
        return make_uint4(v.x, v.y, v.z, v.w);
    };
    uint4 s0 = ld(g0), s1 = ld(g1), s2 = ld(g2), s3 = ld(g3);

    float offset = 0.0f;
    float *dst = disc_raw + (size_t)cb * kDiscOut;
#ifdef M17_STAMPS
    unsigned long long facc_[6] = {0, 0, 0, 0, 0, 0}, flast_ = __builtin_amdgcn_s_memtime();
    const unsigned long long fstart_ = flast_;
#endif

    auto chunk_body = [&](int chunk, auto c5tag) {
        constexpr int C5 = decltype(c5tag)::value;
        FSTAMP(5);
        // raw tile in, next chunk's loads out
        *reinterpret_cast<uint4 *>(&my[l0]) = s0;
        *reinterpret_cast<uint4 *>(&my[l1]) = s1;
        *reinterpret_cast<uint4 *>(&my[l2]) = s2;
        *reinterpret_cast<uint4 *>(&my[l3]) = s3;
        {
            const int nx = ((chunk + 1 < FQ_NCHUNK) ? chunk + 1 : chunk) * (FQ_CHUNK / 4);
            s0 = ld(g0 + nx); s1 = ld(g1 + nx); s2 = ld(g2 + nx); s3 = ld(g3 + nx);
        }
        wave_lds_sync();
        uint32_t w[16];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const uint4 v = *reinterpret_cast<const uint4 *>(&my[cbl * FQ_STRIDE + sub * 16 + q * 4]);
            w[4 * q] = v.x; w[4 * q + 1] = v.y; w[4 * q + 2] = v.z; w[4 * q + 3] = v.w;
        }
        wave_lds_sync();                                       // every lane holds its samples: the tile is free
        FSTAMP(0);
        float pre[16], pim[16];
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            pre[e] = s16_to_float((int)(short)(w[e] & 0xFFFF));
            pim[e] = s16_to_float((int)w[e] >> 16);
            limit(pre[e], pim[e]);
        }
        // the two samples in front of this lane's run: neighbour lane, or the carry for sub 0
        float p0re = dpp_row_shr1(pre[15]), p0im = dpp_row_shr1(pim[15]);
        float p1re = dpp_row_shr1(pre[14]), p1im = dpp_row_shr1(pim[14]);
        if (sub == 0) { p0re = c0re; p0im = c0im; p1re = c1re; p1im = c1im; }
        c0re = dpp_quad_b3(pre[15]); c0im = dpp_quad_b3(pim[15]);
        c1re = dpp_quad_b3(pre[14]); c1im = dpp_quad_b3(pim[14]);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float uh[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int e = 4 * q + k;
                // dsp_arctan_disc2 (m17_dsp.cpp:194-222): z0 = sample e-1, z1 = sample e-2
                const float z0re = (e >= 1) ? pre[e >= 1 ? e - 1 : 0] : p0re, z0im = (e >= 1) ? pim[e >= 1 ? e - 1 : 0] : p0im;
                const float z1re = (e >= 2) ? pre[e >= 2 ? e - 2 : 0] : (e == 1 ? p0re : p1re);
                const float z1im = (e >= 2) ? pim[e >= 2 ? e - 2 : 0] : (e == 1 ? p0im : p1im);
                const float aa = z0im * (pre[e] - z1re);
                const float bb = z0re * (pim[e] - z1im);
                uh[k] = (bb - aa) * M17_LIT_DISC_C;
            }
            *reinterpret_cast<float4 *>(&myf[cbl * FQ_STRIDE + sub * 16 + q * 4]) = make_float4(uh[0], uh[1], uh[2], uh[3]);
        }
        wave_lds_sync();
        FSTAMP(1);
        if (sub == 0) fq_sum_chunk<C5>(&myf[cbl * FQ_STRIDE], offset, &myo[cbl * FQ_STRIDE]);
        wave_lds_sync();
        FSTAMP(2);
    };

    for (int it = 0; it < FQ_NCHUNK / 5; ++it) {
        chunk_body(it * 5 + 0, std::integral_constant<int, 0>{});
        chunk_body(it * 5 + 1, std::integral_constant<int, 1>{});
        chunk_body(it * 5 + 2, std::integral_constant<int, 2>{});
        chunk_body(it * 5 + 3, std::integral_constant<int, 3>{});
        chunk_body(it * 5 + 4, std::integral_constant<int, 4>{});
        // 64 outputs per row: the quad stores its row's 256 bytes
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float4 v = *reinterpret_cast<const float4 *>(&myo[cbl * FQ_STRIDE + q * 16 + sub * 4]);
#ifdef M17_STAMPS
            if (dbg & 2) { if (v.x == 123.456f) *reinterpret_cast<float4 *>(dst) = v; continue; }
#endif
            if (valid) *reinterpret_cast<float4 *>(dst + it * 64 + q * 16 + sub * 4) = v;
        }
        wave_lds_sync();
    }
#ifdef M17_STAMPS
    if (blockIdx.x == 777 && lane == 0) for (int i = 0; i < 6; ++i) g_fe_stamps[i] = facc_[i];
    if (blockIdx.x < 16384 && lane == 0) { g_fe_span[blockIdx.x][0] = fstart_; g_fe_span[blockIdx.x][1] = __builtin_amdgcn_s_memtime(); }
#endif
    if (sub == 0 && valid) {
        offs[cb] = offset / (float)kBlockSamples;
        if (update_state && blk == 0) {
            st[chan].z0re = n0re; st[chan].z0im = n0im; st[chan].z1re = n1re; st[chan].z1im = n1im;
        }
    }
}

// ---------------------------------------------------------------------------
// k_frontend_d: k_frontend_q with the DC sum and the /5 pick kept in registers (round 4).
// k_frontend_q sends every u * 0.5 through LDS a second time -- four ds_write_b128 per lane and chunk (13 LDS-pipe
// cycles each on gfx950 and, 16 dwords apart in a row of 68, two-way bank-conflicted) and sixteen ds_read_b128 by the
// chain lane -- which made the CU's LDS pipe the busiest unit of the kernel (~0.22 of its 0.33 ms at 16,384 x 12: 44 %
// of its LDS cycles were bank conflicts, SQ_LDS_BANK_CONFLICT in profiles/r03_b_pmc_sq_full.txt).  Here the chain
// walks the quad instead: in step s the lane with sub == s adds its sixteen values, in sample order, to the sum its
// left neighbour finished in step s - 1 (quad_perm [0,0,1,2]; sub 0 starts from the row's sum so far).  Lanes whose
// turn has not come compute on stale input and are overwritten when it comes; lanes whose turn has passed recompute
// the same value from the same input.  Four steps of 1 + 1 + 16 instructions on all lanes replace 64 adds + 16 reads on a
// quarter of them.  The picks (sample position % 5 == 4) are at most four of a lane's sixteen values: entries e0, e0+5,
// e0+10 (and 15 when e0 == 0) with e0 = 4 - (lane's first position % 5), written straight into the output tile.
// ---------------------------------------------------------------------------
// dsp_short_to_float + dsp_limit (m17_dsp.cpp:136-141,412-419) of N samples (N even): the exact-arithmetic sequences of
// s16_to_float / sqrt_rn_normal / rcp_rn_normal on two-vectors (same IEEE operations in the same order per sample, so the
// same bits: m17gpu_selftest proves the scalar forms, test_stage_frontend compares this one with the oracle), every stage
// over all N samples before the next, so that no instruction waits on the one in front of it.
constexpr int FE_GROUP = 8;            // samples per fe_convert call in the tile kernels: two independent pair chains fill each other's wait states
template <int N>
__device__ __forceinline__ void fe_convert(const uint32_t *w, v2f *z)
{
    const v2f chi = {0x1.f75104p-16f, 0x1.f75104p-16f}, clo = {0x1.aaa3aep-41f, 0x1.aaa3aep-41f}, half = {0.5f, 0.5f}, one = {1.0f, 1.0f};
    v2f x[N], q[N];
    float a[N], r[N];
#pragma unroll
    for (int e = 0; e < N; ++e) x[e] = (v2f){(float)(int)(short)(w[e] & 0xFFFF), (float)((int)w[e] >> 16)};
#pragma unroll
    for (int e = 0; e < N; ++e) q[e] = x[e] * clo;
#pragma unroll
    for (int e = 0; e < N; ++e) x[e] = __builtin_elementwise_fma(x[e], chi, q[e]);          // s16_to_float
#pragma unroll
    for (int e = 0; e < N; ++e) q[e] = x[e] * x[e];
    // re * re + im * im as a plain add (left to the compiler: three moves and a packed add per sample pair)
#pragma unroll
    for (int e = 0; e < N; ++e) asm("v_add_f32 %0, %1, %2" : "=v"(a[e]) : "v"(q[e].x), "v"(q[e].y));
#pragma unroll
    for (int e = 0; e < N; ++e) r[e] = __builtin_amdgcn_rsqf(a[e]);
    v2f A[N / 2], Q[N / 2], Y[N / 2], R[N / 2], M[N / 2], R0[N / 2], E[N / 2], G[N / 2];
#pragma unroll
    for (int p = 0; p < N / 2; ++p) { A[p] = (v2f){a[2 * p], a[2 * p + 1]}; Q[p] = (v2f){r[2 * p], r[2 * p + 1]}; }
#pragma unroll
    for (int p = 0; p < N / 2; ++p) Y[p] = A[p] * Q[p];                                         // sqrt_rn_normal
#pragma unroll
    for (int p = 0; p < N / 2; ++p) Q[p] = Q[p] * half;
#pragma unroll
    for (int p = 0; p < N / 2; ++p) R[p] = __builtin_elementwise_fma(-Y[p], Y[p], A[p]);
#pragma unroll
    for (int p = 0; p < N / 2; ++p) M[p] = __builtin_elementwise_fma(R[p], Q[p], Y[p]);
#ifdef M17_LIMIT_NO_RCP
    // EXPERIMENT (see limit()): three packed Newton steps from the rsq value instead of v_rcp_f32 + one
#pragma unroll
    for (int p = 0; p < N / 2; ++p) G[p] = (v2f){r[2 * p], r[2 * p + 1]};
#pragma unroll
    for (int it = 0; it < 3; ++it) {
#pragma unroll
        for (int p = 0; p < N / 2; ++p) E[p] = __builtin_elementwise_fma(-M[p], G[p], one);
#pragma unroll
        for (int p = 0; p < N / 2; ++p) G[p] = __builtin_elementwise_fma(E[p], G[p], G[p]);
    }
    (void)R0;
#else
#pragma unroll
    for (int p = 0; p < N / 2; ++p) R0[p] = (v2f){__builtin_amdgcn_rcpf(M[p].x), __builtin_amdgcn_rcpf(M[p].y)};
#pragma unroll
    for (int p = 0; p < N / 2; ++p) E[p] = __builtin_elementwise_fma(-M[p], R0[p], one);        // rcp_rn_normal
#pragma unroll
    for (int p = 0; p < N / 2; ++p) G[p] = __builtin_elementwise_fma(E[p], R0[p], R0[p]);
#endif
#pragma unroll
    for (int p = 0; p < N / 2; ++p) { z[2 * p] = x[2 * p] * (v2f){G[p].x, G[p].x}; z[2 * p + 1] = x[2 * p + 1] * (v2f){G[p].y, G[p].y}; }
}

__device__ __forceinline__ float dpp_quad_rot(float v)        // quad lanes (0,1,2,3) read lanes (3,0,1,2)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x93, 0xF, 0xF, true));
}
__device__ __forceinline__ float dpp_quad_left(float v)       // quad lanes (0,1,2,3) read lanes (0,0,1,2)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x90, 0xF, 0xF, true));
}
// one tile = 16 rows; my / myo: the wave's raw and output tiles in LDS (16 x FQ_STRIDE dwords each).  Which (channel,
// block) row tile row i is, is the caller's: rowmap(i, valid) returns the row's index cb in the [C * nblk] row space
// (channel = cb / nblk) and whether its results are to be stored (rows past the end compute on a valid row's input).
template <class RowMap>
__device__ __forceinline__ void frontend_tile(const uint4 *__restrict__ iq, ChanState *__restrict__ st,
                                              float *__restrict__ disc_raw, float *__restrict__ offs,
                                              int nblk, int update_state, RowMap rowmap, uint32_t *my, float *myo, const int lane)
{
    const int cbl = lane >> 2, sub = lane & 3;
    bool valid;
    const int cb = rowmap(cbl, valid);
    const int chan = cb / nblk, blk = cb - chan * nblk;

    float c0re, c0im, c1re, c1im;
    float n0re = 0.0f, n0im = 0.0f, n1re = 0.0f, n1im = 0.0f;
    if (blk == 0) {
        c0re = st[chan].z0re; c0im = st[chan].z0im; c1re = st[chan].z1re; c1im = st[chan].z1im;
        fe_next_z(iq, cb, nblk, n0re, n0im, n1re, n1im);
    } else {
        const uint32_t *p = reinterpret_cast<const uint32_t *>(iq) + (size_t)cb * kBlockSamples;
        const uint32_t a = p[-2], b = p[-1];
        c1re = s16_to_float((int)(short)(a & 0xFFFF)); c1im = s16_to_float((int)a >> 16);
        c0re = s16_to_float((int)(short)(b & 0xFFFF)); c0im = s16_to_float((int)b >> 16);
        limit(c1re, c1im);
        limit(c0re, c0im);
    }

    const int lr = lane >> 4, c16 = lane & 15;
    auto row_ptr = [&](int j) {
        bool v_;
        const int row = rowmap(j * 4 + lr, v_);
        return iq + (size_t)row * (kBlockSamples / 4) + c16;
    };
    const uint4 *g0 = row_ptr(0), *g1 = row_ptr(1), *g2 = row_ptr(2), *g3 = row_ptr(3);
    const int l0 = lr * FQ_STRIDE + c16 * 4, l1 = l0 + 4 * FQ_STRIDE, l2 = l0 + 8 * FQ_STRIDE, l3 = l0 + 12 * FQ_STRIDE;
    typedef uint32_t u4v __attribute__((ext_vector_type(4)));
    auto ld = [](const uint4 *p) {
        const u4v v = __builtin_nontemporal_load(reinterpret_cast<const u4v *>(p));
        return make_uint4(v.x, v.y, v.z, v.w);
    };
    uint4 s0 = ld(g0), s1 = ld(g1), s2 = ld(g2), s3 = ld(g3);

    float tsum = 0.0f;                                        // TWICE the row's DC sum so far, in lane 3 of the quad
    float *dst = disc_raw + (size_t)cb * kDiscOut;
    const bool sub0 = sub == 0, is1 = sub == 1, is2 = sub == 2, is3 = sub == 3;
    float *orow = &myo[cbl * FQ_STRIDE];

    auto chunk_body = [&](int chunk, auto c5tag) {
        constexpr int C5 = decltype(c5tag)::value;
        *reinterpret_cast<uint4 *>(&my[l0]) = s0;
        *reinterpret_cast<uint4 *>(&my[l1]) = s1;
        *reinterpret_cast<uint4 *>(&my[l2]) = s2;
        *reinterpret_cast<uint4 *>(&my[l3]) = s3;
        {
            const int nx = ((chunk + 1 < FQ_NCHUNK) ? chunk + 1 : chunk) * (FQ_CHUNK / 4);
            s0 = ld(g0 + nx); s1 = ld(g1 + nx); s2 = ld(g2 + nx); s3 = ld(g3 + nx);
        }
        wave_lds_sync();
        uint32_t w[16];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const uint4 v = *reinterpret_cast<const uint4 *>(&my[cbl * FQ_STRIDE + sub * 16 + q * 4]);
            w[4 * q] = v.x; w[4 * q + 1] = v.y; w[4 * q + 2] = v.z; w[4 * q + 3] = v.w;
        }
        wave_lds_sync();                                       // every lane holds its samples: the tile is free
        // conversion + limiter, eight samples at a time and STAGE BY STAGE (fe_convert, FE_GROUP samples at a time): written sample by sample the
        // compiler emits each sample pair's eleven dependent steps back to back with an s_nop behind nearly every one
        // (417 s_nop in this kernel's five-chunk body, round 5)
        v2f z[16];
#pragma unroll
        for (int e = 0; e < 16; e += FE_GROUP) fe_convert<FE_GROUP>(&w[e], &z[e]);
        // samples -1 and -2 of this lane's run: the left neighbour's last two; for sub 0 lane 3's of the PREVIOUS chunk,
        // which the same rotation of that chunk left in c0 / c1 (one lane move and one select per component)
        const v2f r0 = {dpp_quad_rot(z[15].x), dpp_quad_rot(z[15].y)}, r1 = {dpp_quad_rot(z[14].x), dpp_quad_rot(z[14].y)};
        const v2f p0 = {sub0 ? c0re : r0.x, sub0 ? c0im : r0.y};
        const v2f p1 = {sub0 ? c1re : r1.x, sub0 ? c1im : r1.y};
        c0re = r0.x; c0im = r0.y; c1re = r1.x; c1im = r1.y;
        // dsp_arctan_disc2 (m17_dsp.cpp:194-222): z0 = sample e-1, z1 = sample e-2;  u = (z0.re (im - z1.im) - z0.im (re - z1.re)) / 2.
        // u2 = 2 u: the halving is applied to the picks and to the block's sum only (exact: see frontend_lite_tile)
        float u2[16];
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const v2f z0 = (e >= 1) ? z[e >= 1 ? e - 1 : 0] : p0;
            const v2f z1 = (e >= 2) ? z[e >= 2 ? e - 2 : 0] : (e == 1 ? p0 : p1);
            const v2f d = z[e] - z1;
            const v2f pr = d * (v2f){z0.y, z0.x};                          // (aa, bb)
            u2[e] = pr.y - pr.x;
        }
        // ---- strictly sequential DC sum (m17_dsp.cpp:211) through the quad; the lane moves ride on the first add of every step
        float T = dpp_quad_b3(tsum) + u2[0];
#pragma unroll
        for (int e = 1; e < 16; ++e) T = T + u2[e];
#pragma unroll
        for (int s = 1; s < 4; ++s) {
            T = dpp_quad_left(T) + u2[0];
#pragma unroll
            for (int e = 1; e < 16; ++e) T = T + u2[e];
        }
        tsum = T;
        // ---- count % 5 == 0 pick (m17_dsp.cpp:207-210): positions 64 C5 + 16 sub + e of the 320-sample period; the lane's
        // first pick is entry e0 = 4 - (first position % 5) = 4 - (4 C5 + sub) % 5 (64 % 5 == 4, 16 % 5 == 1), then e0 + 5,
        // e0 + 10 (and 15 when e0 == 0): compile-time per (C5, sub), selected by the three lane-class masks
        {
            constexpr int E0 = 4 - (4 * C5) % 5, E1 = 4 - (4 * C5 + 1) % 5, E2 = 4 - (4 * C5 + 2) % 5, E3 = 4 - (4 * C5 + 3) % 5;
            float v0 = u2[E0], v1 = u2[E0 + 5], v2 = u2[E0 + 10];
            v0 = is1 ? u2[E1] : v0;  v1 = is1 ? u2[E1 + 5] : v1;  v2 = is1 ? u2[E1 + 10] : v2;
            v0 = is2 ? u2[E2] : v0;  v1 = is2 ? u2[E2 + 5] : v1;  v2 = is2 ? u2[E2 + 10] : v2;
            v0 = is3 ? u2[E3] : v0;  v1 = is3 ? u2[E3 + 5] : v1;  v2 = is3 ? u2[E3 + 10] : v2;
            const int e0 = 4 - (C5 * 4 + sub) % 5;
            const int o0 = (C5 * 64 + 16 * sub) / 5;           // output index of the first pick within the period (position e0 is the one with % 5 == 4)
            const v2f h01 = (v2f){v0, v1} * (v2f){M17_LIT_DISC_C, M17_LIT_DISC_C}, h23 = (v2f){v2, u2[15]} * (v2f){M17_LIT_DISC_C, M17_LIT_DISC_C};
            orow[o0] = h01.x;
            orow[o0 + 1] = h01.y;
            orow[o0 + 2] = h23.x;
            if (e0 == 0) orow[o0 + 3] = h23.y;
        }
    };

    for (int it = 0; it < FQ_NCHUNK / 5; ++it) {
        chunk_body(it * 5 + 0, std::integral_constant<int, 0>{});
        chunk_body(it * 5 + 1, std::integral_constant<int, 1>{});
        chunk_body(it * 5 + 2, std::integral_constant<int, 2>{});
        chunk_body(it * 5 + 3, std::integral_constant<int, 3>{});
        chunk_body(it * 5 + 4, std::integral_constant<int, 4>{});
        wave_lds_sync();
        // 64 outputs per row: the quad stores its row's 256 bytes
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float4 v = *reinterpret_cast<const float4 *>(&myo[cbl * FQ_STRIDE + q * 16 + sub * 4]);
            if (valid) *reinterpret_cast<float4 *>(dst + it * 64 + q * 16 + sub * 4) = v;
        }
        wave_lds_sync();
    }
    const float offset = dpp_quad_b3(tsum) * M17_LIT_DISC_C;
    if (sub0 && valid) {
        offs[cb] = offset / (float)kBlockSamples;
        if (update_state && blk == 0) {
            st[chan].z0re = n0re; st[chan].z0im = n0im; st[chan].z1re = n1re; st[chan].z1im = n1im;
        }
    }
}
template <int CTRL> __device__ __forceinline__ float dpp_keep(float old, float src)    // lanes without a source keep `old`
{
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(old), __float_as_int(src), CTRL, 0xF, 0xF, false));
}
// value of lane l-1 of the row; lane 0: lane 15's value of `prev` (the row's previous chunk)
__device__ __forceinline__ float fu_left(float cur, float prev)
{
    const float t = dpp_keep<0x121>(prev, prev);          // row_ror:1
    return dpp_keep<0x111>(t, cur);                       // row_shr:1
}

// A QUICK tile: FOUR rows of sixteen lanes (k_rx_fused's row mapping, m17_fused.hip, as a device function that stores to
// the workspace like frontend_tile).  A lone wave needs 23 us for it against 40 for a sixteen-row tile, whatever that
// holds -- each lane converts 120 samples instead of 480, the 1,920-step DC chain costs the same -- so it is what the
// front-end wave (and the idle framer wave) of a channel in k_sync_frame_duo<1> run on short calls (round 5).
//   load    : lane (r, l) of chunk c reads the two uint4 with samples 128c + 8l .. + 7 of row r (512 contiguous bytes per row)
//   compute : fe_convert<8> (packed fp32, stage by stage) on the lane's own eight samples; the two before them by DPP
//             row_shr:1 (lane 0: the row's previous chunk); the discriminator without its halving (u2 = 2u: halved at the
//             picks and at the block's sum; exact, see frontend_lite_tile)
//   DC sum  : the chain runs lane by lane through the row as 16 x 8 dependent adds per chunk (fu_chain8)
//   /5 pick : positions s = 128c + 8l + k with s % 5 == 4 are the outputs: with q = (3c + 3l) % 5 the lane's first pick is
//             its sample k0 = 4 - q, its second one k0 + 5 when k0 <= 2; stored straight to their places in the row
// rowmap(i, valid), i = 0 .. 3: as for frontend_tile.
// fu_chain8 (the eight-sample form of fu_chain, m17_fused.hip): on entry `carry` holds, in lane 0 of each row, twice the row's
// sum so far; u the lane's eight values, a the same with exact zeros in lane 0.  Lane 0 finishes in the first eight adds;
// after step j lanes 0..j hold their final sums (lane l <= j recomputes the same value from lane l-1's final one; lane 0
// has no source for the DPP add -- bound_ctrl 0: it keeps its value -- and adds zeros, which is exact: a running sum that
// starts at +0 never is -0).  Leaves the row's new sum in lane 0 of `carry`.  (s_nop 1: a VALU write followed by a DPP read
// of the same register needs two wait states on gfx9.)
#define FU_STEP8 "s_nop 1\n\tv_add_f32_dpp %0, %0, %10 row_shr:1 row_mask:0xf bank_mask:0xf\n\t" \
                 "v_add_f32 %0, %0, %11\n\tv_add_f32 %0, %0, %12\n\tv_add_f32 %0, %0, %13\n\tv_add_f32 %0, %0, %14\n\t" \
                 "v_add_f32 %0, %0, %15\n\tv_add_f32 %0, %0, %16\n\tv_add_f32 %0, %0, %17\n\t"
__device__ __forceinline__ void fu_chain8(float &carry, const float (&u)[8], const float (&a)[8])
{
    float T;
    asm volatile("v_add_f32 %0, %1, %2\n\tv_add_f32 %0, %0, %3\n\tv_add_f32 %0, %0, %4\n\tv_add_f32 %0, %0, %5\n\t"
                 "v_add_f32 %0, %0, %6\n\tv_add_f32 %0, %0, %7\n\tv_add_f32 %0, %0, %8\n\tv_add_f32 %0, %0, %9\n\t"
                 FU_STEP8 FU_STEP8 FU_STEP8 FU_STEP8 FU_STEP8 FU_STEP8 FU_STEP8 FU_STEP8 FU_STEP8 FU_STEP8 FU_STEP8 FU_STEP8 FU_STEP8 FU_STEP8 FU_STEP8
                 "s_nop 1\n\tv_mov_b32_dpp %1, %0 row_ror:1 row_mask:0xf bank_mask:0xf"
                 : "=&v"(T), "+v"(carry)
                 : "v"(u[0]), "v"(u[1]), "v"(u[2]), "v"(u[3]), "v"(u[4]), "v"(u[5]), "v"(u[6]), "v"(u[7]),
                   "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(a[4]), "v"(a[5]), "v"(a[6]), "v"(a[7]));
}
template <class RowMap>
__device__ __forceinline__ void frontend_quick4p(const uint4 *__restrict__ iq, ChanState *__restrict__ st,
                                                 float *__restrict__ disc_raw, float *__restrict__ offs,
                                                 int nblk, int update_state, RowMap rowmap, const int lane)
{
    constexpr int P = 5, NCHUNK = kBlockSamples / 128;    // 15 chunks, three rounds of five
    static_assert(NCHUNK % P == 0, "quick tile passes");
    const int r = lane >> 4, l = lane & 15;
    bool valid;
    const int cb = rowmap(r, valid);
    const int chan = cb / nblk, blk = cb - chan * nblk;
    float p3re, p3im, p2re, p2im;                          // sample -1 (z0) and -2 (z1) of the row, in every lane
    float n0re = 0.0f, n0im = 0.0f, n1re = 0.0f, n1im = 0.0f;
    if (blk == 0) {
        p3re = st[chan].z0re; p3im = st[chan].z0im; p2re = st[chan].z1re; p2im = st[chan].z1im;
        fe_next_z(iq, cb, nblk, n0re, n0im, n1re, n1im);
    } else {
        const uint32_t *pw = reinterpret_cast<const uint32_t *>(iq) + (size_t)cb * kBlockSamples;
        const uint32_t a = pw[-2], b = pw[-1];
        p2re = s16_to_float((int)(short)(a & 0xFFFF)); p2im = s16_to_float((int)a >> 16);
        p3re = s16_to_float((int)(short)(b & 0xFFFF)); p3im = s16_to_float((int)b >> 16);
        limit(p2re, p2im);
        limit(p3re, p3im);
    }
    typedef uint32_t u4v __attribute__((ext_vector_type(4)));
    const uint4 *const rowp = iq + (size_t)cb * (kBlockSamples / 4) + 2 * l;
    float *const dst = disc_raw + (size_t)cb * kDiscOut;
    float carry = 0.0f;                                    // TWICE the row's DC sum so far, in lane 0 of the row
    int q = (3 * l) % 5;                                   // (3c + 3l) % 5
    const bool first = l == 0;
    auto chunk = [&](const u4v v0, const u4v v1, const int c) {
        const uint32_t w[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
        v2f z[8];
        fe_convert<8>(w, z);
        const v2f m1 = {fu_left(z[7].x, p3re), fu_left(z[7].y, p3im)};            // sample -1 of this lane's run
        const v2f m2 = {fu_left(z[6].x, p2re), fu_left(z[6].y, p2im)};            // sample -2
        p3re = z[7].x; p3im = z[7].y; p2re = z[6].x; p2im = z[6].y;
        float u2[8], a[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            // dsp_arctan_disc2 (m17_dsp.cpp:194-222): z0 = sample e-1, z1 = sample e-2; u2 = 2 u
            const v2f z0 = (e >= 1) ? z[e >= 1 ? e - 1 : 0] : m1;
            const v2f z1 = (e >= 2) ? z[e >= 2 ? e - 2 : 0] : (e == 1 ? m1 : m2);
            const v2f d = z[e] - z1;
            const v2f pr = d * (v2f){z0.y, z0.x};                                  // (aa, bb)
            u2[e] = pr.y - pr.x;
            a[e] = first ? 0.0f : u2[e];
        }
        fu_chain8(carry, u2, a);
        // count % 5 == 0 picks (m17_dsp.cpp:207-210)
        const int k0 = q == 0 ? 4 : 4 - q;
        const float s1 = q == 4 ? u2[0] : (q == 3 ? u2[1] : (q == 2 ? u2[2] : (q == 1 ? u2[3] : u2[4])));
        const float s2 = q == 4 ? u2[5] : (q == 3 ? u2[6] : u2[7]);
        const v2f h = (v2f){s1, s2} * (v2f){M17_LIT_DISC_C, M17_LIT_DISC_C};
        const unsigned oidx = ((unsigned)(128 * c + 8 * l + k0 - 4) * 52429u) >> 18;       // (s - 4) / 5
        if (valid) {
            dst[oidx] = h.x;
            if (q >= 2) dst[oidx + 1] = h.y;
        }
        q = q >= 2 ? q - 2 : q + 3;
    };
    // five chunks of input in flight, each in its own registers: slot j holds chunk c0 + j and is loaded again (chunk
    // c0 + j + 5) as soon as it has been consumed -- a lone wave streams its rows against the full memory latency
    auto ld = [&](int c, int half) { return __builtin_nontemporal_load(reinterpret_cast<const u4v *>(rowp + c * 32 + half)); };
    u4v ws[2 * P];
#pragma unroll
    for (int j = 0; j < P; ++j) { ws[2 * j] = ld(j, 0); ws[2 * j + 1] = ld(j, 1); }
    for (int c0 = 0; c0 < NCHUNK; c0 += P) {
#pragma unroll
        for (int j = 0; j < P; ++j) {
            const u4v v0 = ws[2 * j], v1 = ws[2 * j + 1];
            const int nx = min(c0 + j + P, NCHUNK - 1);              // behind the row's end: its last chunk again, unused
            ws[2 * j] = ld(nx, 0); ws[2 * j + 1] = ld(nx, 1);
            chunk(v0, v1, c0 + j);
        }
    }
    // offset / len (m17_dsp.cpp:213): twice the row's sum sits in lane 0 of `carry`
    if (first && valid) {
        offs[cb] = (carry * M17_LIT_DISC_C) / (float)kBlockSamples;
        if (update_state && blk == 0) {
            st[chan].z0re = n0re; st[chan].z0im = n0im; st[chan].z1re = n1re; st[chan].z1im = n1im;
        }
    }
}

// the 16 consecutive rows cb0 .. cb0 + 15 of the row space (rows from `total` on are not stored)
__device__ __forceinline__ void frontend_d_tile(const uint4 *__restrict__ iq, ChanState *__restrict__ st,
                                                float *__restrict__ disc_raw, float *__restrict__ offs,
                                                int nblk, int total, int update_state, const int cb0, uint32_t *my, float *myo,
                                                const int lane = lane_id())
{
    if (cb0 >= total) return;
    frontend_tile(iq, st, disc_raw, offs, nblk, update_state,
                  [&](int i, bool &valid) { valid = cb0 + i < total; return valid ? cb0 + i : total - 1; }, my, myo, lane);
}
// rows g0 .. g0 + 15 of a GROUP: blocks b0 .. b0 + bc - 1 of the channels chan0, chan0 + 1, ... numbered channel by
// channel (group row g = channel g / bc, block b0 + g % bc); `grows` rows in the group
__device__ __forceinline__ void frontend_g_tile(const uint4 *__restrict__ iq, ChanState *__restrict__ st,
                                                float *__restrict__ disc_raw, float *__restrict__ offs,
                                                int nblk, int chan0, int b0, int bc, int grows, const int g0, uint32_t *my, float *myo,
                                                const int lane)
{
    frontend_tile(iq, st, disc_raw, offs, nblk, 1,
                  [&](int i, bool &valid) {
                      valid = g0 + i < grows;
                      const int g = valid ? g0 + i : grows - 1;
                      const int c = g / bc;
                      return (chan0 + c) * nblk + b0 + (g - c * bc);
                  }, my, myo, lane);
}
// ---------------------------------------------------------------------------
// frontend_lite_tile (round 5): the tile of frontend_tile at HALF the chunk -- 32 samples per row and chunk, eight per
// lane -- for the kernels that have to share a SIMD with five other waves: ~60 live VGPRs instead of ~106 and 4.6 KB of
// LDS instead of 8.7 (a raw tile and an output tile of 16 x 36 dwords; 32 outputs per row = 160 samples = five chunks
// between two stores of 128 B per row).  Same arithmetic, same order: the DC chain walks the quad in four steps of eight
// adds, the picks (sample position % 5 == 4) are the lane's entries e0 and e0 + 5 with e0 = 4 - (first position % 5).
// ---------------------------------------------------------------------------
constexpr int FL_CHUNK = 32, FL_STRIDE = 36, FL_NCHUNK = kBlockSamples / FL_CHUNK;    // 60 chunks per block
constexpr int FL_TILE_BYTES = 16 * FL_STRIDE * 4;                                      // 2,304 B each, raw and output
// (five chunks of input in flight per wave instead of one -- 40 more VGPRs -- changed nothing: profiles/r05_rx_chan_variants.txt)
template <class RowMap>
__device__ __forceinline__ void frontend_lite_tile(const uint4 *__restrict__ iq, ChanState *__restrict__ st,
                                                   float *__restrict__ disc_raw, float *__restrict__ offs,
                                                   int nblk, int update_state, RowMap rowmap, uint32_t *my, float *myo, const int lane)
{
    const int cbl = lane >> 2, sub = lane & 3;
    bool valid;
    const int cb = rowmap(cbl, valid);
    const int chan = cb / nblk, blk = cb - chan * nblk;

    float c0re, c0im, c1re, c1im;
    float n0re = 0.0f, n0im = 0.0f, n1re = 0.0f, n1im = 0.0f;
    if (blk == 0) {
        c0re = st[chan].z0re; c0im = st[chan].z0im; c1re = st[chan].z1re; c1im = st[chan].z1im;
        fe_next_z(iq, cb, nblk, n0re, n0im, n1re, n1im);
    } else {
        const uint32_t *p = reinterpret_cast<const uint32_t *>(iq) + (size_t)cb * kBlockSamples;
        const uint32_t a = p[-2], b = p[-1];
        c1re = s16_to_float((int)(short)(a & 0xFFFF)); c1im = s16_to_float((int)a >> 16);
        c0re = s16_to_float((int)(short)(b & 0xFFFF)); c0im = s16_to_float((int)b >> 16);
        limit(c1re, c1im);
        limit(c0re, c0im);
    }

    // cooperative load: instruction j covers rows 8j .. 8j + 7, 8 lanes x 16 B = one 128-byte line per row
    const int lr = lane >> 3, c8 = lane & 7;
    auto row_ptr = [&](int j) {
        bool v_;
        const int row = rowmap(j * 8 + lr, v_);
        return iq + (size_t)row * (kBlockSamples / 4) + c8;
    };
    const uint4 *g0 = row_ptr(0), *g1 = row_ptr(1);
    const int l0 = lr * FL_STRIDE + c8 * 4, l1 = l0 + 8 * FL_STRIDE;
    typedef uint32_t u4v __attribute__((ext_vector_type(4)));
    auto ld = [](const uint4 *p) {
        const u4v v = __builtin_nontemporal_load(reinterpret_cast<const u4v *>(p));
        return make_uint4(v.x, v.y, v.z, v.w);
    };
    uint4 s0 = ld(g0), s1 = ld(g1);

    float tsum = 0.0f;                                        // TWICE the row's DC sum so far, in lane 3 of the quad
    float *dst = disc_raw + (size_t)cb * kDiscOut;
    const bool sub0 = sub == 0, is1 = sub == 1, is2 = sub == 2, is3 = sub == 3;
    float *orow = &myo[cbl * FL_STRIDE];

    auto chunk_body = [&](int chunk, auto c5tag) {
        constexpr int C5 = decltype(c5tag)::value;
        *reinterpret_cast<uint4 *>(&my[l0]) = s0;
        *reinterpret_cast<uint4 *>(&my[l1]) = s1;
        {
            const int nx = ((chunk + 1 < FL_NCHUNK) ? chunk + 1 : chunk) * (FL_CHUNK / 4);
            s0 = ld(g0 + nx); s1 = ld(g1 + nx);
        }
        wave_lds_sync();
        uint32_t w[8];
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const uint4 v = *reinterpret_cast<const uint4 *>(&my[cbl * FL_STRIDE + sub * 8 + q * 4]);
            w[4 * q] = v.x; w[4 * q + 1] = v.y; w[4 * q + 2] = v.z; w[4 * q + 3] = v.w;
        }
        wave_lds_sync();                                       // every lane holds its samples: the tile is free
        v2f z[8];
        fe_convert<4>(&w[0], &z[0]);
        fe_convert<4>(&w[4], &z[4]);
        // samples -1 and -2 of this lane's run: the left neighbour's last two; for sub 0 lane 3's of the PREVIOUS chunk,
        // which the same rotation of that chunk left in c0 / c1 (one lane move and one select per component)
        const v2f r0 = {dpp_quad_rot(z[7].x), dpp_quad_rot(z[7].y)}, r1 = {dpp_quad_rot(z[6].x), dpp_quad_rot(z[6].y)};
        const v2f p0 = {sub0 ? c0re : r0.x, sub0 ? c0im : r0.y};
        const v2f p1 = {sub0 ? c1re : r1.x, sub0 ? c1im : r1.y};
        c0re = r0.x; c0im = r0.y; c1re = r1.x; c1im = r1.y;
        // dsp_arctan_disc2 (m17_dsp.cpp:194-222): z0 = sample e-1, z1 = sample e-2;  u = (z0.re (im - z1.im) - z0.im (re - z1.re)) / 2.
        // u2 = 2 u: the halving is applied to the picks and to the block's sum only.  Exact: every nonzero z component is
        // >= 2^-16 in magnitude, every nonzero difference of two >= 2^-39, every product >= 2^-55 and every difference
        // of products and partial sum of them a multiple of 2^-78 -- nothing comes near the subnormal range, where alone
        // a scaling by two does not commute with rounding.
        float u2[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const v2f z0 = (e >= 1) ? z[e >= 1 ? e - 1 : 0] : p0;
            const v2f z1 = (e >= 2) ? z[e >= 2 ? e - 2 : 0] : (e == 1 ? p0 : p1);
            const v2f d = z[e] - z1;
            const v2f pr = d * (v2f){z0.y, z0.x};                          // (aa, bb)
            u2[e] = pr.y - pr.x;
        }
        // ---- strictly sequential DC sum (m17_dsp.cpp:211) through the quad: in step s the lane with sub == s adds its
        // eight values to the sum its left neighbour finished in step s - 1; sub 0 starts from the sum lane 3 finished in
        // the previous chunk.  The lane moves ride on the first add of every step (v_add_f32_dpp).
        float T = dpp_quad_b3(tsum) + u2[0];
#pragma unroll
        for (int e = 1; e < 8; ++e) T = T + u2[e];
#pragma unroll
        for (int s_ = 1; s_ < 4; ++s_) {
            T = dpp_quad_left(T) + u2[0];
#pragma unroll
            for (int e = 1; e < 8; ++e) T = T + u2[e];
        }
        tsum = T;
        // ---- count % 5 == 0 pick (m17_dsp.cpp:207-210): positions 32 C5 + 8 sub + e of the 160-sample period; the lane's
        // first pick is entry e0 = 4 - (first position % 5) = 4 - (2 C5 + 3 sub) % 5 (32 % 5 == 2, 8 % 5 == 3), its second
        // e0 + 5 when e0 <= 2: compile-time per (C5, sub), selected by the three lane-class masks
        {
            constexpr int E0 = 4 - (2 * C5) % 5, E1 = 4 - (2 * C5 + 3) % 5, E2 = 4 - (2 * C5 + 6) % 5, E3 = 4 - (2 * C5 + 9) % 5;
            float v0 = u2[E0], v1 = u2[E0 <= 2 ? E0 + 5 : 7];
            v0 = is1 ? u2[E1] : v0;  v1 = is1 ? u2[E1 <= 2 ? E1 + 5 : 7] : v1;
            v0 = is2 ? u2[E2] : v0;  v1 = is2 ? u2[E2 <= 2 ? E2 + 5 : 7] : v1;
            v0 = is3 ? u2[E3] : v0;  v1 = is3 ? u2[E3 <= 2 ? E3 + 5 : 7] : v1;
            const int e0 = 4 - (2 * C5 + 3 * sub) % 5;
            const int o0 = (C5 * 32 + 8 * sub + e0) / 5;       // its output index within the period
            const v2f h = (v2f){v0, v1} * (v2f){M17_LIT_DISC_C, M17_LIT_DISC_C};
            orow[o0] = h.x;
            if (e0 <= 2) orow[o0 + 1] = h.y;
        }
    };

    for (int it = 0; it < FL_NCHUNK / 5; ++it) {
        chunk_body(it * 5 + 0, std::integral_constant<int, 0>{});
        chunk_body(it * 5 + 1, std::integral_constant<int, 1>{});
        chunk_body(it * 5 + 2, std::integral_constant<int, 2>{});
        chunk_body(it * 5 + 3, std::integral_constant<int, 3>{});
        chunk_body(it * 5 + 4, std::integral_constant<int, 4>{});
        wave_lds_sync();
        // 32 outputs per row: the quad stores its row's 128 bytes
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const float4 v = *reinterpret_cast<const float4 *>(&myo[cbl * FL_STRIDE + q * 16 + sub * 4]);
            if (valid) *reinterpret_cast<float4 *>(dst + it * 32 + q * 16 + sub * 4) = v;
        }
        wave_lds_sync();
    }
    const float offset = dpp_quad_b3(tsum) * M17_LIT_DISC_C;
    if (sub0 && valid) {
        offs[cb] = offset / (float)kBlockSamples;
        if (update_state && blk == 0) {
            st[chan].z0re = n0re; st[chan].z0im = n0im; st[chan].z1re = n1re; st[chan].z1im = n1im;
        }
    }
}
// the stand-alone launch of that tile: sixteen consecutive rows per wave (option fe_impl 4)
__global__ __launch_bounds__(64, 6)
void k_frontend_l(const uint4 *__restrict__ iq, ChanState *__restrict__ st,
                  float *__restrict__ disc_raw, float *__restrict__ offs,
                  int nblk, int total, int update_state)
{
    __shared__ __attribute__((aligned(16))) uint32_t tile[16 * FL_STRIDE];
    __shared__ __attribute__((aligned(16))) float otile[16 * FL_STRIDE];
    const int cb0 = (int)blockIdx.x * 16;
    if (cb0 >= total) return;
    frontend_lite_tile(iq, st, disc_raw, offs, nblk, update_state,
                       [&](int i, bool &valid) { valid = cb0 + i < total; return valid ? cb0 + i : total - 1; }, tile, otile, lane_id());
}

__global__ __launch_bounds__(64 * FQ_WAVES, 4)
void k_frontend_d(const uint4 *__restrict__ iq, ChanState *__restrict__ st,
                  float *__restrict__ disc_raw, float *__restrict__ offs,
                  int nblk, int total, int update_state)
{
    __shared__ __attribute__((aligned(16))) uint32_t tile[FQ_WAVES][16 * FQ_STRIDE];  // raw IQ of one chunk
    __shared__ __attribute__((aligned(16))) float otile[FQ_WAVES][16 * FQ_STRIDE];    // 64 picked outputs per row
    const int wave = (int)(threadIdx.x >> 6);
    frontend_d_tile(iq, st, disc_raw, offs, nblk, total, update_state, ((int)blockIdx.x * FQ_WAVES + wave) * 16, tile[wave], otile[wave]);
}

// ---------------------------------------------------------------------------
// k_frontend_afc: the front end of ONE block with the AFC branch taken (m17_dsp.cpp:468): dsp_nco_mixer
// (:390-408) between conversion and limiter, radio_get_afc_delta / radio_afc (radio.cpp:196-208) around it.
// AFC closes a loop through the whole chain -- the correction applied to block b comes from the DC estimate of
// block b-1 and is gated by the framer's in-frame flag after block b-1 (m17_aos / m17_los, m17_dbase.cpp:60-75,
// which is the lock flag) -- so blocks cannot be taken in parallel: m17gpu_rx_blocks runs this kernel and the
// timing / framer kernel once per block when the context has AFC on.  One wave per channel.
// The NCO phase of sample i is acc + i * delta in double (the reference adds delta i times: the two differ by
// rounding noise of ~1e-13 rad), cos / sin come from the device's double library: results agree with the
// reference to the last place except where a cosine lands on a float rounding boundary, so this path is held
// to tolerance parity (identical decoded payloads, correction trace within 1e-4), not bit parity.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(64)
void k_frontend_afc(const uint32_t *__restrict__ iq,      // [C][nblk][1920] packed (re | im << 16)
                    ChanState *__restrict__ st, float *__restrict__ disc_raw, float *__restrict__ offs,
                    int nblk, int b)
{
    __shared__ float2 z[kBlockSamples + 2];               // limited samples, the two before the block in front
    __shared__ float uh[kBlockSamples];
    const int lane = lane_id(), chan = (int)blockIdx.x;
    ChanState &cs = st[chan];
    const size_t row = (size_t)chan * nblk + b;
    const bool in_frame = uni(cs.flock) != 0;             // m17_db_in_frame(): set by m17_aos, cleared by m17_los
    const float delta = in_frame ? unif(cs.afc_delta) : 0.0f;     // radio_get_afc_delta(): dropped outside a frame
    const double acc0 = cs.afc_acc;
    if (lane == 0) { z[0] = make_float2(cs.z1re, cs.z1im); z[1] = make_float2(cs.z0re, cs.z0im); }
    for (int i = lane; i < kBlockSamples; i += 64) {
        const uint32_t w = iq[row * kBlockSamples + i];
        float re = s16_to_float((int)(short)(w & 0xFFFF)), im = s16_to_float((int)w >> 16);
        const double a = acc0 + (double)i * (double)delta;
        const float c = (float)cos(a), s = (float)sin(a);
        const float mre = (re * c) - (im * s);
        const float mim = (re * s) + (im * c);
        re = mre; im = mim;
        limit(re, im);
        z[2 + i] = make_float2(re, im);
    }
    __syncthreads();
    for (int i = lane; i < kBlockSamples; i += 64) {
        const float2 x = z[2 + i], z0 = z[1 + i], z1 = z[i];
        const float a = z0.y * (x.x - z1.x);               // dsp_arctan_disc2 (m17_dsp.cpp:194-222)
        const float bb = z0.x * (x.y - z1.y);
        uh[i] = (bb - a) * M17_LIT_DISC_C;
    }
    __syncthreads();
    for (int k = lane; k < kDiscOut; k += 64) disc_raw[row * kDiscOut + k] = uh[5 * k + 4];
    if (lane == 0) {
        float offset = 0.0f;
        double acc = acc0;
        for (int i = 0; i < kBlockSamples; ++i) { offset += uh[i]; acc += (double)delta; }   // both strictly sequential
        offset = offset / (float)kBlockSamples;
        offs[row] = offset;
        double ip;
        acc = acc / (2.0 * 3.14159265358979323846);        // :402-407
        acc = modf(acc, &ip);
        acc = acc * 2.0 * 3.14159265358979323846;
        if (acc != acc) acc = 0.0;
        cs.afc_acc = acc;
        // radio_afc(offset): float m_afc_delta, double arithmetic (radio.cpp:196-200)
        cs.afc_delta = in_frame ? (float)((double)delta - (double)offset * 0.1) : 0.0f;
        const float2 l0 = z[kBlockSamples + 1], l1 = z[kBlockSamples];
        cs.z0re = l0.x; cs.z0im = l0.y; cs.z1re = l1.x; cs.z1im = l1.y;
    }
}

// out[i] -= offset (m17_dsp.cpp:217-219) for the stand-alone front-end entry point
__global__ void k_dc_remove(float *__restrict__ disc, const float *__restrict__ offs, int total)
{
    const int i = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (i < total * kDiscOut) disc[i] = disc[i] - offs[i / kDiscOut];
}

// ---------------------------------------------------------------------------
// sync correlator (m17_rx_frame.cpp:47-81 + :22-43)
// ---------------------------------------------------------------------------
struct SyncResult { int type; int votes; float variance; };

// (the check itself is sync_check_lanes8, m17_sync_common.hip: one frame head on a wave's lane groups of eight)

// m17_unlocked_sync_check / m17_locked_sync_check (m17_rx_frame.cpp:82-103).
// `variance < 0.3` compares against a double literal: (double)v < 0.3 <=> v < 0.3f
// because 0.3f is the float nearest to and above 0.3; 0.5 is exact.
__device__ __forceinline__ bool sync_accept(const SyncResult &r, bool locked)
{
    if (r.votes > (locked ? M17_LIT_VOTES_LOCKED_MAX : M17_LIT_VOTES_UNLOCKED_MAX)) return false;
    if (r.type >= 1 && r.type <= 4) return (double)r.variance < (locked ? M17_LIT_VAR_LOCKED : M17_LIT_VAR_UNLOCKED);
    return false;
}

// ---------------------------------------------------------------------------
// frame decode: 16 lanes per frame
// ---------------------------------------------------------------------------
constexpr int DEC_FRAMES_PER_WG = 16;          // 256 threads

struct DecShared {
    float soft[kSoftBits];
    float dep[488];                 // first holds the 192 frame symbols (dead once demapped), then the de-punctured soft bits
    uint16_t dec[244];
    uint8_t bits[248];
    uint8_t bytes[32];
};

__device__ __forceinline__ float shfl16(float v, int src_in_group)
{
    const int lane = lane_id();
    const int src = (lane & 48) | (src_in_group & 15);
    return __int_as_float(__builtin_amdgcn_ds_bpermute(src << 2, __float_as_int(v)));
}
__device__ __forceinline__ int shfl16i(int v, int src_in_group)
{
    const int lane = lane_id();
    const int src = (lane & 48) | (src_in_group & 15);
    return __builtin_amdgcn_ds_bpermute(src << 2, v);
}

__device__ __forceinline__ void group_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

// m17_dsp_demap_frame (m17_dsp.cpp:82-95): sym[192] -> soft[368]
__device__ __forceinline__ void demap16(const float *sym, float *soft, int ln)
{
    float sum = 0.0f;
#pragma unroll
    for (int i = 0; i < 8; ++i) sum += fabsf(sym[i]);
    const float cor = (float)M17_LIT_DEMAP_COR_NUM / sum;      // (float)(8.0/(double)sum), see limit()
    for (int i = 8 + ln; i < kFrameSyms; i += 16) {
        const float m = sym[i] * cor;
        soft[2 * (i - 8)]     = -m;
        soft[2 * (i - 8) + 1] = (float)((double)fabsf(m) - M17_LIT_DEMAP_OFFSET);
    }
}

// m17_viterbi_decode (m17_conv.cpp:148-168) with one state per lane.
// dep[0..len) soft bits in LDS; writes bits[0..len/2) (one bit per byte) to LDS.
__device__ __forceinline__ void viterbi16(const float *dep, int len, uint16_t *dec, uint8_t *bits, int ln)
{
    const int steps = len >> 1;
    // branch metric selectors (m17_conv.cpp:93-108): metric[idx] = (idx&2 ? m1 : -m1) + (idx&1 ? m2 : -m2)
    const int ie = c_tab.bm_even[ln], io = c_tab.bm_odd[ln];
    float acm = (ln == 0) ? M17_LIT_ACM0 : 0.0f;         // :150-153
    const int grp_shift = (lane_id() & 48);
    for (int t = 0; t < steps; ++t) {
        const float m1 = dep[2 * t], m2 = dep[2 * t + 1];
        const float n1 = -m1, n2 = -m2;
        const float me = ((ie & 2) ? m1 : n1) + ((ie & 1) ? m2 : n2);
        const float mo = ((io & 2) ? m1 : n1) + ((io & 1) ? m2 : n2);
        const float pe = shfl16(acm, 2 * ln), po = shfl16(acm, 2 * ln + 1);
        const float ta = pe + me, tb = po + mo;
        const bool take_even = ta > tb;                  // strict '>' : ties pick the odd predecessor
        acm = take_even ? ta : tb;
        const unsigned long long odd = __ballot(!take_even);
        if (ln == 0) dec[t] = (uint16_t)(odd >> grp_shift);
    }
    group_sync();
    if (ln == 0) {
        int state = 0;                                   // traceback from state 0 (:160-166)
        for (int t = steps - 1; t >= 0; --t) {
            const int d = dec[t];
            state = ((state << 1) & 15) | ((d >> state) & 1);
            bits[t] = (uint8_t)(state >> 3);
        }
    }
    group_sync();
}

// m_17_golay_decode (m17_golay.cpp:103-116)
__device__ __forceinline__ int golay_decode(uint32_t word, const uint16_t *enc, const uint16_t *err, int &errs)
{
    const uint32_t data = (word >> 12) & 0xFFF, par = word & 0xFFF;
    const uint32_t syn = par ^ enc[data];
    const uint32_t e = err[syn];
    errs = (int)((e & 0xF000) >> 12);
    return (int)(data ^ (e & 0xFFF));
}

// stand-alone stage kernels -------------------------------------------------
__global__ __launch_bounds__(256)
void k_viterbi(const float *__restrict__ soft, uint8_t *__restrict__ bits, int len, int n)
{
    __shared__ DecShared sh_all[DEC_FRAMES_PER_WG];
    const int g = (int)(threadIdx.x >> 4), ln = (int)(threadIdx.x & 15);
    const int item = (int)blockIdx.x * DEC_FRAMES_PER_WG + g;
    const bool active = item < n;
    const int it = active ? item : n - 1;
    DecShared &sh = sh_all[g];
    for (int q = ln; q < len; q += 16) sh.dep[q] = soft[(size_t)it * len + q];
    group_sync();
    viterbi16(sh.dep, len, sh.dec, sh.bits, ln);
    if (active)
        for (int q = ln; q < (len >> 1); q += 16) bits[(size_t)it * (len >> 1) + q] = sh.bits[q];
}

__global__ __launch_bounds__(256)
void k_demap(const float *__restrict__ sym, float *__restrict__ soft, int n)
{
    __shared__ DecShared sh_all[DEC_FRAMES_PER_WG];
    const int g = (int)(threadIdx.x >> 4), ln = (int)(threadIdx.x & 15);
    const int item = (int)blockIdx.x * DEC_FRAMES_PER_WG + g;
    const bool active = item < n;
    const int it = active ? item : n - 1;
    DecShared &sh = sh_all[g];
    for (int q = ln; q < kFrameSyms; q += 16) sh.dep[q] = sym[(size_t)it * kFrameSyms + q];
    group_sync();
    demap16(sh.dep, sh.soft, ln);
    group_sync();
    if (active)
        for (int q = ln; q < kSoftBits; q += 16) soft[(size_t)it * kSoftBits + q] = sh.soft[q];
}

__global__ void k_golay(const uint32_t *__restrict__ words, uint16_t *__restrict__ out, int n,
                        const uint16_t *__restrict__ genc, const uint16_t *__restrict__ gerr)
{
    const int i = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (i >= n) return;
    int e;
    const int d = golay_decode(words[i], genc, gerr, e);
    out[i] = (uint16_t)(d | (e << 12));
}

__global__ void k_reset(ChanState *st, int C)
{
    const int i = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    const int words = (int)(sizeof(ChanState) / 4);
    if (i >= C * words) return;
    uint32_t *p = reinterpret_cast<uint32_t *>(st);
    const int w = i % words;
    uint32_t v = 0;
    if (w == (int)(offsetof(ChanState, clk) / 4)) v = M17_LIT_CLK_INIT;          // m17_rx_sync.cpp:123
    if (w == (int)(offsetof(ChanState, index) / 4)) v = M17_LIT_INDEX_INIT;      // :126
    p[i] = v;
}

} // namespace m17dev
