// m17_decode_quad.hip -- frame decode with FOUR lanes per frame, four trellis states per
// lane (included after m17_sync_*.hip):
//
//   k_worklist     one thread per record slot: the decodable frames of all channels are
//                  appended to one list per frame type (wave-aggregated atomics, ~C*rec_cap/64
//                  of them -- not one returning atomic per frame inside the sequential framer).
//   k_decode_quad  one wave per 16 frames.  The add-compare-select butterfly of the K=5 code
//                  (m17_conv.cpp:73-113) maps onto a DPP quad: lane j holds states 4j..4j+3,
//                  the predecessors {2v, 2v+1} mod 16 of its four new states sit in quad
//                  lanes (2j) mod 4 and (2j+1) mod 4, so the eight old metrics arrive as
//                  quad_perm operands of the eight adds -- no LDS crossbar (ds_bpermute) on
//                  the per-step critical path, which is what bounds the 16-lane form
//                  (viterbi16 in m17_kernels.hip).  Soft bits are produced 16 trellis steps
//                  at a time straight from the frame symbols (demap . de-randomise .
//                  de-interleave . de-puncture as one table), so a frame needs 432 B of LDS
//                  (stream; 624 B any type) and 16 frames fit a wave.
//   (k_book_chan, the in-order per-channel bookkeeping, follows in m17_book.hip.)
//
// Branch metrics: metric[idx] = (idx&2 ? m1 : -m1) + (idx&1 ? m2 : -m2) (m17_conv.cpp:88-91).
// Both generators tap the newest and the oldest register bit, so the two predecessors of a
// state expect complementary dibits: metric[odd] = -metric[even] up to the sign of a zero
// (IEEE negation commutes with round-to-nearest), and a zero's sign never reaches a
// comparison: x + (+-0) == x for x != 0, and +0 == -0 under '>'.  Decisions are identical.
#pragma clang fp contract(off)

namespace m17dev {

constexpr int DQ_FRAMES = 16;                  // frames per wave
constexpr int DQ_CHUNK  = 16;                  // trellis steps per soft-bit chunk
constexpr int DQ_NPER   = DQ_CHUNK / 2;        // soft bits per lane and chunk
constexpr int DQ_RING   = 2 * DQ_CHUNK;
constexpr int DQ_DECW   = 31;                  // ceil(244 / 8) decision rows: any frame type
constexpr int DQ_DECW_STREAM = 19;             // ceil(148 / 8): stream frames

// Per-frame LDS.  The record payload bytes are assembled after the forward pass, in the ring's place.
// DECW = 31: 624 B = 4 x 39 dwords, DECW = 19: 432 B = 4 x 27 dwords -- either way the 16 frames of a wave
// start 4 banks apart modulo 64, so the same field of all frames tiles the banks.
template <int DECW, int NSYM = 0>
struct alignas(16) QuadFrameT {
    uint32_t dec[DECW][4];                     // decision nibbles: byte [t/2][quad lane], two steps per byte
    union {
        float   ring[DQ_RING];                 // soft bits of the current chunk, (m1, m2) pairs
        uint8_t bytes[32];                     // record payload
    };
    float sym[NSYM ? NSYM : 4];                // NSYM = 192 (slot_impl 1): the frame's symbols, staged here from a plain 768-byte slot
    static constexpr int kDecw = DECW;
    static constexpr int kNsym = NSYM;
};
static_assert(sizeof(QuadFrameT<DQ_DECW>) == 16 * DQ_DECW + 8 * DQ_CHUNK + 16 && sizeof(QuadFrameT<DQ_DECW_STREAM>) == 16 * DQ_DECW_STREAM + 8 * DQ_CHUNK + 16, "QuadFrame layout");

typedef float dq_f4 __attribute__((ext_vector_type(4)));

template <int CTRL> __device__ __forceinline__ float dppf(float v)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, true));
}
template <int CTRL> __device__ __forceinline__ int dppi(int v)
{
    return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xF, 0xF, true);
}

// The gather tables (DevTables.gather / .lich: source soft-bit index, 0x4000 = negated, -1 = erasure) are re-coded
// once per workgroup into LDS entries the inner loops can use without decoding:
//   bits 0..9   byte offset of the source symbol in the frame (4 * (8 + s / 2))
//   bit  10     the soft bit is the dibit's second one (|m| - 0.6666), else the first (-m)   m17_dsp.cpp:35-42
//   bit  12     valid (0 = erasure, m17_puncture.cpp:54)
//   bit  31     negate (m17_de_correlate_1)
__host__ __device__ inline uint32_t dq_entry(int g)
{
    if (g < 0) return 32u;
    const uint32_t s = (uint32_t)g & 0x3FFu;
    return (4u * (8u + (s >> 1))) | ((s & 1u) << 10) | (1u << 12) | (((uint32_t)g & 0x4000u) ? 0x80000000u : 0u);
}
__device__ __forceinline__ float dq_symbol(const float *gs, uint32_t e)
{
    return *reinterpret_cast<const float *>(reinterpret_cast<const char *>(gs) + (e & 0x3FFu));
}
// ncor = -cor: symval * ncor == -(symval * cor) bit for bit, and |.| of the two is the same
__device__ __forceinline__ float dq_soft(uint32_t e, float symval, float ncor)
{
    const float nm = symval * ncor;
    const float odd = (float)((double)fabsf(nm) - M17_LIT_DEMAP_OFFSET);
    const uint32_t pm = (uint32_t)__builtin_amdgcn_sbfe((int)e, 10, 1);              // all ones: second bit of the dibit
    uint32_t v = (pm & __float_as_uint(odd)) | (~pm & __float_as_uint(nm));
    v ^= e & 0x80000000u;
    v &= (uint32_t)__builtin_amdgcn_sbfe((int)e, 12, 1);                             // erasure: +0.0
    return __uint_as_float(v);
}
// LICH entries: the hard decision `soft >= 0` (hard_decode_24_bits, m17_golay.cpp) as one compare y >= T:
//   first bit:   -m >= 0 (negated: m >= 0)                        y = -m (m),      T = 0
//   second bit:  (float)((double)|m| - 0.6666) >= 0  <=>  |m| >= RU(0.6666) in fp32 (the difference is an exact
//                non-zero double of magnitude > 1e-9, so the conversion keeps its sign); negated: |m| <= RD(0.6666)
//                                                                  y = |m| (-|m|),  T = RU (-RD)
__host__ __device__ inline DqLich dq_lich_entry(int g)       // e: offset, bit 10 and bit 31 as above
{
    // 0.6666 lies strictly between the floats 0x3F2AA64C (0.66659999) and 0x3F2AA64D (0.66660005)
    static_assert((double)__builtin_bit_cast(float, 0x3F2AA64Cu) < M17_LIT_DEMAP_OFFSET && M17_LIT_DEMAP_OFFSET < (double)__builtin_bit_cast(float, 0x3F2AA64Du),
                  "the LICH thresholds are the two floats around the demapper's offset");
    DqLich L;
    L.e = dq_entry(g);
    const bool second = (L.e >> 10) & 1u, neg = (L.e >> 31) != 0u;
    L.T = !second ? 0.0f : (neg ? -__builtin_bit_cast(float, 0x3F2AA64Cu) : __builtin_bit_cast(float, 0x3F2AA64Du));
    return L;
}

// 1,024 threads per workgroup and one atomic per (workgroup, type): all waves appending to
// the same counter serialise in L2 (measured 40 us for 1,660 wave-level atomics).
__global__ __launch_bounds__(1024)
void k_worklist(const m17gpu_rec_dev *__restrict__ recs, int rec_cap, const int32_t *__restrict__ counts, int C,
                int32_t *__restrict__ work, int32_t *__restrict__ nwork, int cap)
{
    __shared__ int wcount[3][16];                 // [type][wave]
    __shared__ int wbase[3][16];
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const int lane = lane_id(), wave = (int)(threadIdx.x >> 6);
    const int chan = (int)(i / rec_cap), r = (int)(i - (long long)chan * rec_cap);
    int ty = 0;
    if (chan < C && r < min(counts[chan], rec_cap)) {
        const uint32_t *w = reinterpret_cast<const uint32_t *>(&recs[i]);
        const uint32_t w0 = w[0], w1 = w[1];
        if ((w1 & M17_F_PARSED) && (w0 & 0xFF) >= 1 && (w0 & 0xFF) <= 3) ty = (int)(w0 & 0xFF);
    }
    const unsigned long long below = (1ull << lane) - 1ull;
    unsigned long long m[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        m[k] = __ballot(ty == k + 1);
        if (lane == 0) wcount[k][wave] = (int)__popcll(m[k]);
    }
    __syncthreads();
    if (threadIdx.x < 3) {
        const int k = (int)threadIdx.x;
        int tot = 0;
        for (int w = 0; w < 16; ++w) { wbase[k][w] = tot; tot += wcount[k][w]; }
        const int base = tot ? atomicAdd(&nwork[k], tot) : 0;
        for (int w = 0; w < 16; ++w) wbase[k][w] += base;
    }
    __syncthreads();
    // (idx < cap always holds when the counters were zero on entry -- a list has at most one entry per record slot; the
    //  bound keeps a call that found them non-zero, e.g. behind a call that failed between this kernel and the
    //  bookkeeping kernel that resets them, from writing past its list: m17gpu_rx_blocks also zeroes them in that case)
    if (ty) {
        const int idx = wbase[ty - 1][wave] + (int)__popcll(m[ty - 1] & below);
        if (idx < cap) work[(size_t)(ty - 1) * cap + idx] = (int32_t)i;
    }
}

// One frame type per pass: `type` is wave-uniform (a template constant for the stream kernel), so trellis length,
// table row and every loop bound live in SGPRs.  Quads whose frame has another type (plain-batch mode only; the work
// lists are per type) ride along on their own symbols and write nothing.
// The frame's 192 symbols stay in global memory (gs; written by the framer just before): every soft bit is one gather
// through the table, the loads of the NEXT 16 trellis steps fly while the current 16 are processed.  Everything a
// frame needs first -- the 8 sync symbols, the LICH symbols, the first chunk -- is requested in one go at the top, and
// the two dependent Golay table reads ride under the forward pass, so a frame pays one memory round trip, not five.
// Fences inside are wave-local LDS fences: a workgroup-scope fence would also drain the loads in flight.
template <int DECW, int TYPE_CT, int NSYM = 0>
__device__ __forceinline__ void decode_quad_pass(QuadFrameT<DECW, NSYM> &F, const float *__restrict__ gs_in, const uint32_t *gt,
                                                 const DqLich *lich, int type_rt, int j, bool writeback, uint32_t r0_keep,
                                                 m17gpu_rec_dev *rec, const v2f (&C1)[2], const v2f (&C2)[2],
                                                 const uint16_t *genc, const uint16_t *gerr,
                                                 unsigned long long *acc_, unsigned long long &last_,
                                                 const float *__restrict__ gs_next, dq_f4 (&pre)[NSYM ? NSYM / 16 : 1])
{
    const int type = TYPE_CT ? TYPE_CT : type_rt;
    const int steps = (type == 1) ? 244 : (type == 2 ? 148 : 210);       // DevTables.glen / 2
    const int nbits = (type == 1) ? 240 : (type == 2 ? 144 : 208);
    const int boff = (type == 2) ? 6 : 0;
    // soft bits of steps c0 .. c0+31 (row padded with erasures).  Plain 192-symbol frames (TYPE_CT == 0): quad lane j
    // makes ring[j], ring[j+4], ..., each from a gather through the table.  Regrouped stream slots (TYPE_CT == 2,
    // m17_dev.h kSlotFloats): lane j makes ring[16j .. 16j+15] from 64 contiguous bytes of the slot.
    // (the table entry is read again at commit time: holding 16 of them across the butterflies costs registers the
    //  16-waves-per-CU budget of 128 does not have)
    constexpr bool REGROUPED = (TYPE_CT == 2) && NSYM == 0;
    // slot_impl 1 (round 5, NSYM == 192): the slot holds the frame's 192 symbols as the framer has them (768 B instead of the
    // regrouped 1,600); the quad copies them into LDS in one contiguous read and every gather below goes there
    // The loads are one TASK ahead (pre[]: this frame's symbols, requested during the previous task's forward pass by the
    // caller's first request or by the request below): at the two waves per SIMD the staging leaves, nothing else would
    // cover the round trip to memory at the head of every task.
    const float *gs = gs_in;
    if constexpr (NSYM != 0) {
#pragma unroll
        for (int r = 0; r < NSYM / 16; ++r) reinterpret_cast<dq_f4 *>(F.sym)[(NSYM / 16) * j + r] = pre[r];
        wave_fence();
        gs = F.sym;
    }
    float raw[2 * DQ_CHUNK / 4];
    float ncor;
    auto fetch_chunk = [&](int c0) {
        if constexpr (REGROUPED) {
            const float4 *g4 = reinterpret_cast<const float4 *>(gs + 104 + 2 * c0 + DQ_NPER * j);
#pragma unroll
            for (int r = 0; r < DQ_NPER / 4; ++r) { const float4 v = g4[r]; raw[4 * r] = v.x; raw[4 * r + 1] = v.y; raw[4 * r + 2] = v.z; raw[4 * r + 3] = v.w; }
        } else {
#pragma unroll
            for (int r = 0; r < 2 * DQ_CHUNK / 4; ++r) raw[r] = dq_symbol(gs, gt[(2 * c0 + j + 4 * r) & 511]);
        }
    };
    auto commit_chunk = [&](int c0) {
        if constexpr (REGROUPED) {
            float v[DQ_NPER];
#pragma unroll
            for (int r = 0; r < DQ_NPER; ++r) v[r] = dq_soft(gt[(2 * c0 + DQ_NPER * j + r) & 511], raw[r], ncor);
#pragma unroll
            for (int r = 0; r < DQ_NPER / 4; ++r)
                reinterpret_cast<float4 *>(F.ring)[(DQ_NPER / 4) * j + r] = make_float4(v[4 * r], v[4 * r + 1], v[4 * r + 2], v[4 * r + 3]);
        } else {
#pragma unroll
            for (int r = 0; r < 2 * DQ_CHUNK / 4; ++r) F.ring[j + 4 * r] = dq_soft(gt[(2 * c0 + j + 4 * r) & 511], raw[r], ncor);
        }
    };

    // ---- requests: sync symbols, first chunk, LICH symbols
    float s8[8];
    if constexpr (REGROUPED) {
        const float4 a = reinterpret_cast<const float4 *>(gs)[0], b = reinterpret_cast<const float4 *>(gs)[1];
        s8[0] = a.x; s8[1] = a.y; s8[2] = a.z; s8[3] = a.w; s8[4] = b.x; s8[5] = b.y; s8[6] = b.z; s8[7] = b.w;
    } else {
#pragma unroll
        for (int i = 0; i < 8; ++i) s8[i] = gs[i];
    }
    fetch_chunk(0);
    float lraw[24];
    if (type == 2) {
        if constexpr (REGROUPED) {
            const float4 *g4 = reinterpret_cast<const float4 *>(gs + 8 + 24 * j);
#pragma unroll
            for (int r = 0; r < 6; ++r) { const float4 v = g4[r]; lraw[4 * r] = v.x; lraw[4 * r + 1] = v.y; lraw[4 * r + 2] = v.z; lraw[4 * r + 3] = v.w; }
        } else {
#pragma unroll
            for (int k = 0; k < 24; ++k) lraw[k] = dq_symbol(gs, lich[j * 24 + k].e);
        }
    }
    // m17_dsp_demap_frame (m17_dsp.cpp:82-95): amplitude reference from the 8 sync symbols
    {
        float sum = 0.0f;
#pragma unroll
        for (int i = 0; i < 8; ++i) sum += fabsf(s8[i]);
        ncor = -((float)M17_LIT_DEMAP_COR_NUM / sum);      // cor = (float)(8.0/(double)sum), see limit()
    }
    // ---- LICH (m17_rx_parse.cpp:118-135): quad lane j decodes Golay word j
    uint32_t gdata = 0, gpar = 0, genc_v = 0, gerr_v = 0;
    if (type == 2) {
        uint32_t word = 0;
#pragma unroll
        for (int k = 0; k < 24; ++k) {                                   // hard_decode_24_bits: soft >= 0, as y >= T
            const DqLich L = lich[j * 24 + k];
            const float nm = lraw[k] * ncor;
            const uint32_t am = (L.e << 21) & 0x80000000u;               // second bit of the dibit: |.|
            const float y = __uint_as_float((__float_as_uint(nm) & ~am) ^ (L.e & 0x80000000u));
            word = (word << 1) | (y >= L.T ? 1u : 0u);
        }
        gdata = (word >> 12) & 0xFFFu; gpar = word & 0xFFFu;             // m_17_golay_decode (m17_golay.cpp:103-116)
        genc_v = genc[gdata];                                            // consumed after the first chunk
    }
    STAMP(1);

    // ---- forward pass (m17_viterbi_decode, m17_conv.cpp:148-158)
    // path metrics of states 4j .. 4j+3: two register sets, a trellis step reads one and writes the other
    float A0 = (j == 0) ? M17_LIT_ACM0 : 0.0f, A1 = 0.0f, A2 = 0.0f, A3 = 0.0f, B0, B1, B2, B3;     // :150-153
    uint32_t dw = 0;
    // One add-compare-select (BF, m17_conv.cpp:19): ta = acm[even predecessor] + M, tb = acm[odd predecessor] - M with
    // the predecessors' metrics arriving as DPP operands of the adds (old states 2v, 2v+1 mod 16 sit in quad lane
    // (2j) mod 4 for v = 4j, 4j+1 and (2j+1) mod 4 above); strict '>' keeps the even one, ties and NaN the odd one;
    // the decision enters the nibble through the carry of dw + dw.  Written out so that every step is exactly these
    // five instructions per state (the compiler's own choice was ~9, with packed adds fed by eight v_mov_dpp).
    // DPP reads need their source two instructions old: every source here was written at least five earlier.
    // Branch metrics of the lane's four states (round 5): the code is linear, so the expected dibit of state 4j+i (even
    // predecessor) is e(i) xor g(j) -- bm_even = ((j1 ^ i0) << 1) | (j1 ^ j0 ^ i1) -- and with m1' = sg1 m1, m2' = sg2 m2
    // (sg = the lane's signs, those of its state 4j+3) the four metrics are -S, D, -D, S for i = 0..3, S = m1' + m2',
    // D = m1' - m2': one multiply and one packed fma per step instead of two and two, the negations ride on the choice
    // of add / subtract in the butterfly.  Same roundings: sg1 m1 is exact, fma(+-sg2, m2, sg1 m1) rounds the same exact
    // sum the reference's (+-m1) + (+-m2) rounds (m17_conv.cpp:88-91), and -RN(x) = RN(-x).
    // The step's four add-compare-selects as ONE statement, phase by phase (round 5): all eight DPP adds, then the four
    // compares -- each into an SGPR pair of its own, not VCC -- then the selects and the carries.  Written one butterfly
    // after the other (add, sub, compare, select, carry, all through VCC), every instruction waited for the one in front
    // of it; here four independent chains overlap.  Same instructions, same operands.
#define DQ_STEP(m1, m2, a0, a1, a2, a3, n0, n1, n2, n3) do {                                              \
        const float t1_ = sg1 * (m1);                                                                   \
        const v2f SD = __builtin_elementwise_fma(K2, (v2f){m2, m2}, (v2f){t1_, t1_});   /* (S, D) */    \
        float tb3_, tb2_, tb1_, tb0_;                                                                   \
        unsigned long long k3_, k2_, k1_, k0_;                                                          \
        asm volatile("v_add_f32_dpp %0, %15, %17 quad_perm:[1,3,1,3] row_mask:0xf bank_mask:0xf\n\t"     /* state 4j+3: +S */ \
                     "v_sub_f32_dpp %4, %16, %17 quad_perm:[1,3,1,3] row_mask:0xf bank_mask:0xf\n\t"                        \
                     "v_sub_f32_dpp %1, %13, %18 quad_perm:[1,3,1,3] row_mask:0xf bank_mask:0xf\n\t"     /* state 4j+2: -D */ \
                     "v_add_f32_dpp %5, %14, %18 quad_perm:[1,3,1,3] row_mask:0xf bank_mask:0xf\n\t"                        \
                     "v_add_f32_dpp %2, %15, %18 quad_perm:[0,2,0,2] row_mask:0xf bank_mask:0xf\n\t"     /* state 4j+1: +D */ \
                     "v_sub_f32_dpp %6, %16, %18 quad_perm:[0,2,0,2] row_mask:0xf bank_mask:0xf\n\t"                        \
                     "v_sub_f32_dpp %3, %13, %17 quad_perm:[0,2,0,2] row_mask:0xf bank_mask:0xf\n\t"     /* state 4j:   -S */ \
                     "v_add_f32_dpp %7, %14, %17 quad_perm:[0,2,0,2] row_mask:0xf bank_mask:0xf\n\t"                        \
                     "v_cmp_ngt_f32_e64 %8, %0, %4\n\t"                                                                     \
                     "v_cmp_ngt_f32_e64 %9, %1, %5\n\t"                                                                     \
                     "v_cmp_ngt_f32_e64 %10, %2, %6\n\t"                                                                    \
                     "v_cmp_ngt_f32_e64 %11, %3, %7\n\t"                                                                    \
                     "v_cndmask_b32_e64 %0, %0, %4, %8\n\t"                                                                 \
                     "v_addc_co_u32_e64 %12, %8, %12, %12, %8\n\t"      /* descending: nibble bit i = decision of state 4j+i */ \
                     "v_cndmask_b32_e64 %1, %1, %5, %9\n\t"                                                                 \
                     "v_addc_co_u32_e64 %12, %9, %12, %12, %9\n\t"                                                          \
                     "v_cndmask_b32_e64 %2, %2, %6, %10\n\t"                                                                \
                     "v_addc_co_u32_e64 %12, %10, %12, %12, %10\n\t"                                                        \
                     "v_cndmask_b32_e64 %3, %3, %7, %11\n\t"                                                                \
                     "v_addc_co_u32_e64 %12, %11, %12, %12, %11"                                                              \
                     : "=&v"(n3), "=&v"(n2), "=&v"(n1), "=&v"(n0), "=&v"(tb3_), "=&v"(tb2_), "=&v"(tb1_), "=&v"(tb0_),       \
                       "=&s"(k3_), "=&s"(k2_), "=&s"(k1_), "=&s"(k0_), "+v"(dw)                                              \
                     : "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(SD.x), "v"(SD.y)); } while (0)
    const float sg1 = C1[1].y;
    const v2f K2 = {C2[1].y, -C2[1].y};
    commit_chunk(0);
    wave_fence();
    asm volatile("s_nop 1");
    for (int c0 = 0; c0 < steps; c0 += DQ_CHUNK) {
        if (c0 + DQ_CHUNK < steps) fetch_chunk(c0 + DQ_CHUNK);       // in flight during this chunk's butterflies
        STAMP(2);
        const int tend2 = min(DQ_CHUNK, steps - c0) >> 1;     // steps is even
        const float4 *ring4 = reinterpret_cast<const float4 *>(F.ring);
        float4 cur = ring4[0];
        for (int t2 = 0; t2 < tend2; ++t2) {
            const float4 nxt = ring4[min(t2 + 1, DQ_CHUNK / 2 - 1)];            // one pair ahead: no LDS wait per step
            DQ_STEP(cur.x, cur.y, A0, A1, A2, A3, B0, B1, B2, B3);
            DQ_STEP(cur.z, cur.w, B0, B1, B2, B3, A0, A1, A2, A3);
            // two steps per byte: even step in the high nibble
            reinterpret_cast<uint8_t *>(F.dec)[4 * ((c0 >> 1) + t2) + j] = (uint8_t)dw;
            cur = nxt;
        }
        wave_fence();
        if (type == 2 && c0 == 0) gerr_v = gerr[gpar ^ genc_v];        // second Golay table: consumed after the traceback
        if constexpr (NSYM != 0) {
            // the next task's symbols: requested behind the Golay reads (loads return in order: a wait for an earlier
            // request never waits for these), consumed at the head of the next pass
            if (c0 == 0 && gs_next) {
                const dq_f4 *g4 = reinterpret_cast<const dq_f4 *>(gs_next) + (NSYM / 16) * j;
#pragma unroll
                for (int r = 0; r < NSYM / 16; ++r) pre[r] = g4[r];
            }
        }
        if (c0 + DQ_CHUNK < steps) { commit_chunk(c0 + DQ_CHUNK); wave_fence(); }
        STAMP(3);
    }
    // the ring has been read for the last time: its place becomes the record payload
    reinterpret_cast<uint2 *>(F.bytes)[j] = make_uint2(0u, 0u);
    wave_fence();

    // ---- traceback from state 0 (:160-166) and pack_1_to_8(&bits[1], ...) in one go.
    // dword t/2 of dec holds the four lanes' bytes: decision of state s at step t is bit
    // 8 (s >> 2) + (s & 3) + (t even ? 4 : 0).  The walk keeps its decisions in a shift register: the state after
    // step t is its low nibble (d[t+3] d[t+2] d[t+1] d[t]), so bits[t] = state >> 3 = d[t+3] and output bit
    // u = t - 1 is d[u+4]: after step t = 4 + 32 w the register IS output word w, bit k = output bit 32 w + k.
    {
        uint32_t hist = 0;
        const uint4 *dq = reinterpret_cast<const uint4 *>(F.dec);
        for (int g = (steps - 1) >> 3; g >= 0; --g) {
            const uint4 W = dq[g];                         // steps 8g .. 8g+7, one LDS read
#pragma unroll
            for (int tt = 7; tt >= 0; --tt) {
                if (8 * g + tt < steps) {                  // scalar: only the top group is partial
                    const uint32_t wv = (tt >> 1) == 0 ? W.x : ((tt >> 1) == 1 ? W.y : ((tt >> 1) == 2 ? W.z : W.w));
                    const uint32_t pos = (hist & 12u) + (hist & 15u) + ((tt & 1) ? 0u : 4u);
                    hist = (hist << 1) | ((wv >> pos) & 1u);
                }
                if (tt == 4 && (g & 3) == 0 && 8 * g < nbits) {
                    // pack_1_to_8: output bit u sits at bit 7 - (u & 7) of byte u / 8
                    const uint32_t word = __builtin_bswap32(__builtin_bitreverse32(hist));
                    if (j == 0) {
                        uint16_t *o = reinterpret_cast<uint16_t *>(F.bytes + boff + 4 * (g >> 2));
                        o[0] = (uint16_t)word; o[1] = (uint16_t)(word >> 16);
                    }
                }
            }
        }
    }
    uint32_t gerrs = 0;
    if (type == 2) {
        const int e = (int)((gerr_v & 0xF000u) >> 12);
        const int w = (int)(gdata ^ (gerr_v & 0xFFFu));
        const int w0 = dppi<0x00>(w), w1 = dppi<0x55>(w), w2 = dppi<0xAA>(w), w3 = dppi<0xFF>(w);
        gerrs = (uint32_t)(dppi<0x00>(e) + dppi<0x55>(e) + dppi<0xAA>(e) + dppi<0xFF>(e));
        if (j == 0) {                                                    // pack_12_to_8_x4x6
            const uint32_t a = ((uint32_t)w0 << 12) | (uint32_t)w1, b = ((uint32_t)w2 << 12) | (uint32_t)w3;
            F.bytes[0] = (uint8_t)(a >> 16); F.bytes[1] = (uint8_t)(a >> 8); F.bytes[2] = (uint8_t)a;
            F.bytes[3] = (uint8_t)(b >> 16); F.bytes[4] = (uint8_t)(b >> 8); F.bytes[5] = (uint8_t)b;
        }
    }
    wave_fence();
    STAMP(4);
    uint32_t fn = 0;
    if (type == 2) fn = ((uint32_t)F.bytes[6] << 8) | F.bytes[7];                     // pack_8_to_16
    if (type == 3) fn = ((uint32_t)(F.bytes[25] >> 7) << 8) | ((F.bytes[25] >> 2) & 0x1F);
    if (writeback) {
        uint32_t *r = reinterpret_cast<uint32_t *>(rec);
        const uint32_t *bw = reinterpret_cast<const uint32_t *>(F.bytes);
        r[5 + j] = bw[j]; r[9 + j] = bw[4 + j];
        if (j == 0) {
            r[0] = (r0_keep & 0xFF00FFFFu) | ((gerrs & 0xFF) << 16);
            // Stream frames of a call (work-list mode): the flags the bookkeeping kernel will find true for nearly every
            // frame of a transmission -- delivered, and LICH accepted for the counter values that carry a chunk -- are set
            // here, in the word this lane rewrites anyway: k_book_chan clears what does not hold and stores a record's flag
            // word only when it differs, which then is rare (its 4-byte stores at a 64-byte stride were 10 us of its 44).
            uint32_t opt = 0u;
            if constexpr (TYPE_CT == 2) opt = M17_F_DELIVERED | (((uint32_t)F.bytes[5] >> 5) < 6u ? M17_F_LICH_OK : 0u);
            r[1] = (r[1] & 0x0000FFFFu) | opt | (fn << 16);
        }
    }
    wave_fence();
}

// Workgroup-shared storage of the decoder for WAVES waves, each on its own 16 frames.
// ONLY == 2: stream frames only: per-frame LDS sized for 148 trellis steps, one shared table row.
// ONLY == 0: any frame type, a table row per wave.
template <int ONLY, int WAVES, int NSYM = 0>
struct alignas(16) DqShared {
    QuadFrameT<(ONLY == 2) ? DQ_DECW_STREAM : DQ_DECW, NSYM> fr[WAVES][DQ_FRAMES];   // 7 KB (stream; 19 KB with symbol staging) / 10 KB per wave
    uint32_t gt_rows[ONLY ? 1 : WAVES][512];                                   // DevTables.gather row of the current type, re-coded
    DqLich   lich_row[96];
};

// The decoder as workgroup number wg of n_wg workgroups of WAVES waves (threads beyond 64 * WAVES must not enter).
// work != nullptr: lists per type (work[3][cap], nwork[3]; ONLY == 2 takes the stream list, ONLY == 0 all of them or,
// with skip_stream, all but the stream list); else (ONLY == 0) plain batch: frame i of n_plain, type from types[i],
// record i.
template <int ONLY, int WAVES, int NSYM = 0>
__device__ __forceinline__ void decode_quad_body(DqShared<ONLY, WAVES, NSYM> &sh, int wg, int n_wg,
                   const float *__restrict__ fsym, const int32_t *__restrict__ work,
                   const int32_t *__restrict__ nwork, int cap,
                   const uint8_t *__restrict__ types, int n_plain,
                   m17gpu_rec_dev *__restrict__ recs,
                   const uint16_t *__restrict__ genc, const uint16_t *__restrict__ gerr, int skip_stream,
                   int slot_floats, bool spectator = false)
{
    using Frame = QuadFrameT<(ONLY == 2) ? DQ_DECW_STREAM : DQ_DECW, NSYM>;
    const int lane = lane_id(), q = lane >> 2, j = lane & 3, wave = (int)(threadIdx.x >> 6);
    Frame &F = sh.fr[wave][q];
    uint32_t *gt_row = sh.gt_rows[ONLY ? 0 : wave];
    DqLich *lich_row = sh.lich_row;

    // per-lane constants of the butterfly: state v = 4j+i, even predecessor's metric index
    // metric[idx] = (idx & 2 ? m1 : -m1) + (idx & 1 ? m2 : -m2)  (m17_conv.cpp:88-91) = c1 * m1 + c2 * m2 with c = +-1:
    // states (4j, 4j+1) in C[0], (4j+2, 4j+3) in C[1]
    v2f C1[2], C2[2];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int idx = c_tab.bm_even[4 * j + i];
        const float c1 = (idx & 2) ? 1.0f : -1.0f, c2 = (idx & 1) ? 1.0f : -1.0f;
        if (i & 1) { C1[i >> 1].y = c1; C2[i >> 1].y = c2; } else { C1[i >> 1].x = c1; C2[i >> 1].x = c2; }
    }

    // nothing to do for this workgroup (no frame of these types this call, or more workgroups than tasks): leave
    // before the tables are filled -- the any-type kernel normally finds its lists empty
    {
        int m1 = 0, m2 = 0, m3 = 0;
        if (work) { m1 = min(nwork[0], cap); m2 = min(nwork[1], cap); m3 = min(nwork[2], cap); }
        if (ONLY == 0 && skip_stream) m2 = 0;
        const int tasks = ONLY ? (m2 + DQ_FRAMES - 1) / DQ_FRAMES
                               : (work ? (m1 + DQ_FRAMES - 1) / DQ_FRAMES + (m2 + DQ_FRAMES - 1) / DQ_FRAMES + (m3 + DQ_FRAMES - 1) / DQ_FRAMES
                                       : (n_plain + DQ_FRAMES - 1) / DQ_FRAMES);
        if (wg * WAVES >= tasks) return;                                    // the whole workgroup leaves together
    }
    // waves of the workgroup beyond this role's WAVES (k_decode_lists' any-type role uses two of four): they take
    // part in the workgroup's one barrier and leave behind it -- a barrier that part of a workgroup never reaches is
    // undefined in the HIP model, whatever the hardware does with ended waves
    if (spectator) { __syncthreads(); return; }
    for (int i = (int)threadIdx.x; i < 96; i += 64 * WAVES) lich_row[i] = c_tab.lich_q[i];
    int row_type = 0;
    if (ONLY) {
        for (int i = (int)threadIdx.x; i < 512; i += 64 * WAVES) gt_row[i] = dq_entry((i < 488) ? (int)c_tab.gather[ONLY][i] : -1);
        row_type = ONLY;
    }
    __syncthreads();                                                        // the only workgroup barrier: tables are read-only from here

    int n1 = 0, n2 = 0, n3 = 0;
    if (work) { n1 = min(nwork[0], cap); n2 = min(nwork[1], cap); n3 = min(nwork[2], cap); }     // a list never holds more than its capacity (k_worklist)
    if (ONLY == 0 && skip_stream) n2 = 0;
    const int t2 = (n2 + DQ_FRAMES - 1) / DQ_FRAMES, t1 = (n1 + DQ_FRAMES - 1) / DQ_FRAMES,
              t3 = (n3 + DQ_FRAMES - 1) / DQ_FRAMES;
    const int ntask = uni(ONLY ? t2 : (work ? (t2 + t1 + t3) : (n_plain + DQ_FRAMES - 1) / DQ_FRAMES));
    const int stride = n_wg * WAVES;

    // the frame a quad works on in task `task` (work-list mode): list, index and whether it exists
    auto pick = [&](int task, int &qtype, bool &active) -> size_t {
        int seg, base, n;
        if (ONLY || task < t2) { seg = 1; base = task * DQ_FRAMES; n = n2; }               // stream frames first
        else if (task < t2 + t1) { seg = 0; base = (task - t2) * DQ_FRAMES; n = n1; }
        else { seg = 2; base = (task - t2 - t1) * DQ_FRAMES; n = n3; }
        active = base + q < n;
        qtype = seg + 1;
        return (size_t)seg * cap + (active ? base + q : n - 1);
    };

    unsigned long long acc_[12] = {0,0,0,0,0,0,0,0,0,0,0,0}, last_ = 0;
#ifdef M17_STAMPS
    last_ = __builtin_amdgcn_s_memtime();
    __builtin_amdgcn_s_waitcnt(0xC07F);
#endif
    int task = wg * WAVES + wave;
    int slot_next = 0;
    if (work && task < ntask) { int qt; bool ac; slot_next = work[pick(task, qt, ac)]; }
    dq_f4 pre[NSYM ? NSYM / 16 : 1];                                        // plain slots: the symbols of the task to come
    if constexpr (NSYM != 0) {
        if (task < ntask) {
            const dq_f4 *g4 = reinterpret_cast<const dq_f4 *>(fsym + (size_t)slot_next * slot_floats) + (NSYM / 16) * j;
#pragma unroll
            for (int r = 0; r < NSYM / 16; ++r) pre[r] = g4[r];
        }
    }
    for (; task < ntask; task += stride) {
        STAMP(6);
        // ---- which frame (the work-list entry was requested one task ahead)
        int qtype, slot; bool active;
        if (work) {
            (void)pick(task, qtype, active);
            slot = slot_next;
            if (task + stride < ntask) { int qt; bool ac; slot_next = work[pick(task + stride, qt, ac)]; }
        } else {
            const int item = task * DQ_FRAMES + q;
            active = item < n_plain;
            slot = active ? item : n_plain - 1;
            qtype = (int)types[slot];
        }
        const float *gs = fsym + (size_t)slot * slot_floats;         // the frame's slot, read in place
        m17gpu_rec_dev *rec = &recs[slot];
        const uint32_t r0_keep = work ? reinterpret_cast<const uint32_t *>(rec)[0] : (uint32_t)qtype;
        STAMP(0);
        if (ONLY) {
            const float *gs_next = (NSYM != 0 && task + stride < ntask) ? fsym + (size_t)slot_next * slot_floats : nullptr;
            decode_quad_pass<Frame::kDecw, ONLY, NSYM>(F, gs, gt_row, lich_row, ONLY, j, active, r0_keep, rec, C1, C2, genc, gerr, acc_, last_,
                                                       gs_next, pre);
        } else {
#pragma unroll 1
            for (int pass = 0; pass < 3; ++pass) {
                const int type = (pass == 0) ? 2 : (pass == 1 ? 1 : 3);
                if (__ballot(qtype == type) == 0ull) continue;
                if (row_type != type) {
                    for (int i = lane; i < 512; i += 64) gt_row[i] = dq_entry((i < 488) ? (int)c_tab.gather[type][i] : -1);
                    row_type = type;
                    wave_fence();
                }
                decode_quad_pass<Frame::kDecw, 0, NSYM>(F, gs, gt_row, lich_row, type, j, active && qtype == type, r0_keep, rec, C1, C2, genc, gerr, acc_, last_,
                                                        nullptr, pre);
            }
        }
        STAMP(5);
    }
#ifdef M17_STAMPS
    if (ONLY == 2 && wg == 0 && threadIdx.x == 0) { for (int i = 0; i < 7; ++i) g_stamps[i] = acc_[i]; g_stamps[8] = (unsigned long long)ntask; }
#endif
}

// Plain-batch entry and A/B: one role per launch, four waves per workgroup.
template <int ONLY>
__global__ __launch_bounds__(256, ONLY == 2 ? 4 : 2)
void k_decode_quad(const float *__restrict__ fsym, const int32_t *__restrict__ work,
                   const int32_t *__restrict__ nwork, int cap,
                   const uint8_t *__restrict__ types, int n_plain,
                   m17gpu_rec_dev *__restrict__ recs,
                   const uint16_t *__restrict__ genc, const uint16_t *__restrict__ gerr, int skip_stream,
                   int slot_floats)
{
    __shared__ DqShared<ONLY, 4> sh;
    decode_quad_body<ONLY, 4>(sh, (int)blockIdx.x, (int)gridDim.x, fsym, work, nwork, cap, types, n_plain, recs, genc, gerr,
                              skip_stream, slot_floats);
}

// The work lists of one m17gpu_rx_blocks call in ONE launch.  Workgroups [0, n_other) take the link-setup and packet
// lists on two of their four waves (any-type storage for two waves fits the LDS the stream role needs for four), the
// rest the stream list.  A lone link-setup frame costs a wave the latency of a whole 244-step task (~40 us): in a
// launch of its own, behind the stream launch, that was 38 us added to every call; here it runs beside the stream
// tasks.  The any-type workgroups come first so that they are placed at once; with their lists empty they leave
// before touching their tables.
__global__ __launch_bounds__(256, 5)
void k_decode_lists(const float *__restrict__ fsym, const int32_t *__restrict__ work,
                    const int32_t *__restrict__ nwork, int cap, m17gpu_rec_dev *__restrict__ recs,
                    const uint16_t *__restrict__ genc, const uint16_t *__restrict__ gerr, int slot_floats, int n_other)
{
    __shared__ union U { DqShared<2, 4> s; DqShared<0, 2> o; __device__ U() {} } sh;
    static_assert(sizeof(DqShared<0, 2>) <= sizeof(DqShared<2, 4>), "the any-type role must fit the stream role's LDS");
    if ((int)blockIdx.x < n_other) {
        decode_quad_body<0, 2>(sh.o, (int)blockIdx.x, n_other, fsym, work, nwork, cap, nullptr, 0, recs, genc, gerr, 1, slot_floats,
                               threadIdx.x >= 128);
    } else {
        decode_quad_body<2, 4>(sh.s, (int)blockIdx.x - n_other, (int)gridDim.x - n_other, fsym, work, nwork, cap, nullptr, 0,
                               recs, genc, gerr, 0, slot_floats);
    }
}

// slot_impl 1 (round 5, the round-4 review's item 4): the same launch for PLAIN stream slots -- the framer stores a stream
// frame as its 192 symbols (768 B instead of 1,600), the stream role stages them in LDS (19 KB per wave: two workgroups
// per CU instead of five) and gathers there.
__global__ __launch_bounds__(256, 2)
void k_decode_lists_p(const float *__restrict__ fsym, const int32_t *__restrict__ work,
                      const int32_t *__restrict__ nwork, int cap, m17gpu_rec_dev *__restrict__ recs,
                      const uint16_t *__restrict__ genc, const uint16_t *__restrict__ gerr, int slot_floats, int n_other)
{
    __shared__ union U { DqShared<2, 4, kFrameSyms> s; DqShared<0, 2> o; __device__ U() {} } sh;
    if ((int)blockIdx.x < n_other) {
        decode_quad_body<0, 2>(sh.o, (int)blockIdx.x, n_other, fsym, work, nwork, cap, nullptr, 0, recs, genc, gerr, 1, slot_floats,
                               threadIdx.x >= 128);
    } else {
        decode_quad_body<2, 4, kFrameSyms>(sh.s, (int)blockIdx.x - n_other, (int)gridDim.x - n_other, fsym, work, nwork, cap, nullptr, 0,
                                           recs, genc, gerr, 0, slot_floats);
    }
}

} // namespace m17dev
