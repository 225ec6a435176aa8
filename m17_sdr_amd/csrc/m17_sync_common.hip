// m17_sync_common.hip -- helpers shared by the timing/framer kernels (included by m17gpu_capi.hip
// after m17_kernels.hip; same namespace): LDS-only barrier and wave fence, s_memtime phase stamps
// of the instrumented build, wave-parallel sync correlator, hunt pre-filter, packed (matched,
// derivative) FIR.
#pragma clang fp contract(off)

namespace m17dev {

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also drains
// vmcnt, i.e. it would wait for the prefetched next block and for every symbol /
// record store at each barrier -- the threads exchange data through LDS only,
// global memory is written for later kernels.
__device__ __forceinline__ void lds_barrier()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

__device__ __forceinline__ void wave_fence()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront", "local");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront", "local");
}

// A completed frame into its slot of the decoder's workspace (m17_dev.h: kSlotFloats).  src(q) is symbol q of the
// frame.  Lane gl of a channel's LPC lanes writes slot entries 8 + 4 (gl + LPC r) + {0..3} as one 16-byte store; which
// symbols go there is fixed per lane, so the lane keeps those table bytes packed in registers for the whole kernel
// (a table in LDS put two dependent LDS round trips in front of every batch of stores).
template <int LPC> struct RegroupLane {
    static constexpr int R = (kRegroup / 4 + LPC - 1) / LPC;          // 16-byte groups per lane: 7 / 4 / 2
    uint32_t w[R];
    __device__ __forceinline__ void load(int gl)
    {
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int q = 4 * (gl + LPC * r);
            w[r] = (q < kRegroup) ? *reinterpret_cast<const uint32_t *>(&c_tab.regroup[q]) : 0u;
        }
    }
    __device__ __forceinline__ int sym(int r, int c) const { return 8 + (int)((w[r] >> (8 * c)) & 0xFFu); }
};
template <int LPC, class Src>
__device__ __forceinline__ void store_frame_slot(float *__restrict__ fd, int type, int gl, const RegroupLane<LPC> &rg, Src src)
{
    if (type == 2) {
        if (gl < 8) fd[gl] = src(gl);
        constexpr int R = RegroupLane<LPC>::R;
        float4 t[R];
#pragma unroll
        for (int r = 0; r < R; ++r) t[r] = make_float4(src(rg.sym(r, 0)), src(rg.sym(r, 1)), src(rg.sym(r, 2)), src(rg.sym(r, 3)));
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int q = 4 * (gl + LPC * r);
            if (q < kRegroup) *reinterpret_cast<float4 *>(fd + 8 + q) = t[r];
        }
    } else {
        constexpr int R = (kFrameSyms / 4 + LPC - 1) / LPC;            // 3 / 2 / 1
        float4 t[R];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int q = 4 * (gl + LPC * r);
            const int qq = q < kFrameSyms ? q : 0;
            t[r] = make_float4(src(qq), src(qq + 1), src(qq + 2), src(qq + 3));
        }
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int q = 4 * (gl + LPC * r);
            if (q < kFrameSyms) *reinterpret_cast<float4 *>(fd + q) = t[r];
        }
    }
}

// candidate pre-filter of the sync hunt: m17_unlocked_sync_check needs votes == 0 for
// the winning template, i.e. no symbol of the window may have the sign OPPOSITE to
// that template (zeros and NaNs never vote, m17_rx_frame.cpp:77-80).  A window that
// is incompatible with all four acceptable templates (types 1..4) cannot be accepted.
__device__ __forceinline__ bool hunt_compatible(const float v[8])
{
    unsigned pos = 0, neg = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        pos |= (v[i] > 0.0f) ? (1u << i) : 0u;
        neg |= (v[i] < 0.0f) ? (1u << i) : 0u;
    }
    constexpr unsigned tn[4] = {sync_neg_mask(1), sync_neg_mask(2), sync_neg_mask(3), sync_neg_mask(4)};     // bit i set: template symbol i is -1
    bool ok = false;
#pragma unroll
    for (int k = 0; k < 4; ++k) ok = ok || (((pos & tn[k]) == 0u) && ((neg & (~tn[k] & 0xFFu)) == 0u));
    return ok;
}

#ifdef M17_STAMPS
__device__ unsigned long long g_stamps[16];
__device__ unsigned long long g_chan_stamps[4096][8];      // per channel: the phase accumulators of its wave
#define DBGCNT(i) do { if (t == 0) atomicAdd(&g_stamps[12 + (i)], 1ull); } while (0)
#define STAMP(i) do { unsigned long long now_; __builtin_amdgcn_sched_barrier(0); \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(now_) :: "memory"); __builtin_amdgcn_sched_barrier(0); \
    acc_[i] += now_ - last_; last_ = now_; } while (0)
#else
#define STAMP(i) do {} while (0)
#define DBGCNT(i) do {} while (0)
#endif

// The control state of a channel is wave-uniform by construction; telling the
// compiler (readfirstlane) keeps it in SGPRs and the branches on the scalar unit
// instead of EXEC-mask control flow.
__device__ __forceinline__ unsigned long long uni64(unsigned long long v)
{
    const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)v);
    const unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(v >> 32));
    return ((unsigned long long)hi << 32) | lo;
}


// m17_sync_check (m17_rx_frame.cpp:47-81) on ONE 8-symbol vector that every lane
// of the calling wave holds: lane k < 6 accumulates template k, the in-order
// strict-'>' argmax walks the six lanes with readlane, the votes are a ballot.
__device__ __forceinline__ SyncResult sync_check_wave(const float v[8])
{
    constexpr unsigned negs[6] = M17_SYNC_NEG_MASKS;
    const int lane = lane_id();
    unsigned neg = negs[0];
#pragma unroll
    for (int k = 1; k < 6; ++k) neg = (lane == k) ? negs[k] : neg;
    float s = (neg & 1u) ? -v[0] : v[0];
#pragma unroll
    for (int i = 1; i < 8; ++i) s = (neg >> i & 1u) ? s - v[i] : s + v[i];
    float best = 0.0f; int nmax = 0;
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        const float sk = bcast_lane(s, k);
        if (sk > best) { best = sk; nmax = k; }
    }
    unsigned nm = negs[0];
#pragma unroll
    for (int k = 1; k < 6; ++k) nm = (nmax == k) ? negs[k] : nm;
    float mine = v[0];
#pragma unroll
    for (int i = 1; i < 8; ++i) mine = (lane == i) ? v[i] : mine;
    const bool bad = (lane < 8) && ((nm >> lane & 1u) ? (mine > 0.0f) : (mine < 0.0f));
    const int votes = __popcll(__ballot(bad));
    float mmin = fabsf(v[0]), mmax = mmin;
#pragma unroll
    for (int i = 1; i < 8; ++i) {
        const float a = fabsf(v[i]);
        if (a > mmax) mmax = a;
        else if (a < mmin) mmin = a;
    }
    float var = (mmax - mmin) / mmax;
    if (var != var) var = 1.0f;
    SyncResult r; r.type = nmax; r.votes = votes; r.variance = var;
    return r;
}

// wave-wide max / min of a per-lane int on DPP row operations
template <int CTRL, int RM = 0xF> __device__ __forceinline__ int dpp_keep(int v)
{
    return __builtin_amdgcn_update_dpp(v, v, CTRL, RM, 0xF, false);      // lanes without a source keep their own value
}
__device__ __forceinline__ int wave_max_i(int v)
{
    v = max(v, dpp_keep<0x111>(v)); v = max(v, dpp_keep<0x112>(v)); v = max(v, dpp_keep<0x114>(v)); v = max(v, dpp_keep<0x118>(v));
    v = max(v, dpp_keep<0x142, 0xA>(v)); v = max(v, dpp_keep<0x143, 0xC>(v));
    return __builtin_amdgcn_readlane(v, 63);
}
__device__ __forceinline__ int wave_min_i(int v)
{
    v = min(v, dpp_keep<0x111>(v)); v = min(v, dpp_keep<0x112>(v)); v = min(v, dpp_keep<0x114>(v)); v = min(v, dpp_keep<0x118>(v));
    v = min(v, dpp_keep<0x142, 0xA>(v)); v = min(v, dpp_keep<0x143, 0xC>(v));
    return __builtin_amdgcn_readlane(v, 63);
}

// One symbol instant: the matched (s) and derivative (d) 31-tap dot products over the
// delay line xs[0..30], strictly in the reference's order (rx_sync_filter,
// m17_rx_sync.cpp:25-31: bare first product, then += in ascending tap order, separate
// multiply and add).  31 taps in four groups of 8 (last: 7); the next group's LDS reads
// (taps are wave-uniform broadcasts) are issued before the current group's chain.

// tp4: the branch's 32 (matched, derivative) tap pairs = 16 float4 (md4 is unused: kept for call compatibility)
__device__ __forceinline__ void fir_instant(const float *xs, const float4 *tp4, const float4 *, float &s, float &d)
{
    // Packed form: lane pair (s, d) = (matched, derivative) accumulators, one v_pk_mul_f32 and
    // one v_pk_add_f32 per tap -- the two chains of the reference advance in lock step, each
    // still in its own ascending order with separate multiply and add.  Taps sit in LDS as
    // (mf, md) pairs so a ds_read_b128 delivers two ready register pairs.
    const float2 *xp = reinterpret_cast<const float2 *>(xs);
    float2 xv[15];
    float4 tp[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) tp[q] = tp4[q];
#pragma unroll
    for (int q = 0; q < 15; ++q) xv[q] = xp[q];
    const float xl = xs[30];
    v2f acc = (v2f){xv[0].x, xv[0].x} * (v2f){tp[0].x, tp[0].y};          // bare first product
    acc = acc + (v2f){xv[0].y, xv[0].y} * (v2f){tp[0].z, tp[0].w};
#pragma unroll
    for (int q = 1; q < 15; ++q) {
        acc = acc + (v2f){xv[q].x, xv[q].x} * (v2f){tp[q].x, tp[q].y};
        acc = acc + (v2f){xv[q].y, xv[q].y} * (v2f){tp[q].z, tp[q].w};
    }
    acc = acc + (v2f){xl, xl} * (v2f){tp[15].x, tp[15].y};
    s = acc.x; d = acc.y;
}

} // namespace m17dev
