// m17_sync_common.hip -- helpers shared by the timing/framer kernels (included by m17gpu_capi.hip
// after m17_kernels.hip; same namespace): LDS-only barrier and wave fence, s_memtime phase stamps
// of the instrumented build, frame-slot store, the packed (matched, derivative) FIR with
// the taps in VGPRs (two-wave kernel), the frame-sync check on lane groups of eight, the unlocked framer's hunt pass
// and the record writer.
#pragma clang fp contract(off)

// The hand-scheduled packed-FIR sequences here and in m17_fir_sgpr.inc place every v_pk_mul_f32 (op_sel) product two
// instructions ahead of the add that reads it: one wait state is what gfx950 needs for that forwarding, and the
// assembler does not check it.  Another target may need more, silently: refuse to build for it.
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__)
#error "m17_sdr_amd kernels are written and hazard-checked for gfx950 (MI355X) only"
#endif

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

#ifdef M17_STAMPS
__device__ unsigned long long g_stamps[16];
__device__ unsigned long long g_chan_stamps[4096][8];      // per channel: the phase accumulators of its wave
__device__ unsigned long long g_chan_stamps_x[4096][2];    // ... and its epilogue
__device__ unsigned long long g_wave_span[16384][3];       // per channel: s_memrealtime at entry / exit (100 MHz, one clock for the chip), HW_ID
#define DBGCNT(i) do { if (t == 0) atomicAdd(&g_stamps[12 + (i)], 1ull); } while (0)
#define STAMP(i) do { unsigned long long now_; __builtin_amdgcn_sched_barrier(0); \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(now_) :: "memory"); __builtin_amdgcn_sched_barrier(0); \
    acc_[i] += now_ - last_; last_ = now_; } while (0)
#else
#define STAMP(i) do {} while (0)
#define DBGCNT(i) do {} while (0)
#endif

// rx_sync_filter (m17_rx_sync.cpp:25-31): matched and derivative filter as one packed (s, d)
// chain, ascending order, bare first product.  xs = delay line at this instant, tp = 32 tap pairs.
//
// The 31 packed multiplies and 30 packed adds are written out with every product formed two instructions before
// the add that consumes it.  Left to the compiler, half of the products were consumed by the very next instruction:
// a packed multiply with op_sel needs a wait state before a dependent read on gfx950, so the schedule carried 20
// s_nop and 16 separate LDS waits per round of 142 issue slots.  Same instructions, same order of the adds.
//   P, Q alternate as the product in flight; X = (x[2q], x[2q+1]); T = (matched tap, derivative tap) of one sample.
#define M17_FIR4(A, P, Q, X0, X1, T0, T1, T2, T3)                                             \
    asm volatile("v_pk_mul_f32 %1, %3, %5 op_sel_hi:[0,1]\n\t"                                 \
                 "v_pk_add_f32 %0, %0, %2\n\t"                                                 \
                 "v_pk_mul_f32 %2, %3, %6 op_sel:[1,0]\n\t"                                    \
                 "v_pk_add_f32 %0, %0, %1\n\t"                                                 \
                 "v_pk_mul_f32 %1, %4, %7 op_sel_hi:[0,1]\n\t"                                 \
                 "v_pk_add_f32 %0, %0, %2\n\t"                                                 \
                 "v_pk_mul_f32 %2, %4, %8 op_sel:[1,0]\n\t"                                    \
                 "v_pk_add_f32 %0, %0, %1"                                                      \
                 : "+v"(A), "=&v"(P), "+v"(Q) : "v"(X0), "v"(X1), "v"(T0), "v"(T1), "v"(T2), "v"(T3))
__device__ __forceinline__ v2f fir_pair(const float *xs, const float4 (&tp)[16])
{
    // (the offset may be odd: plain float reads, the compiler pairs them into ds_read2_b32)
    v2f xv[16];
#pragma unroll
    for (int q = 0; q < 15; ++q) xv[q] = (v2f){xs[2 * q], xs[2 * q + 1]};
    xv[15] = (v2f){xs[30], 0.0f};
#define M17_TLO(q) ((v2f){tp[q].x, tp[q].y})
#define M17_THI(q) ((v2f){tp[q].z, tp[q].w})
    v2f acc, P, Q;
    // samples 0..3: acc = p0 (bare), then + p1, + p2; p3 stays in flight
    asm volatile("v_pk_mul_f32 %0, %3, %5 op_sel_hi:[0,1]\n\t"
                 "v_pk_mul_f32 %2, %3, %6 op_sel:[1,0]\n\t"
                 "v_pk_mul_f32 %1, %4, %7 op_sel_hi:[0,1]\n\t"
                 "v_pk_add_f32 %0, %0, %2\n\t"
                 "v_pk_mul_f32 %2, %4, %8 op_sel:[1,0]\n\t"
                 "v_pk_add_f32 %0, %0, %1"
                 : "=&v"(acc), "=&v"(P), "=&v"(Q)
                 : "v"(xv[0]), "v"(xv[1]), "v"(M17_TLO(0)), "v"(M17_THI(0)), "v"(M17_TLO(1)), "v"(M17_THI(1)));
    M17_FIR4(acc, P, Q, xv[2], xv[3], M17_TLO(2), M17_THI(2), M17_TLO(3), M17_THI(3));
    M17_FIR4(acc, P, Q, xv[4], xv[5], M17_TLO(4), M17_THI(4), M17_TLO(5), M17_THI(5));
    M17_FIR4(acc, P, Q, xv[6], xv[7], M17_TLO(6), M17_THI(6), M17_TLO(7), M17_THI(7));
    M17_FIR4(acc, P, Q, xv[8], xv[9], M17_TLO(8), M17_THI(8), M17_TLO(9), M17_THI(9));
    M17_FIR4(acc, P, Q, xv[10], xv[11], M17_TLO(10), M17_THI(10), M17_TLO(11), M17_THI(11));
    M17_FIR4(acc, P, Q, xv[12], xv[13], M17_TLO(12), M17_THI(12), M17_TLO(13), M17_THI(13));
    // samples 28, 29, 30 and the product still in flight
    asm volatile("v_pk_mul_f32 %1, %3, %5 op_sel_hi:[0,1]\n\t"
                 "v_pk_add_f32 %0, %0, %2\n\t"
                 "v_pk_mul_f32 %2, %3, %6 op_sel:[1,0]\n\t"
                 "v_pk_add_f32 %0, %0, %1\n\t"
                 "v_pk_mul_f32 %1, %4, %7 op_sel_hi:[0,1]\n\t"
                 "v_pk_add_f32 %0, %0, %2\n\t"
                 "v_pk_add_f32 %0, %0, %1"
                 : "+v"(acc), "=&v"(P), "+v"(Q)
                 : "v"(xv[14]), "v"(xv[15]), "v"(M17_TLO(14)), "v"(M17_THI(14)), "v"(M17_TLO(15)));
#undef M17_TLO
#undef M17_THI
    return acc;
}


// the lane's sign mask of the frame-sync check below: lane 8k+i <-> sframe[k][i] (m17_rx_frame.cpp:5-12)
__device__ __forceinline__ unsigned sync_sign_mask(int lane)
{
    constexpr unsigned sneg[6] = M17_SYNC_NEG_MASKS;
    constexpr unsigned long long sneg48 = (unsigned long long)sneg[0] | ((unsigned long long)sneg[1] << 8) |
        ((unsigned long long)sneg[2] << 16) | ((unsigned long long)sneg[3] << 24) | ((unsigned long long)sneg[4] << 32) |
        ((unsigned long long)sneg[5] << 40);
    return (unsigned)((sneg48 >> lane) & 1ull) << 31;
}

template <int CTRL> __device__ __forceinline__ float dpp_own_f(float v)      // lanes without a source keep their value
{
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(v), __float_as_int(v), CTRL, 0xF, 0xF, false));
}

// m17_sync_check (m17_rx_frame.cpp:47-81) + find_variance (:22-43) on ONE frame head held by the wave, lane l holding
// vect[l & 7] (vs).  The six template sums run side by side in lanes 0..47: lane 8k+i starts from the exact product
// x = vect[i] * sframe[k][i] (a sign flip, sgn = the lane's sign mask), seven row_shr:1 adds then leave the ascending
// sum ((x0+x1)+x2)+... in lane 8k+7.  Argmax with the reference's strict '>' from (0, 0): sums clamped at 0 compare as
// unsigned integers, so the maximum is a scalar max of six lane reads and the winner the lowest template that equals
// it (template 0 when no sum is positive).  |vect| min / max by three DPP exchanges inside the groups of eight lanes;
// v_max / v_min skip NaNs exactly like the reference's two compares, except a NaN in vect[0], which it keeps (-> 1.0).
__device__ __forceinline__ SyncResult sync_check_lanes8(float vs, unsigned sgn)
{
    const float x = __uint_as_float(__float_as_uint(vs) ^ sgn);
    float s = x;
#pragma unroll
    for (int j = 0; j < 7; ++j) s = dpp_row_shr1(s) + x;
    const unsigned tb = __float_as_uint(__builtin_fmaxf(s, 0.0f));
    unsigned M = (unsigned)__builtin_amdgcn_readlane((int)tb, 7);
#pragma unroll
    for (int k = 1; k < 6; ++k) { const unsigned o = (unsigned)__builtin_amdgcn_readlane((int)tb, 8 * k + 7); M = o > M ? o : M; }
    const unsigned long long eq = __builtin_amdgcn_ballot_w64(tb == M) & 0x0000808080808080ull;
    SyncResult r;
    r.type = (int)__builtin_ctzll(eq) >> 3;
    const unsigned long long neg = __builtin_amdgcn_ballot_w64(x < 0.0f);
    r.votes = (int)__popcll((neg >> (8 * r.type)) & 0xFFull);
    const float a = __builtin_fabsf(vs);
    float mx = a, mn = a;
    mx = __builtin_fmaxf(mx, dpp_own_f<0xB1>(mx)); mn = __builtin_fminf(mn, dpp_own_f<0xB1>(mn));      // lane ^ 1
    mx = __builtin_fmaxf(mx, dpp_own_f<0x4E>(mx)); mn = __builtin_fminf(mn, dpp_own_f<0x4E>(mn));      // lane ^ 2
    mx = __builtin_fmaxf(mx, dpp_own_f<0x141>(mx)); mn = __builtin_fminf(mn, dpp_own_f<0x141>(mn));    // row_half_mirror
    mx = unif(mx); mn = unif(mn);
    float var = (mx - mn) / mx;
    if (var != var) var = 1.0f;
    const float v0 = unif(vs);
    if (v0 != v0) var = 1.0f;
    r.variance = var;
    return r;
}

// One pass of the unlocked framer's hunt (m17_rx_sym, m17_rx_frame.cpp:155-171, over m17_unlocked_sync_check :92-103):
// candidates are the block's symbols pos .. pos + 63 (those below n), the window of candidate j the eight symbols
// j - 7 .. j; ld(i) returns symbol i of the block (i >= -7: the ring keeps what came before).  Returns the offset of
// the first accepted candidate, with its check result in `out`, or -1.
//   Pre-filter on sign bits: acceptance needs votes == 0 for the winning template, i.e. no symbol of the window with
//   the sign OPPOSITE to it (zeros and NaNs never vote), and the winner must be one of types 1..4: a window that is
//   incompatible with all four cannot be accepted.  Nor can one that holds a symbol WITHOUT a sign: an exact zero
//   (find_variance then returns 1, or NaN -> 1, never < 0.3) or a NaN (every template sum is NaN, no sum beats 0, the
//   type stays 0) -- without that test a squelched channel, whose symbols are all zeros or NaNs and compatible with
//   everything, would send every candidate to the exact check.  Lane l loads ONE symbol; two ballots give the signs
//   of the 71 symbols as bit strings, the window of candidate c is bits c .. c + 7 of them.  In noise 1.6 % of the windows
//   pass; those are checked exactly, one after the other in position order, by the lane-group check the locked
//   framer uses -- instead of every lane loading its own eight symbols and the whole wave running the six-template
//   check whenever one lane's window passes, which in noise is always.
template <class Ld>
__device__ __forceinline__ int hunt_pass(int pos, int n, int gl, Ld ld, SyncResult &out)
{
    const bool cand = pos + gl < n;
    const float v = ld(cand ? pos + gl : pos);
    const float t = ld(pos - 7 + (gl < 7 ? gl : 0));
    const unsigned long long p64 = __builtin_amdgcn_ballot_w64(cand && v > 0.0f), n64 = __builtin_amdgcn_ballot_w64(cand && v < 0.0f);
    const unsigned long long p7 = __builtin_amdgcn_ballot_w64(gl < 7 && t > 0.0f), n7 = __builtin_amdgcn_ballot_w64(gl < 7 && t < 0.0f);
    // bit i of (hi : lo) = symbol pos - 7 + i
    const unsigned long long plo = p7 | (p64 << 7), nlo = n7 | (n64 << 7);
    const unsigned long long pmid = (plo >> 32) | ((p64 >> 57) << 32), nmid = (nlo >> 32) | ((n64 >> 57) << 32);   // bits 32 .. 95
    const int sh = gl & 31;
    const unsigned pw = (unsigned)((gl < 32 ? plo : pmid) >> sh) & 0xFFu, nw = (unsigned)((gl < 32 ? nlo : nmid) >> sh) & 0xFFu;
    constexpr unsigned tn[4] = {sync_neg_mask(1), sync_neg_mask(2), sync_neg_mask(3), sync_neg_mask(4)};     // bit i set: template symbol i is -1
    bool ok = false;
#pragma unroll
    for (int k = 0; k < 4; ++k) ok = ok || (((pw & tn[k]) | (nw & (~tn[k] & 0xFFu))) == 0u);
    unsigned long long cm = __builtin_amdgcn_ballot_w64(cand && ok && (pw | nw) == 0xFFu);
    const unsigned sgn = sync_sign_mask(gl);
    while (cm) {
        const int l = (int)__builtin_ctzll(cm);
        const float vs = ld(pos + l - 7 + (gl & 7));
        // A window that passed the sign filter is accepted only with (max - min) / max < 0.3 over its magnitudes
        // (find_variance, m17_rx_frame.cpp:22-43,92-103).  min < 0.69 max puts that quotient above 0.3099 whatever the
        // rounding of its two operations: such a window is rejected here for a dozen instructions instead of the check's
        // sixty-odd (template sums, argmax, votes, the exact quotient) -- in noise that is nearly every window that got this
        // far (round 5: item 5 of the round-4 review).  No NaN and no zero reaches this point (the sign filter).
        {
            const float a = __builtin_fabsf(vs);
            float mx = a, mn = a;
            mx = __builtin_fmaxf(mx, dpp_own_f<0xB1>(mx)); mn = __builtin_fminf(mn, dpp_own_f<0xB1>(mn));      // lane ^ 1
            mx = __builtin_fmaxf(mx, dpp_own_f<0x4E>(mx)); mn = __builtin_fminf(mn, dpp_own_f<0x4E>(mn));      // lane ^ 2
            mx = __builtin_fmaxf(mx, dpp_own_f<0x141>(mx)); mn = __builtin_fminf(mn, dpp_own_f<0x141>(mn));    // row_half_mirror
            if (unif(mn) < 0.69f * unif(mx)) { cm &= cm - 1ull; continue; }
        }
        const SyncResult r = sync_check_lanes8(vs, sgn);
        if (sync_accept(r, false)) { out = r; return l; }
        cm &= cm - 1ull;
    }
    return -1;
}

// one record: five words from scalars, eleven zero words, lanes 0..15
__device__ __forceinline__ void emit_record_wave(m17gpu_rec_dev *crecs, int rec_cap, int idx, int gl,
                                                 uint32_t w0, uint32_t w1, float var, uint32_t block, uint32_t sympos)
{
    if (idx >= rec_cap) return;
    int v = 0;
    asm("v_writelane_b32 %0, %1, 0\n\tv_writelane_b32 %0, %2, 1\n\tv_writelane_b32 %0, %3, 2\n\t"
        "v_writelane_b32 %0, %4, 3\n\tv_writelane_b32 %0, %5, 4"
        : "+v"(v) : "s"(uni((int)w0)), "s"(uni((int)w1)), "s"(uni(__float_as_int(var))), "s"(uni((int)block)), "s"(uni((int)sympos)));
    if (gl < 16) reinterpret_cast<int *>(&crecs[idx])[gl] = v;
}


} // namespace m17dev
