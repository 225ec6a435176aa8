// m17_sync_grp.hip -- k_sync_frame_grp<LPC>: timing recovery + sync correlator + framer
// with LPC lanes per channel (64, 32 or 16), i.e. 1, 2 or 4 channels per wave.
//
// Reference: m17_rx_sync_samples (m17_rx_sync.cpp:77-99), m17_rx_sym (m17_rx_frame.cpp:126-177).
//
// What round-1 profiling showed: one channel's control code -- vote
// scan, phase stepping, framer -- is ~1,000 scalar-ish instructions per 1920-sample block,
// and a wave that owns a single channel executes them for one channel's benefit.  Here
// every control variable is a per-lane value that is equal within a lane GROUP, so one
// instruction stream drives 64/LPC channels; divergence between the channels of a wave
// (one hunting, one locked, ...) is ordinary EXEC-mask divergence.  The symbol instants
// are taken in ROUNDS of LPC consecutive instants: the FIR of a round, the vote ballot of
// the group, its popcount prefix and threshold test; a crossing ends the round early and
// the next round starts behind it under the new polyphase branch, so nothing past a
// crossing is ever evaluated.  The branch taps sit in registers across rounds.
//
// LPC is picked from the channel count so that the chip always has >= ~1 wave per SIMD:
// 64 up to 1,024 channels, 32 up to 2,048, 16 beyond (m17gpu_capi.hip).
#pragma clang fp contract(off)

namespace m17dev {

struct GrpChan {                           // LDS of one channel
    float x[kTaps - 1 + kDiscOut + 2];     // delay-line history (30) + this block's 384 inputs
    float h[8 + 208];                      // m_sync (8) followed by the block's symbols
    float f[kFrameSyms];                   // m_f_sym
};

template <int LPC> struct GrpCfg {
    static constexpr int G = 64 / LPC;                 // channels per wave
    static constexpr int CPW = 4 * G;                  // channels per 256-thread workgroup
    static constexpr unsigned long long MASK = (LPC == 64) ? ~0ull : ((1ull << (LPC & 63)) - 1ull);
};

// m17_sync_check (m17_rx_frame.cpp:47-81) on one 8-symbol vector per lane group; every
// lane of the group holds v.  Lane k < 6 of the group accumulates template k.
template <int LPC>
__device__ __forceinline__ SyncResult sync_check_grp(const float v[8], int gl, int gbase, int gshift)
{
    constexpr unsigned negs[6] = M17_SYNC_NEG_MASKS;
    unsigned neg = negs[0];
#pragma unroll
    for (int k = 1; k < 6; ++k) neg = (gl == k) ? negs[k] : neg;
    float s = (neg & 1u) ? -v[0] : v[0];
#pragma unroll
    for (int i = 1; i < 8; ++i) s = (neg >> i & 1u) ? s - v[i] : s + v[i];
    float best = 0.0f; int nmax = 0;
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        const float sk = __shfl(s, gbase + k, 64);
        if (sk > best) { best = sk; nmax = k; }
    }
    unsigned nm = negs[0];
#pragma unroll
    for (int k = 1; k < 6; ++k) nm = (nmax == k) ? negs[k] : nm;
    float mine = v[0];
#pragma unroll
    for (int i = 1; i < 8; ++i) mine = (gl == i) ? v[i] : mine;
    const bool bad = (gl < 8) && ((nm >> gl & 1u) ? (mine > 0.0f) : (mine < 0.0f));
    const int votes = (int)__popcll((__builtin_amdgcn_ballot_w64(bad) >> gshift) & GrpCfg<LPC>::MASK);
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

// one record, written by lanes gl < 16 of the group
__device__ __forceinline__ void emit_record_grp(m17gpu_rec_dev *recs, int rec_cap, int idx, int gl,
                                                uint32_t w0, uint32_t w1, float var, uint32_t block, uint32_t sympos)
{
    if (idx >= rec_cap || gl >= 16) return;
    uint32_t v = 0;
    if (gl == 0) v = w0;
    if (gl == 1) v = w1;
    if (gl == 2) v = __float_as_uint(var);
    if (gl == 3) v = block;
    if (gl == 4) v = sympos;
    reinterpret_cast<uint32_t *>(&recs[idx])[gl] = v;
}

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

// ---- lane-group primitives on DPP rows (a row is 16 lanes; LPC is 16, 32 or 64) -------------------
template <int CTRL, int ROWMASK = 0xF>
__device__ __forceinline__ int dpp_add_i(int acc, int v)            // acc + (v moved by CTRL, 0 where no source lane)
{
    return acc + __builtin_amdgcn_update_dpp(0, v, CTRL, ROWMASK, 0xF, true);
}
// inclusive prefix sum over the LPC lanes of every group
template <int LPC>
__device__ __forceinline__ int group_scan_incl(int v)
{
    v = dpp_add_i<0x111>(v, v);          // row_shr:1
    v = dpp_add_i<0x112>(v, v);          // row_shr:2
    v = dpp_add_i<0x114>(v, v);          // row_shr:4
    v = dpp_add_i<0x118>(v, v);          // row_shr:8
    if (LPC >= 32) v = dpp_add_i<0x142, 0xA>(v, v);      // row_bcast:15 into rows 1 and 3
    if (LPC == 64) v = dpp_add_i<0x143, 0xC>(v, v);      // row_bcast:31 into rows 2 and 3
    return v;
}
// the value held by the LAST lane of each group, in every lane of the group (LDS crossbar, no memory)
template <int LPC>
__device__ __forceinline__ int group_bcast_last_i(int v)
{
    if (LPC == 64) return __builtin_amdgcn_readlane(v, 63);
    // ds_swizzle bit mode: lane' = ((lane & and) | or) ^ xor inside each 32-lane half
    if (LPC == 32) return __builtin_amdgcn_ds_swizzle(v, 0x1F << 5);               // and 0, or 31
    return __builtin_amdgcn_ds_swizzle(v, 0x10 | (0x0F << 5));                      // and 16, or 15
}
template <int LPC>
__device__ __forceinline__ float group_bcast_last_f(float v)
{
    return __int_as_float(group_bcast_last_i<LPC>(__float_as_int(v)));
}

// One pass of the timing loop over up to W*LPC instants starting at input p (clk == 0 on
// entry): W independent FIR chains per lane, then the vote ticks in time order (sync_update
// m17_rx_sync.cpp:38-42, m17_sync_adjust :45-72).  The first threshold crossing ends the pass.
template <int LPC, int W>
__device__ __forceinline__ void timing_pass(GrpChan &my, const float4 (&tp)[16], int gl, int gbase, int gshift,
                                            unsigned long long incl, int thresh, int rem, int &p, int &m_idx,
                                            int &clk, int &thr, int &index, float &sum, float &dif)
{
    using Cfg = GrpCfg<LPC>;
    float s[W], d[W];
#pragma unroll
    for (int j = 0; j < W; ++j) {
        const int inst = gl + LPC * j;
        const v2f a = fir_pair(my.x + p + 2 * (inst < rem ? inst : 0), tp);
        s[j] = a.x; d[j] = a.y;
    }
    if (W == 1) {
        // Fast path, the common case while a channel tracks: a full round, every group of the wave stays inside
        // its threshold.  The vote counter after each tick is a prefix sum over the group's lanes (DPP row
        // operations), no ballots, no per-group mask shifts.  Anything else -- a crossing in any group of the
        // wave, a partial round behind a crossing, the symbol index at -1 after a downward wrap -- takes the
        // general path below with the same s / d.
        const bool vote_ok = (p + 2 * gl + 1 < kDiscOut);
        const float dd = (s[0] < 0.0f) ? -d[0] : d[0];              // sync_update, m17_rx_sync.cpp:38-42
        int v = (dd > 0.0f) ? 1 : ((dd < 0.0f) ? -1 : 0);
        v = vote_ok ? v : 0;
        const int t = thr + group_scan_incl<LPC>(v);
        const bool general = (rem < LPC) | (m_idx < 0) | (t > thresh) | (t < -thresh);
        if (__builtin_amdgcn_ballot_w64(general) == 0ull) {
            my.h[8 + m_idx + gl] = s[0];
            m_idx += LPC;
            thr = group_bcast_last_i<LPC>(t);
            sum = group_bcast_last_f<LPC>(s[0]);
            dif = group_bcast_last_f<LPC>(d[0]);
            if (p + 2 * LPC - 1 < kDiscOut) { clk = 0; p += 2 * LPC; }
            else { clk = 1; p = kDiscOut; }                         // the last vote tick falls into the next block
            return;
        }
    }
    bool crossed = false;
    int ts = 0, kcross = 0;
#pragma unroll
    for (int j = 0; j < W; ++j) {
        const int nv = min(rem - LPC * j, LPC);                     // valid instants of this segment
        if (!crossed && nv > 0) {
            const int inst = gl + LPC * j;
            const bool vote_ok = (gl < nv) && (p + 2 * inst + 1 < kDiscOut);
            const float dd = (s[j] < 0.0f) ? -d[j] : d[j];
            const unsigned long long um = (__builtin_amdgcn_ballot_w64(vote_ok && dd > 0.0f) >> gshift) & Cfg::MASK;
            const unsigned long long dm = (__builtin_amdgcn_ballot_w64(vote_ok && dd < 0.0f) >> gshift) & Cfg::MASK;
            const int tk = thr + (int)__popcll(um & incl) - (int)__popcll(dm & incl);
            const unsigned long long cr =
                (__builtin_amdgcn_ballot_w64(vote_ok && (tk > thresh || tk < -thresh)) >> gshift) & Cfg::MASK;
            const int kl = cr ? (int)__ffsll((long long)cr) - 1 : 0;
            const int naccept = cr ? kl + 1 : nv;
            if (gl < naccept && (m_idx + gl) >= 0) my.h[8 + m_idx + gl] = s[j];
            m_idx += naccept;
            sum = __shfl(s[j], gbase + naccept - 1, 64);
            dif = __shfl(d[j], gbase + naccept - 1, 64);
            if (cr) { crossed = true; ts = __shfl(tk, gbase + kl, 64); kcross = LPC * j + kl; }
            else thr += (int)__popcll(um) - (int)__popcll(dm);
        }
    }
    if (crossed) {
        thr = 0; clk = 0;
        if (ts > thresh) {
            index = (index + 1 == kPhases) ? 0 : index + 1;
            if (index == 0) { clk = 1; if (m_idx >= 0 && gl == 0) my.h[8 + m_idx] = 0.0f; m_idx++; }
        } else {
            index = (index == 0) ? kPhases - 1 : index - 1;
            if (index == kPhases - 1) { clk = 1; m_idx--; }
        }
        p = p + 2 * kcross + 2;
    } else {
        const int ilast = p + 2 * (min(rem, W * LPC) - 1);
        if (ilast + 1 < kDiscOut) { clk = 0; p = ilast + 2; }
        else { clk = 1; p = kDiscOut; }
    }
}

template <int LPC>
__global__ __launch_bounds__(256)
void k_sync_frame_grp(const float *__restrict__ disc,     // [C][nblk][384]
                      const float *__restrict__ offs,     // [C][nblk] or null (already DC-free)
                      ChanState *__restrict__ st, int C, int nblk, int mode, int ext_lock,
                      m17gpu_rec_dev *__restrict__ recs, int rec_cap, int32_t *__restrict__ counts,
                      float *__restrict__ syms, int32_t *__restrict__ nsyms,
                      float *__restrict__ fsym, int b0, int bcount)
{
    using Cfg = GrpCfg<LPC>;
    __shared__ __attribute__((aligned(16))) float taps[kPhases * 64];      // (matched, derivative) pairs per branch
    __shared__ __attribute__((aligned(16))) GrpChan chs[Cfg::CPW];
    const int wave = (int)(threadIdx.x >> 6), lane = lane_id();
    const int grp = lane / LPC, gl = lane % LPC, gbase = lane - gl, gshift = grp * LPC;
    for (int q = (int)threadIdx.x; q < kPhases * 32; q += 256) {
        taps[2 * q] = (&c_tab.mf[0][0])[q];
        taps[2 * q + 1] = (&c_tab.md[0][0])[q];
    }
    __syncthreads();                                    // the only workgroup barrier
    const int chan = ((int)blockIdx.x * 4 + wave) * Cfg::G + grp;
    if (chan >= C) return;
    GrpChan &my = chs[wave * Cfg::G + grp];
    ChanState &cs = st[chan];
    m17gpu_rec_dev *crecs = recs ? recs + (size_t)chan * rec_cap : nullptr;
    if (!recs) rec_cap = 0;
    const unsigned long long incl = (gl == 63) ? ~0ull : ((2ull << gl) - 1ull);
    RegroupLane<LPC> rg;
    rg.load(gl);

    // group-uniform control state, one copy per lane
    int clk = cs.clk, thr = cs.thr, index = cs.index;
    float sum = cs.sum, dif = cs.dif;
    int flock = cs.flock, fclk = cs.fclk, ferr = cs.ferr;
    uint32_t block_count = cs.block_count;
    int nrec = (b0 == 0) ? 0 : counts[chan];
    int sym_total = (b0 == 0) ? 0 : cs.sym_total;
    for (int q = gl; q < kTaps - 1; q += LPC) my.x[q] = cs.buff[q + 1];
    if (gl < 8) my.h[gl] = cs.sync[gl];
    for (int q = gl; q < kFrameSyms; q += LPC) my.f[q] = cs.fsym[q];
    const size_t sym_base = (size_t)chan * M17_SYM_STRIDE(nblk);
    const float *dsrc = disc + (size_t)chan * nblk * kDiscOut;
    const float *osrc = offs ? offs + (size_t)chan * nblk : nullptr;
    {
        const float off = osrc ? osrc[b0] : 0.0f;
        for (int q = gl; q < kDiscOut; q += LPC) {
            float v = dsrc[(size_t)b0 * kDiscOut + q];
            if (osrc) v = v - off;                                   // out[i] - offset (m17_dsp.cpp:217-219)
            my.x[kTaps - 1 + q] = v;
        }
    }
    wave_fence();

    const int bend = b0 + bcount;
#ifdef M17_STAMPS
    unsigned long long acc_[12] = {0,0,0,0,0,0,0,0,0,0,0,0}, last_ = __builtin_amdgcn_s_memtime();
    __builtin_amdgcn_s_waitcnt(0xC07F);
#define GCNT(i) acc_[i] += 1
#else
#define GCNT(i) do {} while (0)
#endif
    for (int b = b0; b < bend; ++b) {
        STAMP(5);
        // ---- timing recovery in rounds of LPC instants; x[i .. i+30] is the delay line at input i
        const int lockv = (ext_lock >= 0) ? ext_lock : flock;
        const int thresh = lockv ? 80 : 10;
        int p = 0, m_idx = 0;
        {
            float4 tp[16];                                            // 32 tap pairs of the current branch
            int tap_index = -1;
            while (p < kDiscOut) {
                if (clk == 1) {
                    GCNT(9);
                    // vote tick on the carried sum/dif (sync_update :38-42, m17_sync_adjust :45-72)
                    clk = 0;
                    const float d0 = (sum < 0.0f) ? -dif : dif;
                    if (d0 > 0.0f) thr++;
                    if (d0 < 0.0f) thr--;
                    if (thr > thresh) {
                        index = (index + 1 == kPhases) ? 0 : index + 1; thr = 0;
                        if (index == 0) { clk = 1; if (m_idx >= 0 && gl == 0) my.h[8 + m_idx] = 0.0f; m_idx++; }
                    }
                    if (thr < -thresh) {
                        thr = 0; index = (index == 0) ? kPhases - 1 : index - 1;
                        if (index == kPhases - 1) { clk = 1; m_idx--; }
                    }
                    p++;
                    STAMP(0);
                    continue;
                }
                GCNT(8);
                if (tap_index != index) {
                    const float4 *t4 = reinterpret_cast<const float4 *>(&taps[64 * index]);
#pragma unroll
                    for (int q = 0; q < 16; ++q) tp[q] = t4[q];
                    tap_index = index;
                }
                STAMP(0);
                // one pass = one round: lane gl takes the filter tick at input p + 2 gl and the vote tick
                // after it.  (Passes of three rounds with three FIR chains per lane were measured while
                // locked: same time at 64 lanes per channel -- a lone wave is issue-bound, not latency-
                // bound -- and 15% slower at 32 / 16 lanes from the extra registers.  W stays 1.)
                const int rem = (kDiscOut - p + 1) >> 1;              // filter instants left in the block
                timing_pass<LPC, 1>(my, tp, gl, gbase, gshift, incl, thresh, rem, p, m_idx, clk, thr, index, sum, dif);
                STAMP(2);
            }
        }
        const int n = m_idx > 0 ? m_idx : 0;
        wave_fence();

        // next block's input: loads issued now, committed after the framer
        constexpr int PF = kDiscOut / LPC;
        float pf[PF];
        float noff = 0.0f;
        if (b + 1 < bend) {
            const float *nx = dsrc + (size_t)(b + 1) * kDiscOut;
            noff = osrc ? osrc[b + 1] : 0.0f;
#pragma unroll
            for (int r = 0; r < PF; ++r) pf[r] = __builtin_nontemporal_load(&nx[gl + LPC * r]);   // read once
        }

        // symbols out (optional)
        if (syms) {
            // at most 193 symbols: a fixed number of predicated stores (the generic strided loop compiled to
            // 169 instructions of 64-bit index arithmetic)
            float *so = syms + sym_base + sym_total;
#pragma unroll
            for (int r = 0; r < (193 + LPC - 1) / LPC; ++r) {
                const int q = gl + LPC * r;
                if (q < n) __builtin_nontemporal_store(my.h[8 + q], &so[q]);     // output, not read again here
            }
        }
        if (nsyms && gl == 0) nsyms[(size_t)chan * nblk + b] = n;
        sym_total += n;

        STAMP(3);
        // ---- framer (m17_rx_frame.cpp:126-177)
        int pos = (ext_lock >= 0) ? n : 0;
        while (pos < n) {
            GCNT(10);
            if (flock) {
                const int cnt = min(kFrameSyms - fclk, n - pos);
                {
                    // reads batched ahead of the writes: the strided loop paid one LDS round trip per element
                    constexpr int R = kFrameSyms / LPC;
                    float t[R];
#pragma unroll
                    for (int r = 0; r < R; ++r) t[r] = (gl + LPC * r < cnt) ? my.h[8 + pos + gl + LPC * r] : 0.0f;
#pragma unroll
                    for (int r = 0; r < R; ++r) if (gl + LPC * r < cnt) my.f[fclk + gl + LPC * r] = t[r];
                }
                fclk += cnt; pos += cnt;
                wave_fence();
                if (fclk == kFrameSyms) {
                    fclk = 0;
                    float v[8];
#pragma unroll
                    for (int i = 0; i < 8; ++i) v[i] = my.f[i];
                    const SyncResult r = sync_check_grp<LPC>(v, gl, gbase, gshift);
                    uint32_t flags = 0;
                    bool parse = false, unlock = false;
                    if (r.type == 5) { flags |= M17_F_EOT; unlock = true; }
                    else if (sync_accept(r, true)) { flags |= M17_F_SYNC_OK; parse = true; ferr = 0; }
                    else {
                        ferr++;
                        if (ferr > 5) { flags |= M17_F_LOST; unlock = true; }
                        else parse = true;
                    }
                    if (parse && mode == 1) flags |= M17_F_PARSED;
                    const uint32_t w0 = (uint32_t)r.type | ((uint32_t)r.votes << 8) | ((uint32_t)(ferr & 0xFF) << 24);
                    emit_record_grp(crecs, rec_cap, nrec, gl, w0, flags, r.variance, block_count, (uint32_t)(pos - 1));
                    if ((flags & M17_F_PARSED) && nrec < rec_cap && r.type >= 1 && r.type <= 3) {
                        float *fd = fsym + ((size_t)chan * rec_cap + nrec) * kSlotFloats;
                        store_frame_slot<LPC>(fd, r.type, gl, rg, [&](int q) { return my.f[q]; });
                    }
                    nrec++;
                    if (unlock) {
                        flock = 0;
                        // reset_sync(): the next hunt windows must see zeros behind them
                        if (gl < 8) { my.h[pos + gl] = 0.0f; cs.sync[gl] = 0.0f; }
                        wave_fence();
                    }
                }
            } else {
                // hunt: candidate j = pos+gl, window = m_sync after shifting symbol j in
                const int jc = pos + gl;
                const bool cand = jc < n;
                const int jj = cand ? jc : pos;
                float v[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) v[i] = my.h[jj + 1 + i];
                unsigned long long hm = 0;
                SyncResult r; r.type = 0; r.votes = 8; r.variance = 1.0f;
                if (cand && hunt_compatible(v)) r = sync_check(v);           // most windows are rejected by sign
                hm = (__builtin_amdgcn_ballot_w64(cand && sync_accept(r, false)) >> gshift) & Cfg::MASK;
                if (hm) {
                    const int l = (int)__ffsll((long long)hm) - 1;
                    const int js = pos + l;
                    // copy_sync(); m_fclk = 8; lock; m17_aos()
                    float wv = 0.0f;
                    if (gl < 8) wv = my.h[js + 1 + gl];
                    wave_fence();
                    if (gl < 8) { my.f[gl] = wv; cs.sync[gl] = wv; }
                    fclk = 8; ferr = 0; flock = 1;
                    const int ty = __shfl(r.type, gbase + l, 64), vo = __shfl(r.votes, gbase + l, 64);
                    const float va = __shfl(r.variance, gbase + l, 64);
                    emit_record_grp(crecs, rec_cap, nrec, gl, (uint32_t)ty | ((uint32_t)vo << 8), M17_F_AOS, va,
                                    block_count, (uint32_t)js);
                    nrec++;
                    pos = js + 1;
                    wave_fence();
                } else {
                    pos = min(n, pos + LPC);
                }
            }
        }
        STAMP(4);
        // m_sync for the next block while hunting: last 8 entries of h; delay line: last 30
        // inputs; then the prefetched block moves in
        {
            float keep_h = 0.0f;
            if (gl < 8) keep_h = my.h[n + gl];
            float keep_x[(kTaps - 1 + LPC - 1) / LPC];
#pragma unroll
            for (int r = 0; r < (kTaps - 1 + LPC - 1) / LPC; ++r)
                keep_x[r] = (gl + LPC * r < kTaps - 1) ? my.x[kDiscOut + gl + LPC * r] : 0.0f;
            wave_fence();
            // (the lock-forced stage entry m17gpu_sync_samples has no framer: hunt window, block counter
            //  and record count of the context stay as they were)
            if (!flock && gl < 8 && ext_lock < 0) { my.h[gl] = keep_h; cs.sync[gl] = keep_h; }
#pragma unroll
            for (int r = 0; r < (kTaps - 1 + LPC - 1) / LPC; ++r)
                if (gl + LPC * r < kTaps - 1) my.x[gl + LPC * r] = keep_x[r];
            if (b + 1 < bend) {
#pragma unroll
                for (int r = 0; r < PF; ++r)
                    my.x[kTaps - 1 + gl + LPC * r] = osrc ? (pf[r] - noff) : pf[r];     // out[i] - offset
            }
        }
        if (ext_lock < 0) block_count++;
        wave_fence();
    }

#ifdef M17_STAMPS
    if (chan == 0 && gl == 0) for (int i = 0; i < 12; ++i) g_stamps[i] = acc_[i];
    if (chan < 4096 && gl == 0) { for (int i = 0; i < 6; ++i) g_chan_stamps[chan][i] = acc_[i]; g_chan_stamps[chan][6] = acc_[8]; g_chan_stamps[chan][7] = acc_[10]; }
#endif
    // ---- store state
    if (gl == 0) {
        cs.clk = clk; cs.thr = thr; cs.index = index; cs.sum = sum; cs.dif = dif;
        cs.flock = flock; cs.fclk = fclk; cs.ferr = ferr; cs.block_count = block_count;
        cs.buff[0] = 0.0f;
        if (ext_lock < 0) { cs.sym_total = sym_total; if (counts) counts[chan] = nrec; }
    }
    for (int q = gl; q < kTaps - 1; q += LPC) cs.buff[q + 1] = my.x[q];
    for (int q = gl; q < kFrameSyms; q += LPC) cs.fsym[q] = my.f[q];
}

} // namespace m17dev
