// m17_sync_wg.hip -- k_sync_frame_wg: timing recovery + sync correlator + framer
// with one 256-thread WORKGROUP per channel (included by m17gpu_capi.hip after
// m17_kernels.hip; same namespace, same helpers).
//
// Reference: m17_rx_sync_samples (m17_rx_sync.cpp:77-99) and m17_rx_sym
// (m17_rx_frame.cpp:126-177).  The time axis of one channel is sequential only
// through a handful of integers (m_clk, m_thr, m_index, lock state); the
// arithmetic -- two 31-tap dot products per symbol instant -- depends on them
// only through the polyphase branch m_index.  So for every 1920-sample block
// all (up to 192) symbol instants are evaluated at once under the current
// branch, one instant per thread, each thread keeping the reference's ascending
// mul/add order; the early/late vote counter then becomes a popcount prefix over
// four wave ballots and the first threshold crossing (rare once locked) cuts
// the block and re-runs the remainder under the new branch.  While hunting
// (threshold 10, a crossing every >= 11 instants) only one wave speculates.
//
// One workgroup barrier per pass: the ballots / filter outputs are double
// buffered by pass parity and every wave scans all four ballots redundantly.
//
// LDS: both tap tables (2 x 40 x 32 floats), the block's input with its
// 30-sample delay line twice (second copy shifted by one float so that every
// (x[a], x[a+1]) pair is an aligned ds_read_b64 whatever the parity of a), the
// symbol/frame buffers, and the cross-wave ballots.
#pragma clang fp contract(off)

namespace m17dev {

constexpr int WG_T = 128;                                // threads per channel (two waves)
constexpr int NW   = WG_T / 64;
constexpr int PF_N = kDiscOut * 4 / WG_T;                // staged inputs per thread (12)

constexpr int WIN  = 4;                                   // blocks staged in LDS at a time
constexpr int XLEN = kTaps - 1 + kDiscOut * WIN + 2;

struct WgShared {
    float mf[kPhases][32];
    float md[kPhases][32];
    float xa[XLEN];                        // xa[i] = x[i]   (x[0..29] history, then WIN blocks of 384 inputs)
    float xb[XLEN];                        // xb[i] = x[i+1]
    float sums[2][WG_T], difs[2][WG_T];
    float h[8 + kFrameSyms * WIN + 16];    // m_sync (8) followed by the symbols of the block / window
    float f[kFrameSyms];                   // m_f_sym
    unsigned long long up[2][NW], dn[2][NW], hit[NW];
    unsigned long long fup[3 * 4], fdn[3 * 4];             // fast window: ballots of 12 x 64 instants
    float bc_var; int bc_type, bc_votes;
    float fr_var[WIN]; int fr_type[WIN], fr_votes[WIN];    // fast window: sync class of each frame
    float last_sum, last_dif; int fast_abort;
    float poff[WIN];                       // DC offsets of the staged blocks
};

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

__device__ __forceinline__ int first_set4(const unsigned long long m[NW])
{
    int k = -1;
#pragma unroll
    for (int w = NW - 1; w >= 0; --w)
        if (m[w]) k = 64 * w + __ffsll((long long)m[w]) - 1;
    return k;
}

// m17_sync_check (m17_rx_frame.cpp:47-81) on ONE 8-symbol vector that every lane
// of the calling wave holds: lane k < 6 accumulates template k, the in-order
// strict-'>' argmax walks the six lanes with readlane, the votes are a ballot.
__device__ __forceinline__ SyncResult sync_check_wave(const float v[8])
{
    constexpr unsigned negs[6] = {0xAA, 0xB0, 0x4F, 0xF2, 0x0D, 0x40};
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

// One symbol instant: the matched (s) and derivative (d) 31-tap dot products over the
// delay line xs[0..30], strictly in the reference's order (rx_sync_filter,
// m17_rx_sync.cpp:25-31: bare first product, then += in ascending tap order, separate
// multiply and add).  31 taps in four groups of 8 (last: 7); the next group's LDS reads
// (taps are wave-uniform broadcasts) are issued before the current group's chain.
typedef float v2f __attribute__((ext_vector_type(2)));

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

__global__ __launch_bounds__(WG_T, 2)     // 1,024 channels x 2 waves = 2 waves per SIMD: <= 256 VGPRs, no spills
void k_sync_frame_wg(const float *__restrict__ disc,     // [C][nblk][384]
                     const float *__restrict__ offs,     // [C][nblk] or null (already DC-free)
                     ChanState *__restrict__ st, int C, int nblk, int mode, int ext_lock,
                     m17gpu_rec_dev *__restrict__ recs, int rec_cap, int32_t *__restrict__ counts,
                     float *__restrict__ syms, int32_t *__restrict__ nsyms,
                     float *__restrict__ fsym, int32_t *__restrict__ work, int32_t *__restrict__ nwork,
                     int allow_fast)
{
    __shared__ __attribute__((aligned(16))) WgShared sh;
    const int t = (int)threadIdx.x, w = t >> 6, lane = t & 63;
    const int chan = (int)blockIdx.x;
    ChanState &cs = st[chan];
    m17gpu_rec_dev *crecs = recs ? recs + (size_t)chan * rec_cap : nullptr;

    // ---- tables and state into LDS / uniform registers
    for (int q = t; q < kPhases * 32; q += WG_T) {
        // taps interleaved as (matched, derivative) pairs: row idx = 64 floats starting at &sh.mf[0][0] + 64 idx
        (&sh.mf[0][0])[2 * q] = (&c_tab.mf[0][0])[q];
        (&sh.mf[0][0])[2 * q + 1] = (&c_tab.md[0][0])[q];
    }
    int clk = uni(cs.clk), thr = uni(cs.thr), index = uni(cs.index);
    float sum = unif(cs.sum), dif = unif(cs.dif);
    int flock = uni(cs.flock), fclk = uni(cs.fclk), ferr = uni(cs.ferr);
    uint32_t block_count = (uint32_t)uni((int)cs.block_count);
    if (t < kTaps - 1) {
        const float v = cs.buff[t + 1];
        sh.xa[t] = v;
        if (t >= 1) sh.xb[t - 1] = v;
    }
    if (t < 8) sh.h[t] = cs.sync[t];
    for (int q = t; q < kFrameSyms; q += WG_T) sh.f[q] = cs.fsym[q];
    if (t == 0) sh.fast_abort = 0;
    int nrec = 0, sym_total = 0, par = 0;
    const size_t sym_base = (size_t)chan * M17_SYM_STRIDE(nblk);
    const float *dsrc = disc + (size_t)chan * nblk * kDiscOut;
    const float *osrc = offs ? offs + (size_t)chan * nblk : nullptr;

    // ---- staging of up to WIN blocks: 6 inputs per thread, prefetched one window ahead.
    // Element r of thread t is input (t + 256 r) of the window: block (t+256r)/384.
    float pf[PF_N], po1 = 0.0f;                 // po1: DC offset of block (t & 3) of the prefetched window
// (to_ is an opaque copy of t: keeps the compiler from hoisting ~30 loop-invariant
//  addresses out of the block loop and spilling them)
#define SF_PREFETCH(WB) { int to_ = t; asm volatile("" : "+v"(to_));                 \
    _Pragma("unroll") for (int r = 0; r < PF_N; ++r) {                                  \
        const int e_ = to_ + WG_T * r, jj_ = e_ / kDiscOut;                          \
        pf[r] = (((WB) + jj_) < nblk) ? dsrc[(size_t)(WB) * kDiscOut + e_] : 0.0f;   \
    }                                                                                \
    po1 = (osrc && t < WIN && ((WB) + t) < nblk) ? osrc[(WB) + t] : 0.0f; }
#define SF_POFF()   if (t < WIN) sh.poff[t] = po1;       /* before the barrier that precedes SF_COMMIT */
#define SF_COMMIT() { int to_ = t; asm volatile("" : "+v"(to_)); /* out[i] - offset (m17_dsp.cpp:217-219) */ \
    _Pragma("unroll") for (int r = 0; r < PF_N; ++r) {                                  \
        const int e_ = to_ + WG_T * r;                                               \
        const float v_ = osrc ? (pf[r] - sh.poff[e_ / kDiscOut]) : pf[r];            \
        sh.xa[kTaps - 1 + e_] = v_;                                                  \
        sh.xb[kTaps - 2 + e_] = v_;                                                  \
    } }
    SF_PREFETCH(0)
    SF_POFF()
    lds_barrier();
    SF_COMMIT()
    int wb = 0, staged = min(WIN, nblk);
    if (staged < nblk) { SF_PREFETCH(WIN) }
    lds_barrier();

#ifdef M17_STAMPS
    unsigned long long acc_[12] = {0,0,0,0,0,0,0,0,0,0,0,0}, last_ = __builtin_amdgcn_s_memtime();
    __builtin_amdgcn_s_waitcnt(0xC07F);
#endif
    int b = 0;
    while (b < nblk) {
        STAMP(0);
        if (b == wb + staged) {
            // restage: the last 30 inputs become the delay line, the prefetched window moves in
            float keep_x = 0.0f;
            if (t < kTaps - 1) keep_x = sh.xa[kDiscOut * staged + t];
            SF_POFF()
            lds_barrier();
            if (t < kTaps - 1) {
                sh.xa[t] = keep_x;
                if (t >= 1) sh.xb[t - 1] = keep_x;
            }
            SF_COMMIT()
            wb = b; staged = min(WIN, nblk - b);
            if (wb + WIN < nblk) { SF_PREFETCH(wb + WIN) }
            lds_barrier();
        }
        const int j = b - wb;
        const int xoff = kDiscOut * j;
        const int brem = staged - j;

        // =========================== fast window ===========================
        if (allow_fast && ext_lock < 0 && flock && brem >= 2 && fclk >= 8) {
            const int NT = kFrameSyms * brem;            // filter instants in the window
            int thr0 = thr, p0 = 0;
            bool go = true;
            if (clk == 1) {                              // carried vote tick, tentatively
                const float d0 = (sum < 0.0f) ? -dif : dif;
                if (d0 > 0.0f) thr0++;
                if (d0 < 0.0f) thr0--;
                p0 = 1;
                go = !(thr0 > 80 || thr0 < -80);
            }
            DBGCNT(0);
            if (go) {
                const float4 *fmf4 = reinterpret_cast<const float4 *>(&sh.mf[0][0] + 64 * index);
                const float4 *fmd4 = reinterpret_cast<const float4 *>(sh.md[index]);
#pragma unroll 3          // three instants in flight per thread: six independent add chains hide the VALU latency
                for (int r = 0; r < 768 / WG_T; ++r) {
                    const int k = t + WG_T * r;
                    if (WG_T * r < NT) {                                  // uniform
                        const bool have = k < NT;
                        const int ik = p0 + 2 * (have ? k : 0);
                        const int a = xoff + ik;
                        const float *xs = (a & 1) ? (sh.xb + (a - 1)) : (sh.xa + a);
                        float s, d;
                        fir_instant(xs, fmf4, fmd4, s, d);
                        if (have) sh.h[8 + k] = s;
                        if (k == NT - 1) { sh.last_sum = s; sh.last_dif = d; }
                        const bool vote_ok = have && (p0 + 2 * k + 1 < kDiscOut * brem);
                        const float dd = (s < 0.0f) ? -d : d;
                        const unsigned long long upm = __ballot(vote_ok && dd > 0.0f);
                        const unsigned long long dnm = __ballot(vote_ok && dd < 0.0f);
                        if (lane == 0) { sh.fup[r * NW + w] = upm; sh.fdn[r * NW + w] = dnm; }
                    } else if (lane == 0) { sh.fup[r * NW + w] = 0; sh.fdn[r * NW + w] = 0; }
                }
                STAMP(8);
                lds_barrier();
                // scan: segment g = r*NW + w' holds instants 64 g .. 64 g + 63; wave w checks 12/NW of the 12
                int tot_all = 0;
                {
                    const unsigned long long incl = (lane == 63) ? ~0ull : ((2ull << lane) - 1ull);
                    int running = thr0;
                    bool crossed = false;
#pragma unroll
                    for (int g = 0; g < 12; ++g) {
                        const unsigned long long Ug = uni64(sh.fup[g]), Dg = uni64(sh.fdn[g]);
                        if (g >= (12 / NW) * w && g < (12 / NW) * (w + 1)) {
                            const int tk = running + __popcll(Ug & incl) - __popcll(Dg & incl);
                            crossed = crossed || (tk > 80 || tk < -80);
                        }
                        running += (int)__popcll(Ug) - (int)__popcll(Dg);
                    }
                    tot_all = running;
                    if (__ballot(crossed) != 0ull && lane == 0) sh.fast_abort = 1;
                }
                STAMP(9);
                // frames: wave w classifies frames w, w+NW, ... by their leading sync words
                for (int fq = w; fq < brem; fq += NW) {
                    float v[8];
                    const int first = (fq == 0) ? 0 : (8 + (kFrameSyms - fclk) + kFrameSyms * (fq - 1));
#pragma unroll
                    for (int i = 0; i < 8; ++i) v[i] = (fq == 0) ? sh.f[i] : sh.h[first + i];
                    const SyncResult r0 = sync_check_wave(v);
                    const bool clean = r0.type >= 1 && r0.type <= 4 && sync_accept(r0, true);
                    if (lane == 0) {
                        sh.fr_type[fq] = r0.type; sh.fr_votes[fq] = r0.votes; sh.fr_var[fq] = r0.variance;
                        if (!clean) sh.fast_abort = 1;
                    }
                }
                lds_barrier();
                STAMP(10);
                const int aborted = uni(sh.fast_abort);
                if (!aborted) {
                    DBGCNT(1);
                    // ---- commit the window
                    thr = tot_all; clk = (p0 == 0) ? 0 : 1;
                    sum = unif(sh.last_sum); dif = unif(sh.last_dif);
                    ferr = 0;
#pragma unroll
                    for (int r = 0; r < 768 / WG_T; ++r) {
                        const int k = t + WG_T * r;
                        if (syms && k < NT) syms[sym_base + sym_total + k] = sh.h[8 + k];
                    }
                    if (nsyms && t < brem) nsyms[(size_t)chan * nblk + b + t] = kFrameSyms;
                    const uint32_t fl = M17_F_SYNC_OK | ((mode == 1) ? M17_F_PARSED : 0u);
                    if (w == 0) {
                        for (int fq = 0; fq < brem; ++fq) {
                            const int cpos = (kFrameSyms - 1 - fclk) + kFrameSyms * fq;    // completing symbol
                            emit_record(crecs, rec_cap, nrec + fq,
                                        (uint32_t)uni(sh.fr_type[fq]) | ((uint32_t)uni(sh.fr_votes[fq]) << 8), fl,
                                        unif(sh.fr_var[fq]), block_count + (uint32_t)(cpos / kFrameSyms),
                                        (uint32_t)(cpos % kFrameSyms));
                        }
                    }
                    if (mode == 1) {
                        int wbase = 0;
                        if (work && t == 0) {
                            int nq = 0;
                            for (int q = 0; q < brem; ++q) {
                                const int ty = sh.fr_type[q];
                                if (nrec + q < rec_cap && ty >= 1 && ty <= 3) nq++;
                            }
                            wbase = nq ? atomicAdd(nwork, nq) : 0;
                            int at = 0;
                            for (int q = 0; q < brem; ++q) {
                                const int ty = sh.fr_type[q];
                                if (nrec + q < rec_cap && ty >= 1 && ty <= 3) work[wbase + at++] = chan * rec_cap + nrec + q;
                            }
                        }
                        for (int q = 0; q < brem; ++q) {
                            const int ty = uni(sh.fr_type[q]);
                            if (nrec + q < rec_cap && ty >= 1 && ty <= 3) {
                                float *fd = fsym + ((size_t)chan * rec_cap + nrec + q) * kFrameSyms;
                                for (int u = t; u < kFrameSyms; u += WG_T)
                                    fd[u] = (q == 0) ? ((u < fclk) ? sh.f[u] : sh.h[8 + u - fclk])
                                                     : sh.h[8 + (kFrameSyms - fclk) + kFrameSyms * (q - 1) + u];
                            }
                        }
                    }
                    lds_barrier();                              // frame 0 has read f
                    for (int u = t; u < fclk; u += WG_T) sh.f[u] = sh.h[8 + (kFrameSyms - fclk) + kFrameSyms * (brem - 1) + u];
                    nrec += brem; block_count += (uint32_t)brem; sym_total += NT; b += brem;
                    lds_barrier();
                    STAMP(11);
                    continue;
                }
                if (t == 0) sh.fast_abort = 0;
                lds_barrier();
            }
        }

        DBGCNT(2);
        // =========================== exact per-block path ===========================
        // ---- timing recovery: x[i .. i+30] is the delay line at input i; symbols go to h[8+..]
        // ext_lock >= 0: timing recovery alone, lock flag supplied by the caller's framer (m17_rx_sync.cpp:92-95)
        const int lockv = (ext_lock >= 0) ? ext_lock : flock;
        const int thresh = lockv ? 80 : 10;
        const int width = lockv ? WG_T : 64;
        int p = 0, m_idx = 0;
        while (p < kDiscOut) {
            p = uni(p); m_idx = uni(m_idx); thr = uni(thr); index = uni(index); clk = uni(clk);
            if (clk == 1) {
                // vote tick on the carried sum/dif (sync_update :38-42, m17_sync_adjust :45-72)
                clk = 0;
                const float d = (sum < 0.0f) ? -dif : dif;
                if (d > 0.0f) thr++;
                if (d < 0.0f) thr--;
                if (thr > thresh) {
                    index = (index + 1) % kPhases; thr = 0;
                    if (index == 0) { clk = 1; if (m_idx >= 0 && t == 0) sh.h[8 + m_idx] = 0.0f; m_idx++; }
                }
                if (thr < -thresh) {
                    thr = 0; index = (index + kPhases - 1) % kPhases;
                    if (index == kPhases - 1) { clk = 1; m_idx--; }
                }
                p++;
                continue;
            }
            // one pass: thread k = filter tick at input p+2k and the vote tick after it
            const int nf = min(width, (kDiscOut - p + 1) >> 1);
            const int ik = p + 2 * t;
            const bool have = t < nf;
            float s = 0.0f, d = 0.0f;
            if (w * 64 < nf) {                                  // wave-uniform: this wave has instants to evaluate
                const int a = xoff + (have ? ik : p);
                const float *xs = (a & 1) ? (sh.xb + (a - 1)) : (sh.xa + a);
                fir_instant(xs, reinterpret_cast<const float4 *>(&sh.mf[0][0] + 64 * index),
                            reinterpret_cast<const float4 *>(sh.md[index]), s, d);
            }
            STAMP(1);
            sh.sums[par][t] = s; sh.difs[par][t] = d;
            const bool vote_ok = have && (ik + 1 < kDiscOut);
            const float dd = (s < 0.0f) ? -d : d;
            const unsigned long long upm = __ballot(vote_ok && dd > 0.0f);
            const unsigned long long dnm = __ballot(vote_ok && dd < 0.0f);
            if (lane == 0) { sh.up[par][w] = upm; sh.dn[par][w] = dnm; }
            lds_barrier();
            STAMP(2);
            // every wave scans all four ballots: lane = position inside a 64-instant segment
            const unsigned long long incl = (lane == 63) ? ~0ull : ((2ull << lane) - 1ull);
            int running = thr, kstar = -1, ts = 0;
#pragma unroll
            for (int q = 0; q < NW; ++q) {
                const unsigned long long Uq = uni64(sh.up[par][q]), Dq = uni64(sh.dn[par][q]);
                if (q * 64 < nf) {
                    const int tk = running + __popcll(Uq & incl) - __popcll(Dq & incl);
                    const int idx = q * 64 + lane;
                    const bool ok = (idx < nf) && (p + 2 * idx + 1 < kDiscOut);
                    const unsigned long long cr = __ballot(ok && (tk > thresh || tk < -thresh));
                    if (kstar < 0 && cr) {
                        const int kl = __ffsll((long long)cr) - 1;
                        kstar = q * 64 + kl;
                        ts = bcast_lane_i(tk, kl);
                    }
                    running += (int)__popcll(Uq) - (int)__popcll(Dq);
                }
            }
            const int naccept = (kstar >= 0) ? kstar + 1 : nf;
            if (t < naccept && (m_idx + t) >= 0) sh.h[8 + m_idx + t] = s;
            m_idx += naccept;
            sum = unif(sh.sums[par][naccept - 1]);
            dif = unif(sh.difs[par][naccept - 1]);
            if (kstar >= 0) {
                thr = 0; clk = 0;
                if (ts > thresh) {
                    index = (index + 1) % kPhases;
                    if (index == 0) { clk = 1; if (m_idx >= 0 && t == 0) sh.h[8 + m_idx] = 0.0f; m_idx++; }
                } else {
                    index = (index + kPhases - 1) % kPhases;
                    if (index == kPhases - 1) { clk = 1; m_idx--; }
                }
                p = p + 2 * kstar + 2;
            } else {
                thr = running;
                const int ilast = p + 2 * (nf - 1);
                if (ilast + 1 < kDiscOut) { clk = 0; p = ilast + 2; }
                else { clk = 1; p = kDiscOut; }
            }
            par ^= 1;
            STAMP(3);
        }
        const int n = m_idx > 0 ? m_idx : 0;
        lds_barrier();
        STAMP(4);

        // symbols out (optional)
        if (syms) for (int q = t; q < n; q += WG_T) syms[sym_base + sym_total + q] = sh.h[8 + q];
        if (nsyms && t == 0) nsyms[(size_t)chan * nblk + b] = n;
        sym_total += n;

        // ---- framer (m17_rx_frame.cpp:126-177)
        int pos = (ext_lock >= 0) ? n : 0;
        while (pos < n) {
            if (flock) {
                const int cnt = min(kFrameSyms - fclk, n - pos);
                for (int q = t; q < cnt; q += WG_T) sh.f[fclk + q] = sh.h[8 + pos + q];
                fclk += cnt; pos += cnt;
                if (fclk == kFrameSyms) {
                    fclk = 0;
                    lds_barrier();
                    // classify the frame by its own leading sync word: one wave computes, all read
                    if (w == 0) {
                        float v[8];
#pragma unroll
                        for (int i = 0; i < 8; ++i) v[i] = sh.f[i];
                        const SyncResult r0 = sync_check_wave(v);
                        if (lane == 0) { sh.bc_type = r0.type; sh.bc_votes = r0.votes; sh.bc_var = r0.variance; }
                    }
                    lds_barrier();
                    SyncResult r;
                    r.type = uni(sh.bc_type); r.votes = uni(sh.bc_votes); r.variance = unif(sh.bc_var);
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
                    if (w == 0)
                        emit_record(crecs, rec_cap, nrec, w0, flags, r.variance, block_count, (uint32_t)(pos - 1));
                    if ((flags & M17_F_PARSED) && nrec < rec_cap && r.type >= 1 && r.type <= 3) {
                        float *fd = fsym + ((size_t)chan * rec_cap + nrec) * kFrameSyms;
                        for (int q = t; q < kFrameSyms; q += WG_T) fd[q] = sh.f[q];
                        if (work && t == 0) work[atomicAdd(nwork, 1)] = chan * rec_cap + nrec;
                    }
                    nrec++;
                    if (unlock) {
                        flock = 0;
                        // reset_sync(): the next hunt windows must see zeros behind them
                        if (t < 8) { sh.h[pos + t] = 0.0f; cs.sync[t] = 0.0f; }
                    }
                    lds_barrier();
                }
            } else {
                // hunt: candidate j = pos+t, window = m_sync after shifting symbol j in
                const int jc = pos + t;
                const bool cand = jc < n;
                const int jj = cand ? jc : pos;
                float v[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) v[i] = sh.h[jj + 1 + i];
                const SyncResult r = sync_check(v);
                const unsigned long long hm = __ballot(cand && sync_accept(r, false));
                if (lane == 0) sh.hit[w] = hm;
                lds_barrier();
                unsigned long long H[NW];
#pragma unroll
                for (int q = 0; q < NW; ++q) H[q] = uni64(sh.hit[q]);
                const int l = first_set4(H);
                if (l >= 0) {
                    const int js = pos + l;
                    if (t == l) { sh.bc_type = r.type; sh.bc_votes = r.votes; sh.bc_var = r.variance; }
                    // copy_sync(); m_fclk = 8; lock; m17_aos()
                    float wv = 0.0f;
                    if (t < 8) wv = sh.h[js + 1 + t];
                    lds_barrier();
                    if (t < 8) { sh.f[t] = wv; cs.sync[t] = wv; }
                    fclk = 8; ferr = 0; flock = 1;
                    if (w == 0)
                        emit_record(crecs, rec_cap, nrec, (uint32_t)uni(sh.bc_type) | ((uint32_t)uni(sh.bc_votes) << 8),
                                    M17_F_AOS, unif(sh.bc_var), block_count, (uint32_t)js);
                    nrec++;
                    pos = js + 1;
                } else {
                    pos = min(n, pos + WG_T);
                }
                lds_barrier();
            }
        }
        STAMP(5);
        // m_sync for the next block while hunting: last 8 entries of h
        {
            float keep_h = 0.0f;
            if (t < 8) keep_h = sh.h[n + t];
            lds_barrier();
            if (!flock && t < 8) { sh.h[t] = keep_h; cs.sync[t] = keep_h; }
        }
        block_count++;
        b++;
        lds_barrier();
        STAMP(6);
    }
#ifdef M17_STAMPS
    if (chan == 0 && t == 0) for (int i = 0; i < 12; ++i) g_stamps[i] = acc_[i];
#endif

    // ---- store state: the last 30 inputs of the staged buffer are the delay line
    if (t == 0) {
        cs.clk = clk; cs.thr = thr; cs.index = index; cs.sum = sum; cs.dif = dif;
        cs.flock = flock; cs.fclk = fclk; cs.ferr = ferr; cs.block_count = block_count;
        cs.buff[0] = 0.0f;
        if (counts) counts[chan] = nrec;
    }
    if (t < kTaps - 1) cs.buff[t + 1] = sh.xa[kDiscOut * staged + t];
    for (int q = t; q < kFrameSyms; q += WG_T) cs.fsym[q] = sh.f[q];
}

} // namespace m17dev
