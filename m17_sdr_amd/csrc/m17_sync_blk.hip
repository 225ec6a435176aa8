// m17_sync_blk.hip -- k_sync_frame_blk: timing recovery + sync correlator + framer, one wave per channel, a whole
// 1920-sample block per pass.
//
// Reference: m17_rx_sync_samples (m17_rx_sync.cpp:77-99), m17_rx_sym (m17_rx_frame.cpp:126-177).
//
// While a channel tracks, the timing loop's state hardly moves: the polyphase branch steps about once in ten
// blocks, the tick parity only when the branch wraps, and the matched / derivative filter outputs -- all of the
// arithmetic -- depend on branch and parity alone, not on the vote counter or the lock flag.  So a block is
// first taken AS A WHOLE under the branch and parity it starts with:
//   * all 192 instants filtered at once, three per lane (three independent add chains in flight),
//   * the votes reduced to their sum and the largest / smallest prefix sum (ballots, popcounts, one DPP max / min),
//   * if start value + extreme prefix sums stay inside the threshold (and the carried vote tick, if any, does
//     too), no branch step can have happened: the reference's sample-by-sample loop would have produced exactly
//     these filter outputs, votes and symbols.  The counter advances by the vote sum and, when the framer is
//     locked, the one frame that ends in the block gets its sync check straight from the symbol ring.
//   * otherwise the block is run by the general code -- rounds of 64 instants with ballots, stopping at each
//     threshold crossing, and the full framer including the hunt -- which is also what an unlocked channel uses.
// Nothing speculative is ever kept: the whole-block result is used only when its assumptions were verified.
// Measured on the benchmark signal 92 % of the blocks of a tracking channel take the first path.
//
// The control state of a channel is wave-uniform (scalar registers, scalar branches); four channels share a
// workgroup only for the tap table in LDS.  No workgroup barrier after the prologue.
#pragma clang fp contract(off)

namespace m17dev {

constexpr int kBlkRing = 1024;                 // symbol ring: the frame in progress (<= 191 back) + a block (<= 193)
constexpr int BLK_WAVES = 4;

struct BlkChan {
    float x[kTaps - 1 + kDiscOut + 2];         // 30 history samples + the block's 384 inputs
    float H[kBlkRing];                         // symbol ring
};

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

__global__ __launch_bounds__(64 * BLK_WAVES)
void k_sync_frame_blk(const float *__restrict__ disc,     // [C][nblk][384]
                      const float *__restrict__ offs,     // [C][nblk] or null (already DC-free)
                      ChanState *__restrict__ st, int C, int nblk, int mode,
                      m17gpu_rec_dev *__restrict__ recs, int rec_cap, int32_t *__restrict__ counts,
                      float *__restrict__ syms, int32_t *__restrict__ nsyms,
                      float *__restrict__ fsym, int b0, int bcount)
{
    constexpr int RM = kBlkRing - 1;
    __shared__ __attribute__((aligned(16))) float taps[kPhases * 64];      // (matched, derivative) pairs per branch
    __shared__ __attribute__((aligned(16))) BlkChan chs[BLK_WAVES];
    const int gl = lane_id();
    const int wave = uni((int)(threadIdx.x >> 6));
    for (int q = (int)threadIdx.x; q < kPhases * 16; q += 64 * BLK_WAVES)
        reinterpret_cast<float4 *>(taps)[q] = reinterpret_cast<const float4 *>(&c_tab.tap_pairs[0][0])[q];
    __syncthreads();                                        // the only workgroup barrier
    const int chan = (int)blockIdx.x * BLK_WAVES + wave;
    if (chan >= C) return;
    BlkChan &my = chs[wave];
    ChanState &cs = st[chan];
    const int bend = b0 + bcount;
    const unsigned long long incl = (gl == 63) ? ~0ull : ((2ull << gl) - 1ull);
    const unsigned incl_lo = (unsigned)incl, incl_hi = (unsigned)(incl >> 32);

    int clk = uni(cs.clk), thr = uni(cs.thr), index = uni(cs.index);
    float sum = unif(cs.sum), dif = unif(cs.dif);
    int flock = uni(cs.flock), fclk = uni(cs.fclk), ferr = uni(cs.ferr);
    uint32_t block_count = (uint32_t)uni((int)cs.block_count);
    int nrec = (b0 == 0) ? 0 : uni(counts[chan]);
    int sym_total = (b0 == 0) ? 0 : uni(cs.sym_total);
    int hp = 256;                                           // ring position of the current block's first symbol
    m17gpu_rec_dev *crecs = recs ? recs + (size_t)chan * rec_cap : nullptr;
    if (!recs) rec_cap = 0;
    float *sym_base = syms ? syms + (size_t)chan * M17_SYM_STRIDE(nblk) : nullptr;
    // m_f_sym[0 .. fclk) is the frame in progress: ring [hp - fclk, hp); m_sync is the last 8 symbols
    if (flock) { for (int q = gl; q < fclk; q += 64) my.H[(hp - fclk + q) & RM] = cs.fsym[q]; }
    else if (gl < 8) my.H[(hp - 8 + gl) & RM] = cs.sync[gl];
    const float *dsrc = disc + (size_t)chan * nblk * kDiscOut;
    const float *osrc = offs ? offs + (size_t)chan * nblk : nullptr;

    // ---- one frame's sync check result -> flags, record, frame symbols for the decoder, unlock (m17_rx_frame.cpp:132-152)
    // returns true when the framer unlocked
    auto frame_done = [&](const SyncResult &r, int hpb, int pos) -> bool {
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
            const int fs = hpb + pos - kFrameSyms;               // the frame sits in the ring, in place
            float *fd = fsym + ((size_t)chan * rec_cap + nrec) * kFrameSyms;
#pragma unroll
            for (int q = 0; q < 3; ++q) fd[gl + 64 * q] = my.H[(fs + gl + 64 * q) & RM];
        }
        nrec++;
        if (unlock) {
            flock = 0;
            // reset_sync(): the next hunt windows must see zeros behind them
            wave_fence();
            if (gl < 8) my.H[(hpb + pos - 8 + gl) & RM] = 0.0f;
            wave_fence();
        }
        return unlock;
    };
    // ---- the framer over ring symbols [hpb + pos0, hpb + n)   (m17_rx_frame.cpp:126-177)
    auto framer_ring = [&](int hpb, int pos0, int n) {
        int pos = pos0;
        while (pos < n) {
            if (flock) {
                const int cnt = min(kFrameSyms - fclk, n - pos);
                fclk += cnt; pos += cnt;
                if (fclk == kFrameSyms) {
                    fclk = 0;
                    const int fs = hpb + pos - kFrameSyms;
                    float v[8];
#pragma unroll
                    for (int i = 0; i < 8; ++i) v[i] = my.H[(fs + i) & RM];
                    const SyncResult r = sync_check_grp<64>(v, gl, 0, 0);
                    (void)frame_done(r, hpb, pos);
                }
            } else {
                // hunt: candidate symbol j = pos+gl, window = ring [hpb+j-7, hpb+j]
                const int jc = pos + gl;
                const bool cand = jc < n;
                const int jj = cand ? jc : pos;
                float v[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) v[i] = my.H[(hpb + jj - 7 + i) & RM];
                SyncResult r; r.type = 0; r.votes = 8; r.variance = 1.0f;
                if (cand && hunt_compatible(v)) r = sync_check(v);           // most windows are rejected by sign
                const unsigned long long hm = __builtin_amdgcn_ballot_w64(cand && sync_accept(r, false));
                if (hm) {
                    const int l = (int)__ffsll((long long)hm) - 1;
                    const int js = pos + l;
                    // copy_sync(); m_fclk = 8; lock; m17_aos(): the window already is the head of the frame
                    fclk = 8; ferr = 0; flock = 1;
                    const int ty = __shfl(r.type, l, 64), vo = __shfl(r.votes, l, 64);
                    const float va = __shfl(r.variance, l, 64);
                    emit_record_grp(crecs, rec_cap, nrec, gl, (uint32_t)ty | ((uint32_t)vo << 8), M17_F_AOS, va,
                                    block_count, (uint32_t)js);
                    nrec++;
                    pos = js + 1;
                } else {
                    pos = min(n, pos + 64);
                }
            }
        }
    };

    // ---- block input: element i of x is stream sample b*384 - 30 + i; the 30 history samples are the tail of the
    // block before with ITS DC estimate (the channel state for the first block of the call)
    float pf[7];
    auto fetch = [&](int bb) {
        const float off_cur = osrc ? osrc[bb] : 0.0f;
        const float off_prv = (osrc && bb > b0) ? osrc[bb - 1] : 0.0f;
        const float *src = dsrc + (size_t)bb * kDiscOut - (kTaps - 1);
#pragma unroll
        for (int r = 0; r < 7; ++r) {
            const int i = gl + 64 * r;
            float v = 0.0f;
            if (i < kTaps - 1 + kDiscOut) {
                if (i < kTaps - 1 && bb == b0) v = cs.buff[i + 1];
                else { v = src[i]; if (osrc) v = v - (i < kTaps - 1 ? off_prv : off_cur); }   // out[i] - offset (m17_dsp.cpp:217-219)
            }
            pf[r] = v;
        }
    };
    fetch(b0);

#ifdef M17_STAMPS
    unsigned long long acc_[12] = {0,0,0,0,0,0,0,0,0,0,0,0}, last_ = __builtin_amdgcn_s_memtime();
    __builtin_amdgcn_s_waitcnt(0xC07F);
#define BCNT(i, v) acc_[i] += (v)
#else
#define BCNT(i, v) do {} while (0)
#endif
    for (int b = b0; b < bend; ++b) {
        STAMP(0);
        // ---- commit this block's input, start fetching the next
#pragma unroll
        for (int r = 0; r < 7; ++r) {
            const int i = gl + 64 * r;
            if (i < kTaps - 1 + kDiscOut) my.x[i] = pf[r];
        }
        if (b + 1 < bend) fetch(b + 1);
        wave_fence();
        STAMP(1);

        const int thresh = flock ? 80 : 10;
        const int p0 = clk ? 1 : 0;
        // ---- the whole block under the current branch and tick parity
        float s[3], d[3];
        {
            float4 tp[16];
            const float4 *t4 = reinterpret_cast<const float4 *>(&taps[64 * index]);
#pragma unroll
            for (int q = 0; q < 16; ++q) tp[q] = t4[q];
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                const v2f a = fir_pair(my.x + p0 + 2 * (gl + 64 * r), tp);
                s[r] = a.x; d[r] = a.y;
                __builtin_amdgcn_sched_barrier(0);      // one window of 31 inputs in registers at a time
            }
        }
        STAMP(2);
        // votes (sync_update, m17_rx_sync.cpp:38-42) in time order: segment r, lane
        unsigned long long um[3], dm[3];
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const bool vok = (p0 + 2 * (gl + 64 * r) + 1 < kDiscOut);     // the last instant's tick falls into the next block when p0 == 1
            const float dd = (s[r] < 0.0f) ? -d[r] : d[r];
            um[r] = __builtin_amdgcn_ballot_w64(vok && dd > 0.0f);
            dm[r] = __builtin_amdgcn_ballot_w64(vok && dd < 0.0f);
        }
        int vsum = 0, mx = -1000, mn = 1000;
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const int pr = vsum + (int)__builtin_popcount((unsigned)um[r] & incl_lo) + (int)__builtin_popcount((unsigned)(um[r] >> 32) & incl_hi)
                                - (int)__builtin_popcount((unsigned)dm[r] & incl_lo) - (int)__builtin_popcount((unsigned)(dm[r] >> 32) & incl_hi);
            mx = max(mx, pr); mn = min(mn, pr);
            vsum += (int)__popcll(um[r]) - (int)__popcll(dm[r]);
        }
        const int vmax = wave_max_i(mx), vmin = wave_min_i(mn);
        // ---- would the sample-by-sample loop have stepped the branch anywhere in this block?
        bool ok = true;
        int thr_c = thr;
        if (clk == 1) {
            // the vote tick carried over from the block before (sync_update / m17_sync_adjust)
            const float d0 = (sum < 0.0f) ? -dif : dif;
            if (d0 > 0.0f) thr_c++;
            if (d0 < 0.0f) thr_c--;
            if (thr_c > thresh || thr_c < -thresh) ok = false;
        }
        if (thr_c + vmax > thresh || thr_c + vmin < -thresh) ok = false;
        STAMP(3);
        int n;
        if (ok) {
            BCNT(8, 1);
            // ---- accepted as computed
            thr = thr_c + vsum;
            sum = bcast_lane(s[2], 63); dif = bcast_lane(d[2], 63);
            n = kFrameSyms;
#pragma unroll
            for (int r = 0; r < 3; ++r) my.H[(hp + gl + 64 * r) & RM] = s[r];
            if (sym_base) {
                float *so = sym_base + sym_total;
#pragma unroll
                for (int r = 0; r < 3; ++r) so[gl + 64 * r] = s[r];
            }
            wave_fence();
        } else {
            BCNT(9, 1);
            // ---- the general timing loop (rounds of 64 instants, stop at each threshold crossing)
            int p = 0, m_idx = 0;
            while (p < kDiscOut) {
                if (clk == 1) {
                    clk = 0;
                    const float d0 = (sum < 0.0f) ? -dif : dif;
                    if (d0 > 0.0f) thr++;
                    if (d0 < 0.0f) thr--;
                    if (thr > thresh) {
                        index = (index + 1 == kPhases) ? 0 : index + 1; thr = 0;
                        if (index == 0) { clk = 1; if (m_idx >= 0 && gl == 0) my.H[(hp + m_idx) & RM] = 0.0f; m_idx++; }
                    }
                    if (thr < -thresh) {
                        thr = 0; index = (index == 0) ? kPhases - 1 : index - 1;
                        if (index == kPhases - 1) { clk = 1; m_idx--; }
                    }
                    p++;
                    continue;
                }
                float4 tp[16];
                {
                    const float4 *t4 = reinterpret_cast<const float4 *>(&taps[64 * index]);
#pragma unroll
                    for (int q = 0; q < 16; ++q) tp[q] = t4[q];
                }
                const int rem = (kDiscOut - p + 1) >> 1;          // filter instants left in the block
                const int nv = rem < 64 ? rem : 64;
                const v2f a = fir_pair(my.x + p + 2 * (gl < nv ? gl : 0), tp);
                const float s1 = a.x, d1 = a.y;
                const bool vote_ok = (gl < nv) && (p + 2 * gl + 1 < kDiscOut);
                const float dd = (s1 < 0.0f) ? -d1 : d1;
                const unsigned long long u1 = __builtin_amdgcn_ballot_w64(vote_ok && dd > 0.0f);
                const unsigned long long d1m = __builtin_amdgcn_ballot_w64(vote_ok && dd < 0.0f);
                const int tk = thr + (int)__popcll(u1 & incl) - (int)__popcll(d1m & incl);
                const unsigned long long cr = __builtin_amdgcn_ballot_w64(vote_ok && (tk > thresh || tk < -thresh));
                const int kl = cr ? (int)__ffsll((long long)cr) - 1 : 0;
                const int naccept = cr ? kl + 1 : nv;
                if (gl < naccept && (m_idx + gl) >= 0) my.H[(hp + m_idx + gl) & RM] = s1;
                m_idx += naccept;
                sum = __shfl(s1, naccept - 1, 64);
                dif = __shfl(d1, naccept - 1, 64);
                if (cr) {
                    const int ts = __shfl(tk, kl, 64);
                    thr = 0; clk = 0;
                    if (ts > thresh) {
                        index = (index + 1 == kPhases) ? 0 : index + 1;
                        if (index == 0) { clk = 1; if (m_idx >= 0 && gl == 0) my.H[(hp + m_idx) & RM] = 0.0f; m_idx++; }
                    } else {
                        index = (index == 0) ? kPhases - 1 : index - 1;
                        if (index == kPhases - 1) { clk = 1; m_idx--; }
                    }
                    p = p + 2 * kl + 2;
                } else {
                    thr += (int)__popcll(u1) - (int)__popcll(d1m);
                    const int ilast = p + 2 * (nv - 1);
                    if (ilast + 1 < kDiscOut) { clk = 0; p = ilast + 2; }
                    else { clk = 1; p = kDiscOut; }
                }
            }
            n = m_idx > 0 ? m_idx : 0;
            wave_fence();
            if (sym_base) {
                float *so = sym_base + sym_total;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int q = gl + 64 * r;
                    if (q < n) so[q] = my.H[(hp + q) & RM];
                }
            }
        }
        STAMP(4);
        if (nsyms && gl == 0) nsyms[(size_t)chan * nblk + b] = n;
        // ---- framer
        if (flock && n == kFrameSyms) {
            // locked, a whole frame's worth of symbols: the frame in progress ends at symbol 191 - fclk, the next one
            // is fclk symbols in at the end of the block
            const int pos = kFrameSyms - fclk;
            const int fs = hp + pos - kFrameSyms;
            float v[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) v[i] = my.H[(fs + i) & RM];
            const SyncResult r = sync_check_grp<64>(v, gl, 0, 0);
            if (frame_done(r, hp, pos)) { fclk = 0; framer_ring(hp, pos, n); }       // unlocked: the rest of the block hunts
        } else
            framer_ring(hp, 0, n);
        block_count++;
        hp += n; sym_total += n;
        STAMP(5);
    }
#ifdef M17_STAMPS
    if (chan == 7 && gl == 0) for (int i = 0; i < 12; ++i) g_stamps[i] = acc_[i];
#endif

    // ---- store state in the reference's layout
    if (gl == 0) {
        cs.clk = clk; cs.thr = thr; cs.index = index; cs.sum = sum; cs.dif = dif; cs.buff[0] = 0.0f;
        cs.flock = flock; cs.fclk = fclk; cs.ferr = ferr; cs.block_count = block_count; cs.sym_total = sym_total;
        if (counts) counts[chan] = nrec;
    }
    if (gl < kTaps - 1) cs.buff[gl + 1] = my.x[kDiscOut + gl];      // m_buff: the last 30 inputs
    if (flock) { for (int q = gl; q < kFrameSyms; q += 64) cs.fsym[q] = my.H[(hp - fclk + q) & RM]; }
    else if (gl < 8) cs.sync[gl] = my.H[(hp - 8 + gl) & RM];
}

} // namespace m17dev
