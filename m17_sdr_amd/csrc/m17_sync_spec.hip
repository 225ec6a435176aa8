// m17_sync_spec.hip -- k_sync_frame_spec<S>: timing recovery + sync correlator + framer with S CONSECUTIVE
// BLOCKS of one channel in flight at once, one wave per block.
//
// Reference: m17_rx_sync_samples (m17_rx_sync.cpp:77-99), m17_rx_sym (m17_rx_frame.cpp:126-177).
//
// Why.  A channel's blocks form a chain: the timing loop carries (m_clk, m_thr, m_index, sum, dif) from block to
// block and takes the framer's lock flag of the block before.  One wave walking that chain pays every
// instruction of every block in sequence (~1,300 per block); with few channels the chain IS the run time.
// But while a channel tracks, almost nothing of that state moves: the polyphase branch steps about once in
// ten blocks, the tick parity only on a wrap of the branch, and the matched / derivative filter outputs --
// all the arithmetic -- depend on branch and parity alone, not on the vote counter or the lock flag.  So:
//
//   step 1 (S waves, one block each)  assume the branch and parity the pass starts with; filter all 192 instants
//            of the block (three per lane, three independent chains), turn them into votes, and reduce the votes to
//            three numbers: their sum and the largest / smallest prefix sum.  Symbols go to the channel's ring.
//   step 2 (S waves)  the sync check of the frame that ends in the block, assuming the framer stays locked with
//            the frame phase it has.
//   step 3 (wave 0)   walk the S blocks in order with the real state: a block whose votes keep the counter inside
//            the threshold (start value + extreme prefix sums) and whose framer was locked is ACCEPTED as computed
//            -- the counter advances by the vote sum, the framer consumes the prepared sync check.  The first block
//            that fails the test is run again by the general code (rounds of 64 instants with ballots, full framer
//            incl. the hunt), exactly like the reference would, and the pass ends behind it: what the later waves
//            assumed no longer holds.
//
// Nothing speculative is ever committed: an accepted block is one for which the assumptions were verified, and then
// the reference's sequential loop produces the same filter outputs, votes, counter and symbols by construction.
// All cross-wave hand-offs go through LDS behind workgroup barriers; every wave executes the same three barriers
// per pass and the pass count depends only on state wave 0 publishes, so no wave can wait for a barrier alone.
#pragma clang fp contract(off)

namespace m17dev {

constexpr int kSpecRing = 2048;                // symbol ring: S * 193 new symbols + the frame in progress, S <= 8

template <int S> struct SpecChan {
    float taps[kPhases * 64];                  // (matched, derivative) tap pairs of all 40 branches
    float x[S][kTaps - 1 + kDiscOut + 2];      // per wave: 30 history samples + its block's 384 inputs
    float H[kSpecRing];                        // symbol ring of the channel
    int   mail[S][8];                          // per block: vote sum, max prefix, min prefix, last s, last d, sync type|votes<<8, variance
    int   state[12];                           // published by wave 0 at the end of a pass
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

template <int S>
__global__ __launch_bounds__(64 * S, S)                     // 4 workgroups per CU: 1,024 channels are resident at once
void k_sync_frame_spec(const float *__restrict__ disc,     // [C][nblk][384]
                       const float *__restrict__ offs,     // [C][nblk] or null (already DC-free)
                       ChanState *__restrict__ st, int C, int nblk, int mode,
                       m17gpu_rec_dev *__restrict__ recs, int rec_cap, int32_t *__restrict__ counts,
                       float *__restrict__ syms, int32_t *__restrict__ nsyms,
                       float *__restrict__ fsym, int b0, int bcount)
{
    constexpr int RM = kSpecRing - 1;
    __shared__ __attribute__((aligned(16))) SpecChan<S> my;
    const int gl = lane_id();
    const int wave = uni((int)(threadIdx.x >> 6));
    const int chan = (int)blockIdx.x;
    if (chan >= C) return;                                  // whole workgroups only
    ChanState &cs = st[chan];
    const int bend = b0 + bcount;
    const unsigned long long incl = (gl == 63) ? ~0ull : ((2ull << gl) - 1ull);
    const unsigned incl_lo = (unsigned)incl, incl_hi = (unsigned)(incl >> 32);

    // ---- pass state: every wave holds what step 1 / 2 need; wave 0 holds all of it
    int b = b0;
    int clk = uni(cs.clk), index = uni(cs.index), flock = uni(cs.flock), fclk = uni(cs.fclk);
    int sym_total = (b0 == 0) ? 0 : uni(cs.sym_total);
    int hp = 256;                                           // ring position of block b's first symbol
    // wave 0 only
    int thr = uni(cs.thr), ferr = uni(cs.ferr);
    float sum = unif(cs.sum), dif = unif(cs.dif);
    uint32_t block_count = (uint32_t)uni((int)cs.block_count);
    int nrec = (b0 == 0) ? 0 : uni(counts[chan]);
    m17gpu_rec_dev *crecs = recs ? recs + (size_t)chan * rec_cap : nullptr;
    if (!recs) rec_cap = 0;
    float *sym_base = syms ? syms + (size_t)chan * M17_SYM_STRIDE(nblk) : nullptr;

    for (int q = (int)threadIdx.x; q < kPhases * 16; q += 64 * S)
        reinterpret_cast<float4 *>(my.taps)[q] = reinterpret_cast<const float4 *>(&c_tab.tap_pairs[0][0])[q];
    if (wave == 0) {
        // m_f_sym[0 .. fclk) is the frame in progress: ring [hp - fclk, hp); m_sync is the last 8 symbols
        if (flock) { for (int q = gl; q < fclk; q += 64) my.H[(hp - fclk + q) & RM] = cs.fsym[q]; }
        else if (gl < 8) my.H[(hp - 8 + gl) & RM] = cs.sync[gl];
    }
    const float *dsrc = disc + (size_t)chan * nblk * kDiscOut;
    const float *osrc = offs ? offs + (size_t)chan * nblk : nullptr;
    lds_barrier();

    // =====================================================================================================
    // general code (wave 0): the framer over ring symbols [hpb + pos0, hpb + n)   (m17_rx_frame.cpp:126-177)
    // =====================================================================================================
    auto framer_ring = [&](int hpb, int pos0, int n) {
        int pos = pos0;
        while (pos < n) {
            if (flock) {
                const int cnt = min(kFrameSyms - fclk, n - pos);
                fclk += cnt; pos += cnt;
                if (fclk == kFrameSyms) {
                    fclk = 0;
                    const int fs = hpb + pos - kFrameSyms;               // the frame sits in the ring, in place
                    float v[8];
#pragma unroll
                    for (int i = 0; i < 8; ++i) v[i] = my.H[(fs + i) & RM];
                    const SyncResult r = sync_check_grp<64>(v, gl, 0, 0);
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
                        float *fd = fsym + ((size_t)chan * rec_cap + nrec) * kFrameSyms;
                        for (int q = gl; q < kFrameSyms; q += 64) fd[q] = my.H[(fs + q) & RM];
                    }
                    nrec++;
                    if (unlock) {
                        flock = 0;
                        // reset_sync(): the next hunt windows must see zeros behind them
                        wave_fence();
                        if (gl < 8) my.H[(hpb + pos - 8 + gl) & RM] = 0.0f;
                        wave_fence();
                    }
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

    float keep_s[3] = {0.0f, 0.0f, 0.0f};
#ifdef M17_STAMPS
    unsigned long long acc_[12] = {0,0,0,0,0,0,0,0,0,0,0,0}, last_ = __builtin_amdgcn_s_memtime();
    __builtin_amdgcn_s_waitcnt(0xC07F);
#define SCNT(i, v) acc_[i] += (v)
#else
#define SCNT(i, v) do {} while (0)
#endif
    while (b < bend) {
        STAMP(0);
        const int bj = b + wave;
        const bool active = bj < bend;
        const int p0 = clk ? 1 : 0;
        // =================================================================================================
        // step 1: this wave's block under the pass's branch and tick parity
        // =================================================================================================
        if (active) {
            float *xw = my.x[wave];
            {
                // 30 history samples (the tail of the block before, with ITS DC estimate; the channel state for the
                // first block of the call) + the block's 384 inputs
                const float off_cur = osrc ? osrc[bj] : 0.0f;
                const float off_prv = (osrc && bj > b0) ? osrc[bj - 1] : 0.0f;
                const float *src = dsrc + (size_t)bj * kDiscOut - (kTaps - 1);
#pragma unroll
                for (int r = 0; r < 7; ++r) {
                    const int i = gl + 64 * r;
                    if (i < kTaps - 1 + kDiscOut) {
                        float v;
                        if (i < kTaps - 1 && bj == b0) v = cs.buff[i + 1];
                        else { v = src[i]; if (osrc) v = v - (i < kTaps - 1 ? off_prv : off_cur); }   // out[i] - offset (m17_dsp.cpp:217-219)
                        xw[i] = v;
                    }
                }
            }
            wave_fence();
            float4 tp[16];
            {
                const float4 *t4 = reinterpret_cast<const float4 *>(&my.taps[64 * index]);
#pragma unroll
                for (int q = 0; q < 16; ++q) tp[q] = t4[q];
            }
            float s[3], d[3];
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                const v2f a = fir_pair(xw + p0 + 2 * (gl + 64 * r), tp);
                s[r] = a.x; d[r] = a.y;
                __builtin_amdgcn_sched_barrier(0);      // one window of 31 inputs in registers at a time
            }
            // votes (sync_update, m17_rx_sync.cpp:38-42) in time order: segment r, lane
            unsigned long long um[3], dm[3];
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                const bool vok = (p0 + 2 * (gl + 64 * r) + 1 < kDiscOut);     // the last instant's tick falls into the next block when p0 == 1
                const float dd = (s[r] < 0.0f) ? -d[r] : d[r];
                um[r] = __builtin_amdgcn_ballot_w64(vok && dd > 0.0f);
                dm[r] = __builtin_amdgcn_ballot_w64(vok && dd < 0.0f);
            }
            int base = 0, mx = -1000, mn = 1000;
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                const int pr = base + (int)__builtin_popcount((unsigned)um[r] & incl_lo) + (int)__builtin_popcount((unsigned)(um[r] >> 32) & incl_hi)
                                    - (int)__builtin_popcount((unsigned)dm[r] & incl_lo) - (int)__builtin_popcount((unsigned)(dm[r] >> 32) & incl_hi);
                mx = max(mx, pr); mn = min(mn, pr);
                base += (int)__popcll(um[r]) - (int)__popcll(dm[r]);
            }
            const int wmx = wave_max_i(mx), wmn = wave_min_i(mn);
            const float ls = bcast_lane(s[2], 63), ld = bcast_lane(d[2], 63);
            if (gl == 0) {
                int *ml = my.mail[wave];
                ml[0] = base; ml[1] = wmx; ml[2] = wmn; ml[3] = __float_as_int(ls); ml[4] = __float_as_int(ld);
            }
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                my.H[(hp + kFrameSyms * wave + gl + 64 * r) & RM] = s[r];
                keep_s[r] = s[r];
            }
        }
        STAMP(1);
        lds_barrier();
        STAMP(2);
        // =================================================================================================
        // step 2: the sync check of the frame that ends in this block, if the framer stays locked in phase
        // =================================================================================================
        if (active && flock) {
            const int fs = hp + kFrameSyms * wave - fclk;            // frame start: fclk symbols before the block
            float v[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) v[i] = my.H[(fs + i) & RM];
            const SyncResult r = sync_check_grp<64>(v, gl, 0, 0);
            if (gl == 0) { my.mail[wave][5] = r.type | (r.votes << 8); my.mail[wave][6] = __float_as_int(r.variance); }
        }
        STAMP(3);
        lds_barrier();
        STAMP(4);
        // =================================================================================================
        // step 3 (wave 0): accept the leading blocks whose assumptions hold; run the first other one in full
        // =================================================================================================
        const int sym_pass = sym_total;
        if (wave == 0) {
            const int npass = min(S, bend - b);
            const int fclk_pass = fclk;
            bool phase_ok = (flock != 0);
            int j = 0, nclean = 0, hp_c = hp;
            // the whole mailbox in one LDS read (word w of block j in lane 8 j + w); the walk below takes its scalars
            // from that register with v_readlane instead of one LDS round trip per word
            const int mbox = reinterpret_cast<const int *>(my.mail)[gl < 8 * S ? gl : 0];
            auto mword = [&](int blk, int w) { return __builtin_amdgcn_readlane(mbox, 8 * blk + w); };
            for (; j < npass; ++j) {
                bool ok = phase_ok && flock && fclk == fclk_pass;
                const int thresh = flock ? 80 : 10;
                int thr_c = thr;
                if (ok && clk == 1) {
                    // the vote tick carried over from the block before (sync_update / m17_sync_adjust)
                    const float d0 = (sum < 0.0f) ? -dif : dif;
                    if (d0 > 0.0f) thr_c++;
                    if (d0 < 0.0f) thr_c--;
                    if (thr_c > thresh || thr_c < -thresh) ok = false;
                }
                const int vsum = mword(j, 0), vmax = mword(j, 1), vmin = mword(j, 2);
                if (ok && (thr_c + vmax > thresh || thr_c + vmin < -thresh)) ok = false;
                if (!ok) break;
                // ---- accepted as computed: timing state after the block
                thr = thr_c + vsum;
                sum = __int_as_float(mword(j, 3)); dif = __int_as_float(mword(j, 4));
                // ---- framer: 192 symbols, the frame in progress completes at symbol 191 - fclk
                {
                    const int pos = kFrameSyms - fclk;
                    const int w5 = mword(j, 5);
                    SyncResult r; r.type = w5 & 0xFF; r.votes = (w5 >> 8) & 0xFF; r.variance = __int_as_float(mword(j, 6));
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
                        const int fs = hp_c + pos - kFrameSyms;
                        float *fd = fsym + ((size_t)chan * rec_cap + nrec) * kFrameSyms;
                        for (int q = gl; q < kFrameSyms; q += 64) fd[q] = my.H[(fs + q) & RM];
                    }
                    nrec++;
                    if (unlock) {
                        flock = 0; fclk = 0;
                        wave_fence();
                        if (gl < 8) my.H[(hp_c + pos - 8 + gl) & RM] = 0.0f;       // reset_sync()
                        wave_fence();
                        framer_ring(hp_c, pos, kFrameSyms);                         // the rest of the block: hunt
                        phase_ok = false;                                           // prepared sync checks are void now
                    }
                    // still locked: fclk + 192 - 192, unchanged
                }
                if (nsyms && gl == 0) nsyms[(size_t)chan * nblk + b + j] = kFrameSyms;
                block_count++;
                hp_c += kFrameSyms; sym_total += kFrameSyms; nclean++;
            }
            STAMP(5);
            SCNT(8, 1); SCNT(9, nclean);
            if (j < npass) {
                SCNT(10, 1);
                // ---- the general timing loop on block b + j (rounds of 64 instants), then the general framer
                const float *xb = my.x[j];
                const int thresh = flock ? 80 : 10;
                int p = 0, m_idx = 0;
                while (p < kDiscOut) {
                    if (clk == 1) {
                        clk = 0;
                        const float d0 = (sum < 0.0f) ? -dif : dif;
                        if (d0 > 0.0f) thr++;
                        if (d0 < 0.0f) thr--;
                        if (thr > thresh) {
                            index = (index + 1 == kPhases) ? 0 : index + 1; thr = 0;
                            if (index == 0) { clk = 1; if (m_idx >= 0 && gl == 0) my.H[(hp_c + m_idx) & RM] = 0.0f; m_idx++; }
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
                        const float4 *t4 = reinterpret_cast<const float4 *>(&my.taps[64 * index]);
#pragma unroll
                        for (int q = 0; q < 16; ++q) tp[q] = t4[q];
                    }
                    const int rem = (kDiscOut - p + 1) >> 1;          // filter instants left in the block
                    const int nv = rem < 64 ? rem : 64;
                    const v2f a = fir_pair(xb + p + 2 * (gl < nv ? gl : 0), tp);
                    const float s = a.x, d = a.y;
                    const bool vote_ok = (gl < nv) && (p + 2 * gl + 1 < kDiscOut);
                    const float dd = (s < 0.0f) ? -d : d;
                    const unsigned long long um = __builtin_amdgcn_ballot_w64(vote_ok && dd > 0.0f);
                    const unsigned long long dm = __builtin_amdgcn_ballot_w64(vote_ok && dd < 0.0f);
                    const int tk = thr + (int)__popcll(um & incl) - (int)__popcll(dm & incl);
                    const unsigned long long cr = __builtin_amdgcn_ballot_w64(vote_ok && (tk > thresh || tk < -thresh));
                    const int kl = cr ? (int)__ffsll((long long)cr) - 1 : 0;
                    const int naccept = cr ? kl + 1 : nv;
                    if (gl < naccept && (m_idx + gl) >= 0) my.H[(hp_c + m_idx + gl) & RM] = s;
                    m_idx += naccept;
                    sum = __shfl(s, naccept - 1, 64);
                    dif = __shfl(d, naccept - 1, 64);
                    if (cr) {
                        const int ts = __shfl(tk, kl, 64);
                        thr = 0; clk = 0;
                        if (ts > thresh) {
                            index = (index + 1 == kPhases) ? 0 : index + 1;
                            if (index == 0) { clk = 1; if (m_idx >= 0 && gl == 0) my.H[(hp_c + m_idx) & RM] = 0.0f; m_idx++; }
                        } else {
                            index = (index == 0) ? kPhases - 1 : index - 1;
                            if (index == kPhases - 1) { clk = 1; m_idx--; }
                        }
                        p = p + 2 * kl + 2;
                    } else {
                        thr += (int)__popcll(um) - (int)__popcll(dm);
                        const int ilast = p + 2 * (nv - 1);
                        if (ilast + 1 < kDiscOut) { clk = 0; p = ilast + 2; }
                        else { clk = 1; p = kDiscOut; }
                    }
                }
                const int n = m_idx > 0 ? m_idx : 0;
                wave_fence();
                if (sym_base) {
                    float *so = sym_base + sym_total;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int q = gl + 64 * r;
                        if (q < n) so[q] = my.H[(hp_c + q) & RM];
                    }
                }
                if (nsyms && gl == 0) nsyms[(size_t)chan * nblk + b + j] = n;
                framer_ring(hp_c, 0, n);
                block_count++;
                hp_c += n; sym_total += n; j++;
            }
            STAMP(6);
            b += j; hp = hp_c;
            wave_fence();
            if (gl == 0) {
                int *ps = my.state;
                ps[0] = b; ps[1] = clk; ps[2] = index; ps[3] = hp; ps[4] = sym_total; ps[5] = flock; ps[6] = fclk; ps[7] = nclean;
            }
        }
        lds_barrier();
        // ---- the accepted blocks' symbols go out to the symbol stream from the waves that made them
        {
            const int sbox = my.state[gl & 7];                           // one LDS read, scalars by v_readlane
            const int nclean = __builtin_amdgcn_readlane(sbox, 7);
            if (sym_base && active && wave < nclean) {
                float *so = sym_base + sym_pass + kFrameSyms * wave;
#pragma unroll
                for (int r = 0; r < 3; ++r) so[gl + 64 * r] = keep_s[r];
            }
            if (wave != 0) {
                b = __builtin_amdgcn_readlane(sbox, 0); clk = __builtin_amdgcn_readlane(sbox, 1);
                index = __builtin_amdgcn_readlane(sbox, 2); hp = __builtin_amdgcn_readlane(sbox, 3);
                sym_total = __builtin_amdgcn_readlane(sbox, 4); flock = __builtin_amdgcn_readlane(sbox, 5);
                fclk = __builtin_amdgcn_readlane(sbox, 6);
            }
        }
    }

#ifdef M17_STAMPS
    if (chan == 7 && gl == 0 && wave == 0) for (int i = 0; i < 12; ++i) g_stamps[i] = acc_[i];
#endif
    // ---- store state in the reference's layout (wave 0 holds all of it)
    if (wave == 0) {
        if (gl == 0) {
            cs.clk = clk; cs.thr = thr; cs.index = index; cs.sum = sum; cs.dif = dif; cs.buff[0] = 0.0f;
            cs.flock = flock; cs.fclk = fclk; cs.ferr = ferr; cs.block_count = block_count; cs.sym_total = sym_total;
            if (counts) counts[chan] = nrec;
        }
        // m_buff: the last 30 inputs of the last block, DC-free
        if (gl < kTaps - 1) {
            float v = dsrc[(size_t)(bend - 1) * kDiscOut + kDiscOut - (kTaps - 1) + gl];
            if (osrc) v = v - osrc[bend - 1];
            cs.buff[gl + 1] = v;
        }
        if (flock) { for (int q = gl; q < kFrameSyms; q += 64) cs.fsym[q] = my.H[(hp - fclk + q) & RM]; }
        else if (gl < 8) cs.sync[gl] = my.H[(hp - 8 + gl) & RM];
    }
}

} // namespace m17dev
