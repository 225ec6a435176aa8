// m17_sync_tri.hip -- k_sync_frame_tri: timing recovery + sync correlator + framer of one channel on a
// workgroup of THREE waves: 192 lanes = the (at most) 192 symbol instants of a 1920-sample block, all taken
// in one round.  Used when channels are few (up to ~2,048): there the per-channel chain of blocks is the
// critical path, and a channel that owns one wave pays ~1,000 instructions per block at lone-wave speed.
//
// Reference: m17_rx_sync_samples (m17_rx_sync.cpp:77-99), m17_rx_sym (m17_rx_frame.cpp:126-177).
//
// Structure of a block:
//   all waves   FIR of "their" 64 instants under the current polyphase branch (speculative: a threshold
//               crossing invalidates what lies behind it); (matched, derivative) outputs and the two vote
//               ballots go to LDS; ONE workgroup barrier.
//   all waves   the SAME scalar control, replicated: every wave reads the three waves' ballots, finds the
//               first crossing (prefix counts over the 64-bit masks), accepts the instants up to it, steps
//               the branch, and -- rarely -- runs another round behind the crossing.  Identical inputs,
//               identical decisions: no wave ever waits for another's decision, and all waves execute the
//               same number of barriers by construction.
//   wave 0      additionally owns the symbol ring and the framer (frame sync check every 192 symbols, hunt,
//               records, frame symbols for the decoder); it posts the lock flag, which the others pick up
//               behind the next barrier -- exactly the flag the reference's timing loop would see
//               (m17_rx_lock(), m17_rx_sync.cpp:92-95), no speculation on it.
//   waves 1, 2  go straight to the next block's FIR, which does not depend on the framer.
// LDS hand-offs are double-buffered by round / block parity so that a wave running ahead never overwrites
// what a slower wave still has to read; every read of another wave's data sits behind the barrier that
// follows the write.
#pragma clang fp contract(off)

namespace m17dev {

constexpr int kTriRing = 1024;                 // symbol ring (power of two; a block adds <= 193, a frame spans 192)
constexpr int kTriX = kTaps - 1 + kDiscOut + 2;

struct TriChan {
    float    taps[kPhases * 64];               // (matched, derivative) tap pairs of all 40 branches
    float    x[2][kTriX];                      // [block parity]: 30 history samples + the block's 384 inputs
    float    H[kTriRing];                      // symbol ring, wave 0 only
    float2   sd[2][192];                       // [round parity][instant]: matched / derivative filter outputs
    uint32_t vote[2][3][4];                    // [round parity][wave]: up.lo, up.hi, dn.lo, dn.hi
    int      lock_after[2];                    // [block parity]: lock flag after the framer of that block
    int      pad[2];
};

__global__ __launch_bounds__(192)
void k_sync_frame_tri(const float *__restrict__ disc,     // [C][nblk][384]
                      const float *__restrict__ offs,     // [C][nblk] or null (already DC-free)
                      ChanState *__restrict__ st, int C, int nblk, int mode,
                      m17gpu_rec_dev *__restrict__ recs, int rec_cap, int32_t *__restrict__ counts,
                      float *__restrict__ syms, int32_t *__restrict__ nsyms,
                      float *__restrict__ fsym, int b0, int bcount)
{
    __shared__ __attribute__((aligned(16))) TriChan my;
    const int gl = lane_id();
    const int wave = uni((int)(threadIdx.x >> 6));
    const int k = (int)threadIdx.x;                         // this lane's instant inside a round
    const int chan = (int)blockIdx.x;
    if (chan >= C) return;                                  // whole workgroups only
    ChanState &cs = st[chan];
    const int bend = b0 + bcount;
    const unsigned long long incl = (gl == 63) ? ~0ull : ((2ull << gl) - 1ull);
    const unsigned incl_lo = (unsigned)incl, incl_hi = (unsigned)(incl >> 32);

    // ---- timing state, replicated in the three waves (wave-uniform => scalar registers)
    int clk = uni(cs.clk), thr = uni(cs.thr), index = uni(cs.index);
    float sum = unif(cs.sum), dif = unif(cs.dif);
    int lockv = uni(cs.flock);                              // what m17_rx_lock() returns during the current block
    int sym_total = (b0 == 0) ? 0 : uni(cs.sym_total);
    int hp = 256;                                           // ring position of the current block's first symbol
    // ---- framer state, wave 0
    int flock = lockv, fclk = uni(cs.fclk), ferr = uni(cs.ferr);
    uint32_t block_count = (uint32_t)uni((int)cs.block_count);
    int nrec = (b0 == 0) ? 0 : uni(counts[chan]);
    m17gpu_rec_dev *crecs = recs ? recs + (size_t)chan * rec_cap : nullptr;
    if (!recs) rec_cap = 0;
    if (wave == 0) {
        // m_f_sym[0 .. fclk) is the frame in progress: ring [hp - fclk, hp); m_sync is the last 8 symbols
        if (flock) { for (int q = gl; q < fclk; q += 64) my.H[(hp - fclk + q) & (kTriRing - 1)] = cs.fsym[q]; }
        else if (gl < 8) my.H[(hp - 8 + gl) & (kTriRing - 1)] = cs.sync[gl];
    }
    const float *dsrc = disc + (size_t)chan * nblk * kDiscOut;
    const float *osrc = offs ? offs + (size_t)chan * nblk : nullptr;
    float *sym_out = syms ? syms + (size_t)chan * M17_SYM_STRIDE(nblk) + sym_total : nullptr;

    // ---- tap table and first block in: history from the state, inputs from the discriminator stream
    for (int q = k; q < kPhases * 16; q += 192)
        reinterpret_cast<float4 *>(my.taps)[q] = reinterpret_cast<const float4 *>(&c_tab.tap_pairs[0][0])[q];
    {
        const float off = osrc ? osrc[b0] : 0.0f;
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const int i = k + 192 * r;
            if (i < kTaps - 1 + kDiscOut) {
                float v;
                if (i < kTaps - 1) v = cs.buff[i + 1];
                else { v = dsrc[(size_t)b0 * kDiscOut + (i - (kTaps - 1))]; if (osrc) v = v - off; }   // out[i] - offset (m17_dsp.cpp:217-219)
                my.x[0][i] = v;
            }
        }
    }
    lds_barrier();

    int rnd = 0;                                            // rounds so far: parity selects the hand-off buffers
#ifdef M17_STAMPS
    unsigned long long acc_[12] = {0,0,0,0,0,0,0,0,0,0,0,0}, last_ = __builtin_amdgcn_s_memtime();
    __builtin_amdgcn_s_waitcnt(0xC07F);
#endif
    for (int b = b0; b < bend; ++b) {
        const int cur = (b - b0) & 1;
        const float *xb = my.x[cur];
        // ---- next block's input: loads issued now, committed before this block's first barrier.
        // Element i of the next buffer is stream sample (b+1)*384 - 30 + i: the 30 history samples are the tail
        // of THIS block (with this block's DC estimate), the rest the next block (with its own).
        float pf[3] = {0.0f, 0.0f, 0.0f};
        const bool have_next = (b + 1 < bend);
        if (have_next) {
            const float off_cur = osrc ? osrc[b] : 0.0f, off_nxt = osrc ? osrc[b + 1] : 0.0f;
            const float *nx = dsrc + (size_t)(b + 1) * kDiscOut - (kTaps - 1);
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                const int i = k + 192 * r;
                if (i < kTaps - 1 + kDiscOut) {
                    const float v = nx[i];
                    pf[r] = osrc ? (v - (i < kTaps - 1 ? off_cur : off_nxt)) : v;
                }
            }
        }
        bool committed = !have_next;
        STAMP(0);

        // ---- the lock flag of this block.  A vote tick carried over from the previous block comes before this
        // block's first barrier, so the flag has to be fetched behind a barrier of its own.
        if (b > b0 && clk == 1) {
            lds_barrier();
            lockv = uni(my.lock_after[(b - 1 - b0) & 1]);
        }
        bool lock_known = (b == b0) || (clk == 1);
        int thresh = lockv ? 80 : 10;
        int p = 0, m_idx = 0;
        bool rewrite = false;           // a downward wrap took the last symbol back: its slot in the symbol stream is
                                        // written again, possibly by another wave -- the two stores must not overtake
        while (p < kDiscOut) {
            if (clk == 1) {
                // vote tick on the carried sum / dif (sync_update :38-42, m17_sync_adjust :45-72)
                clk = 0;
                const float d0 = (sum < 0.0f) ? -dif : dif;
                if (d0 > 0.0f) thr++;
                if (d0 < 0.0f) thr--;
                if (thr > thresh) {
                    index = (index + 1 == kPhases) ? 0 : index + 1; thr = 0;
                    if (index == 0) {
                        clk = 1;
                        if (wave == 0 && m_idx >= 0 && gl == 0) my.H[(hp + m_idx) & (kTriRing - 1)] = 0.0f;
                        if (sym_out && wave == 0 && m_idx >= 0 && gl == 0) sym_out[m_idx] = 0.0f;
                        m_idx++;
                    }
                }
                if (thr < -thresh) {
                    thr = 0; index = (index == 0) ? kPhases - 1 : index - 1;
                    if (index == kPhases - 1) { clk = 1; m_idx--; rewrite = true; }
                }
                p++;
                continue;
            }
            const int rp = rnd & 1;
            const int rem = (kDiscOut - p + 1) >> 1;          // filter instants left in the block (<= 192)
            const bool valid = k < rem;
            float s = 0.0f, d = 0.0f;
            if (wave * 64 < rem) {                           // scalar: a wave with no instant left skips the filter
                // the branch's taps come from LDS every round (wave-uniform broadcast reads): one round per block is
                // the rule here, and tap registers kept across rounds cost 64 VGPRs plus a register shuffle per round
                float4 tp[16];
                const float4 *t4 = reinterpret_cast<const float4 *>(&my.taps[64 * index]);
#pragma unroll
                for (int q = 0; q < 16; ++q) tp[q] = t4[q];
                const v2f a = fir_pair(xb + p + 2 * (valid ? k : 0), tp);
                s = a.x; d = a.y;
            }
            const bool vote_ok = valid && (p + 2 * k + 1 < kDiscOut);
            const float dd = (s < 0.0f) ? -d : d;
            const unsigned long long um = __builtin_amdgcn_ballot_w64(vote_ok && dd > 0.0f);
            const unsigned long long dm = __builtin_amdgcn_ballot_w64(vote_ok && dd < 0.0f);
            if (valid) my.sd[rp][k] = make_float2(s, d);
            if (gl == 0) *reinterpret_cast<uint4 *>(my.vote[rp][wave]) =
                make_uint4((unsigned)um, (unsigned)(um >> 32), (unsigned)dm, (unsigned)(dm >> 32));
            if (!committed) {
#pragma unroll
                for (int r = 0; r < 3; ++r) {
                    const int i = k + 192 * r;
                    if (i < kTaps - 1 + kDiscOut) my.x[cur ^ 1][i] = pf[r];
                }
                committed = true;
            }
            if (rewrite) { if (sym_out) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); rewrite = false; }
            STAMP(1);
            lds_barrier();
            STAMP(2);
            rnd++;
            if (!lock_known) {                                // posted by wave 0 before it reached this barrier
                lockv = uni(my.lock_after[(b - 1 - b0) & 1]);
                thresh = lockv ? 80 : 10;
                lock_known = true;
            }
            // ---- replicated control: first crossing over the (up to) three segments
            int base = thr, kcross = -1, tcross = 0;
            const int nseg = (rem + 63) >> 6;
            unsigned long long U[3], D[3];
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const uint4 w = *reinterpret_cast<const uint4 *>(my.vote[rp][j]);
                U[j] = uni64(((unsigned long long)w.y << 32) | w.x);
                D[j] = uni64(((unsigned long long)w.w << 32) | w.z);
            }
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                if (j < nseg && kcross < 0) {
                    // vote counter after the tick of instant 64 j + lane
                    const int tk = base + (int)__builtin_popcount((unsigned)U[j] & incl_lo) + (int)__builtin_popcount((unsigned)(U[j] >> 32) & incl_hi)
                                        - (int)__builtin_popcount((unsigned)D[j] & incl_lo) - (int)__builtin_popcount((unsigned)(D[j] >> 32) & incl_hi);
                    const int kk = 64 * j + gl;
                    const bool vok = (kk < rem) && (p + 2 * kk + 1 < kDiscOut);
                    const unsigned long long cr = __builtin_amdgcn_ballot_w64(vok && (tk > thresh || tk < -thresh));
                    if (cr) {
                        const int kl = (int)__ffsll((long long)cr) - 1;
                        kcross = 64 * j + kl;
                        tcross = __builtin_amdgcn_readlane(tk, kl);
                    } else
                        base += (int)__popcll(U[j]) - (int)__popcll(D[j]);
                }
            }
            const int naccept = (kcross >= 0) ? kcross + 1 : rem;
            STAMP(3);
            // symbols: wave 0 moves the accepted ones into its ring; each wave streams its own out
            if (wave == 0) {
#pragma unroll
                for (int r = 0; r < 3; ++r) {
                    const int kk = gl + 64 * r;
                    if (kk < naccept && m_idx + kk >= 0) my.H[(hp + m_idx + kk) & (kTriRing - 1)] = my.sd[rp][kk].x;
                }
            }
            if (sym_out && k < naccept && m_idx + k >= 0) sym_out[m_idx + k] = s;
            {
                const float2 last = my.sd[rp][naccept - 1];   // sum / dif as the reference leaves them (static, :78)
                sum = unif(last.x); dif = unif(last.y);
            }
            m_idx += naccept;
            if (kcross >= 0) {
                thr = 0; clk = 0;
                if (tcross > thresh) {
                    index = (index + 1 == kPhases) ? 0 : index + 1;
                    if (index == 0) {
                        clk = 1;
                        if (wave == 0 && m_idx >= 0 && gl == 0) my.H[(hp + m_idx) & (kTriRing - 1)] = 0.0f;
                        if (sym_out && wave == 0 && m_idx >= 0 && gl == 0) sym_out[m_idx] = 0.0f;
                        m_idx++;
                    }
                } else {
                    index = (index == 0) ? kPhases - 1 : index - 1;
                    if (index == kPhases - 1) { clk = 1; m_idx--; rewrite = true; }
                }
                p = p + 2 * kcross + 2;
            } else {
                thr = base;
                const int ilast = p + 2 * (rem - 1);
                if (ilast + 1 < kDiscOut) { clk = 0; p = ilast + 2; }
                else { clk = 1; p = kDiscOut; }
            }
        }
        const int n = m_idx > 0 ? m_idx : 0;
        STAMP(4);
        if (!committed) {
            // a block made of vote ticks only cannot happen (384 inputs), but keep the hand-off unconditional
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                const int i = k + 192 * r;
                if (i < kTaps - 1 + kDiscOut) my.x[cur ^ 1][i] = pf[r];
            }
        }
        // a symbol taken back by a downward wrap at the very end of the call sits in the stream behind the last
        // counted one (the reference's out[] holds it too, but it is not a symbol): clear the slot, behind the
        // store of whichever wave wrote it
        if (sym_out && rewrite && b + 1 == bend) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            lds_barrier();
            if (wave == 0 && gl == 0) sym_out[n] = 0.0f;
        }
        if (sym_out) sym_out += n;
        sym_total += n;

        if (wave == 0) {
            wave_fence();
            if (nsyms && gl == 0) nsyms[(size_t)chan * nblk + b] = n;
            // ---- framer (m17_rx_frame.cpp:126-177) over ring symbols hp .. hp+n-1
            int pos = 0;
            while (pos < n) {
                if (flock) {
                    const int cnt = min(kFrameSyms - fclk, n - pos);
                    fclk += cnt; pos += cnt;
                    if (fclk == kFrameSyms) {
                        fclk = 0;
                        const int fs = hp + pos - kFrameSyms;               // the frame sits in the ring, in place
                        float v[8];
#pragma unroll
                        for (int i = 0; i < 8; ++i) v[i] = my.H[(fs + i) & (kTriRing - 1)];
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
                            for (int q = gl; q < kFrameSyms; q += 64) fd[q] = my.H[(fs + q) & (kTriRing - 1)];
                        }
                        nrec++;
                        if (unlock) {
                            flock = 0;
                            // reset_sync(): the next hunt windows must see zeros behind them
                            wave_fence();
                            if (gl < 8) my.H[(hp + pos - 8 + gl) & (kTriRing - 1)] = 0.0f;
                            wave_fence();
                        }
                    }
                } else {
                    // hunt: candidate symbol j = pos+gl, window = ring [hp+j-7, hp+j]
                    const int jc = pos + gl;
                    const bool cand = jc < n;
                    const int jj = cand ? jc : pos;
                    float v[8];
#pragma unroll
                    for (int i = 0; i < 8; ++i) v[i] = my.H[(hp + jj - 7 + i) & (kTriRing - 1)];
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
            block_count++;
            if (gl == 0) my.lock_after[(b - b0) & 1] = flock;
            wave_fence();
        }
        hp += n;
        STAMP(5);
    }
#ifdef M17_STAMPS
    if (chan == 7 && gl == 0 && wave < 2) for (int i = 0; i < 6; ++i) g_stamps[6 * wave + i] = acc_[i];
#endif

    // ---- store state in the reference's layout (wave 0 holds all of it; the last block's inputs are in x[last])
    if (wave == 0) {
        const float *xl = my.x[(bend - 1 - b0) & 1];
        if (gl == 0) {
            cs.clk = clk; cs.thr = thr; cs.index = index; cs.sum = sum; cs.dif = dif; cs.buff[0] = 0.0f;
            cs.flock = flock; cs.fclk = fclk; cs.ferr = ferr; cs.block_count = block_count; cs.sym_total = sym_total;
            if (counts) counts[chan] = nrec;
        }
        if (gl < kTaps - 1) cs.buff[gl + 1] = xl[kDiscOut + gl];
        if (flock) { for (int q = gl; q < kFrameSyms; q += 64) cs.fsym[q] = my.H[(hp - fclk + q) & (kTriRing - 1)]; }
        else if (gl < 8) cs.sync[gl] = my.H[(hp - 8 + gl) & (kTriRing - 1)];
    }
}

} // namespace m17dev
