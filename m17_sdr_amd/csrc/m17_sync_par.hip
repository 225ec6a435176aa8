// m17_sync_par.hip -- k_sync_frame_par: timing recovery of ONE channel on EIGHT waves of one workgroup -- six
// filter waves, one control wave, one framer wave -- for launches with so few channels (up to 1,024) that the chip
// has a SIMD per channel to spare and the serial latency of a channel's blocks is the whole stage.
//
// Reference: m17_rx_sync_samples (m17_rx_sync.cpp:77-99), m17_rx_sym (m17_rx_frame.cpp:126-177).
//
// What is serial in the reference's loop and what is not.  The state that runs through a block is (m_clk, m_thr,
// m_index).  The filter outputs of an instant -- (sum, dif) = the 31-tap matched / derivative pair at input sample i
// under polyphase branch m_index -- depend on the branch and on where the instants fall (even or odd samples; a
// wrap of the branch 39 <-> 0 shifts them by one), NOT on m_thr.  The vote of an instant is the sign of
// (sum < 0 ? -dif : dif): a function of the same two things.  So for a HYPOTHESIS (branch, parity) the 192 outputs
// and votes of a block are independent of each other and of the loop state, and the loop itself is a walk over
// vote bit masks: add votes to m_thr until it leaves [-thresh, thresh], step the branch, go on behind the crossing
// under the new branch's masks.
//   filter waves  : wave u computes the 64 instants of one third of the block under one hypothesis -- the current
//                   branch and the neighbour the counter is heading for (a tracked signal dithers between two
//                   branches; the noiseless bench signal does so every 81 votes) -- with the branch's 62 tap values
//                   in SGPRs (m17_sync_wave.hip; loaded inside the filter statement here) and stores (sum, dif) and the two vote masks in LDS;
//   control wave  : walks the masks with scalar instructions exactly as k_sync_frame_wave does behind its filter
//                   (popcounts; v_mbcnt prefix counts only where a crossing is possible), copies the accepted sums
//                   into the symbol ring, and when the walk needs a hypothesis that was not computed (the branch
//                   moved on, or wrapped) has the filter waves compute it, and its likely successor, for the rest
//                   of the block;
//   framer wave   : k_sync_frame_duo's, one block behind, with the same lock-flag speculation in the control wave.
// Results are the serial results: every accepted output was computed under the branch and parity the walk was in.
// A block then costs one filter round (all six waves side by side) plus the walk, instead of one round per 64
// accepted instants or per crossing, one after the other on one wave (k_sync_frame_duo: 5.2 rounds per block on the
// bench signal, 3.1 us; a lone wave issues one instruction per 5.1 cycles).
//
// All waits are LDS mailbox polls between waves of one workgroup (resident together by construction); every wave
// leaves through the control wave's EXIT job or its own block count: no wait can outlive the kernel.
#pragma clang fp contract(off)

namespace m17dev {

#ifndef M17_PAR_SLEEP
#define M17_PAR_SLEEP 1
#endif
constexpr int kParRing    = 1024;              // as kDuoRing: the framer reads one block behind
constexpr int kParWorkers = 6;
constexpr int kParSlots   = 3;                 // hypotheses held per block: the one left, the current one, its successor
constexpr int kParSegs    = kFrameSyms / 64;   // 192 instants per block and parity = 3 rounds of 64
// x ring: head (the 30 samples in front of region 0) + three block regions; region r of block lb % 3 starts at
// float 32 + 384 r, so the delay line in front of regions 1 and 2 is the tail of the region before them, in place,
// and the tail of region 2 is copied into the head.  xx(r)[k] = xr[2 + 384 r + k] is sample k - 30 of the block.
constexpr int kParXr      = 32 + 3 * kDiscOut;

struct ParChan {
    float xr[kParXr];                                    // 4,736 B
    float H[kParRing];                                   // symbol ring: control wave writes, framer wave reads
    v2f   sd[kParSlots][kFrameSyms];                     // (sum, dif) of every instant of a hypothesis
    unsigned long long mask[kParSlots][kParSegs][2];     // votes up / down, one bit per instant
    int   unit[8];                                       // job descriptor per filter wave
    int   done[8];                                       // job number each filter wave has finished
    int   job_seq;                                       // job number posted by the control wave
    int   nsym[4], lock_after[4], tim_blk, frm_blk;      // framer mailbox (k_sync_frame_duo)
    int   pad[5];
};

// job descriptor bits
constexpr int PAR_VALID = 1 << 30, PAR_EXIT = 1 << 29, PAR_NEWBLOCK = 1 << 28;
__device__ __forceinline__ int par_unit(int idx, int q, int slot, int seg, int region)
{
    return idx | (q << 6) | (slot << 7) | (seg << 9) | (region << 11);
}

__global__ __launch_bounds__(512, 8)
void k_sync_frame_par(const float *__restrict__ disc,     // [C][nblk][384]
                      const float *__restrict__ offs,     // [C][nblk] or null (already DC-free)
                      ChanState *__restrict__ st, int C, int nblk, int mode,
                      m17gpu_rec_dev *__restrict__ recs, int rec_cap, int32_t *__restrict__ counts,
                      float *__restrict__ syms, int32_t *__restrict__ nsyms,
                      float *__restrict__ fsym, int b0, int bcount)
{
    constexpr int RM = kParRing - 1;
    __shared__ __attribute__((aligned(16))) ParChan my;
    const int wave = uni((int)(threadIdx.x >> 6)), gl = lane_id();
    const int chan = (int)blockIdx.x;                    // one channel per workgroup: grid = C
    ChanState &cs = st[chan];
    const float *dsrc = disc + ((size_t)chan * nblk + b0) * kDiscOut;          // block lb of this launch at dsrc + 384 lb
    const float *osrc = offs ? offs + (size_t)chan * nblk + b0 : nullptr;

    // ---- prologue, all waves: block 0 of the launch into region 0, the delay line into the head
    {
        const int t = (int)threadIdx.x;
        if (t < kDiscOut) {
            float v = __builtin_nontemporal_load(&dsrc[t]);
            if (osrc) v = v - osrc[0];                                          // out[i] - offset (m17_dsp.cpp:217-219)
            my.xr[32 + t] = v;
        } else if (t - kDiscOut < kTaps - 1) {
            my.xr[2 + (t - kDiscOut)] = cs.buff[1 + (t - kDiscOut)];
        }
        if (t < 8) { my.done[t] = 0; my.unit[t] = 0; }
        if (t == 0) { my.job_seq = 0; my.tim_blk = 0; my.frm_blk = 0; }
    }
    __syncthreads();                                     // the only workgroup barrier

    if (wave == 7) {
        framer_wave<kParRing>(my, cs, chan, gl, nblk, mode, recs, rec_cap, counts, syms, nsyms, fsym, b0, bcount);
        return;
    }

    if (wave >= 1) {
        // =========================== filter waves ===========================
        // Wave u also owns samples [64u, 64u + 64) of every block on their way from HBM into the x ring: block lb + 1
        // is stored during the first job of block lb (its region is not read by that block's filters), block lb + 2
        // is requested right behind that.
        const int u = wave - 1, smp = 64 * u + gl;
        int lb = 0, seq = 0;
        float pf = 0.0f, pfo = 0.0f;
        if (1 < bcount) { pf = __builtin_nontemporal_load(&dsrc[kDiscOut + smp]); pfo = osrc ? osrc[1] : 0.0f; }
        for (;;) {
            ++seq;
            while (lds_peek(&my.job_seq) < seq) __builtin_amdgcn_s_sleep(M17_PAR_SLEEP);
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
            const int d = uni(lds_peek(&my.unit[u]));
            if (d & PAR_EXIT) break;
            if (d & PAR_NEWBLOCK) {
                if (lb + 1 < bcount) {
                    const int r = (lb + 1) % 3;
                    const float v = osrc ? pf - pfo : pf;                       // out[i] - offset
                    my.xr[32 + kDiscOut * r + smp] = v;
                    if (r == 2 && smp >= kDiscOut - (kTaps - 1)) my.xr[2 + smp - (kDiscOut - (kTaps - 1))] = v;
                }
                if (lb + 2 < bcount) {
                    pf = __builtin_nontemporal_load(&dsrc[(size_t)(lb + 2) * kDiscOut + smp]);
                    pfo = osrc ? osrc[lb + 2] : 0.0f;
                }
                ++lb;
            }
            if (d & PAR_VALID) {
                const int idx = d & 63, q = (d >> 6) & 1, slot = (d >> 7) & 3, seg = (d >> 9) & 3, region = (d >> 11) & 3;
                // instant j = 64 seg + lane of parity q is input sample q + 2j: window xx[q + 2j .. q + 2j + 30]
                const unsigned xa = (unsigned)(uintptr_t)(lds_cfp)(my.xr + 2 + kDiscOut * region + 2 * (64 * seg + gl));
                const v2f a = fir_window_ld_s(&c_tab.tap_pairs[idx][0], xa, q != 0);
                my.sd[slot][64 * seg + gl] = a;
                const float dd = (a.x < 0.0f) ? -a.y : a.y;                     // sync_update, m17_rx_sync.cpp:38-42
                const unsigned long long um = __builtin_amdgcn_ballot_w64(dd > 0.0f);
                const unsigned long long dm = __builtin_amdgcn_ballot_w64(dd < 0.0f);
                if (gl == 0) { my.mask[slot][seg][0] = um; my.mask[slot][seg][1] = dm; }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
            if (gl == 0) lds_poke(&my.done[u], seq);
        }
        return;
    }

    // =========================== control wave ===========================
#ifdef M17_PAR_PRIO
    __builtin_amdgcn_s_setprio(3);
#endif
    int clk = uni(cs.clk), thr = uni(cs.thr), index = uni(cs.index);
    float sum = unif(cs.sum), dif = unif(cs.dif);
    int known_lock = uni(cs.flock);
    int hp = 256, n_job = 0, prev_index = -1;
    int key0 = -1, key1 = -1, key2 = -1;                 // hypothesis held by slot k: idx | q << 6 | first valid segment << 8
    int mv = 0;                                          // lane l: dword l of mask[][][]
#ifdef M17_STAMPS
    // phase accumulators in LDS: 0 filter jobs (post .. done), 1 walk, 2 wait for the framer's lock flag, 3 rest;
    // 8 blocks, 9 jobs, 10 runs
    __shared__ unsigned pstamps[12];
    if (gl < 12) pstamps[gl] = 0;
    unsigned plast_ = (unsigned)__builtin_amdgcn_s_memtime();
#define PSTAMP(i) do { __builtin_amdgcn_sched_barrier(0); const unsigned now_ = (unsigned)__builtin_amdgcn_s_memtime(); \
    __builtin_amdgcn_sched_barrier(0); if (gl == 0) pstamps[i] += now_ - plast_; plast_ = now_; } while (0)
#define PCNT(i) do { if (gl == 0) pstamps[i] += 1; } while (0)
#else
#define PSTAMP(i) do {} while (0)
#define PCNT(i) do {} while (0)
#endif

    auto post = [&](int dsc) {                           // dsc: lane u's descriptor (lanes 0..5)
        if (gl < kParWorkers) lds_poke(&my.unit[gl], dsc);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
        ++n_job;
        if (gl == 0) lds_poke(&my.job_seq, n_job);
    };
    auto wait_done = [&]() {
        for (;;) {
            const int v = (gl < kParWorkers) ? lds_peek(&my.done[gl]) : n_job;
            if (__builtin_amdgcn_ballot_w64(v < n_job) == 0ull) break;
            __builtin_amdgcn_s_sleep(1);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
        mv = (gl < kParSlots * kParSegs * 4) ? lds_peek(reinterpret_cast<const int *>(&my.mask[0][0][0]) + gl) : 0;
    };
    auto fetch_sd = [&](int slot, int j) {               // the carried sum / dif: the last accepted instant's outputs
        const v2f a = my.sd[slot][j];
        sum = unif(a.x); dif = unif(a.y);
    };

    for (int lb = 0; lb < bcount; ++lb) {
        PCNT(8);
        const int region = lb % 3;
        const int s_clk = clk, s_thr = thr, s_index = index, s_prev = prev_index;
        const float s_sum = sum, s_dif = dif;
        int lockv = known_lock, n = 0;
        bool posted = false;
        for (int attempt = 0; attempt < 2; ++attempt) {
            const int thresh = lockv ? 80 : 10;
            int p = 0, m_idx = 0;
            while (p < kDiscOut) {
                if (clk == 1) {
                    // vote tick on the carried sum/dif (sync_update :38-42, m17_sync_adjust :45-72)
                    clk = 0;
                    const float d0 = (sum < 0.0f) ? -dif : dif;
                    if (d0 > 0.0f) thr++;
                    if (d0 < 0.0f) thr--;
                    if (thr > thresh) {
                        prev_index = index;
                        index = (index + 1 == kPhases) ? 0 : index + 1; thr = 0;
                        if (index == 0) { clk = 1; if (m_idx >= 0 && gl == 0) my.H[(hp + m_idx) & RM] = 0.0f; m_idx++; }
                    }
                    if (thr < -thresh) {
                        prev_index = index;
                        thr = 0; index = (index == 0) ? kPhases - 1 : index - 1;
                        if (index == kPhases - 1) { clk = 1; m_idx--; }
                    }
                    p++;
                    continue;
                }
                // ---- a run of instants under (index, q) from instant j0 = p >> 1 (sample p = q + 2 j0)
                const int q = p & 1, j0 = p >> 1, seg = j0 >> 6, sh = j0 & 63;
                const int want = index | (q << 6);
                int s = -1;
                if ((key0 & 0x7F) == want && (key0 >> 8) <= seg && key0 >= 0) s = 0;
                if ((key1 & 0x7F) == want && (key1 >> 8) <= seg && key1 >= 0) s = 1;
                if ((key2 & 0x7F) == want && (key2 >> 8) <= seg && key2 >= 0) s = 2;
                if (!posted) {
                    // first filter job of the block: the current branch in slot 0, the branch the counter is heading
                    // for (or, with the counter at rest, the one just left) in slot 1, whole block each
                    int other = thr > 0 ? index + 1 : thr < 0 ? index - 1 : (prev_index >= 0 && prev_index != index) ? prev_index : index + 1;
                    other = other >= kPhases ? other - kPhases : other < 0 ? other + kPhases : other;
                    const int hyp = gl < 3 ? index : other;
                    const int sg = gl < 3 ? gl : gl - 3;
                    post(PAR_VALID | PAR_NEWBLOCK | par_unit(hyp, q, gl < 3 ? 0 : 1, sg, region));
                    key0 = want; key1 = other | (q << 6); key2 = -1;
                    posted = true;
                    s = 0;
                    PSTAMP(1); PCNT(9);
                    wait_done();
                    PSTAMP(0);
                } else if (s < 0) {
                    // the walk left the computed hypotheses: this branch, and the next one in the direction of the
                    // step that brought it here, from this segment to the end of the block; the slot of the
                    // hypothesis just left is kept (a dithering loop comes back to it)
                    const int keep = ((key0 & 0x3F) == prev_index && key0 >= 0) ? 0 : ((key1 & 0x3F) == prev_index && key1 >= 0) ? 1 : 2;
                    const int s1 = keep == 0 ? 1 : 0, s2 = keep == 2 ? 1 : 2;
                    int dir = index - prev_index;                               // +-1, or -+39 across the wrap
                    dir = (prev_index < 0) ? 1 : (dir == 1 || dir == -(kPhases - 1)) ? 1 : -1;
                    int nxt = index + dir;
                    const bool nxt_ok = nxt >= 0 && nxt < kPhases;              // beyond the wrap the instants move: not predicted
                    const int hyp = gl < 3 ? index : nxt;
                    const int sg = gl < 3 ? gl : gl - 3;
                    const bool ok = sg >= seg && (gl < 3 || nxt_ok);
                    post((ok ? PAR_VALID : 0) | par_unit(hyp & 63, q, gl < 3 ? s1 : s2, sg, region));
                    const int k1 = want | (seg << 8), k2 = nxt_ok ? (nxt | (q << 6) | (seg << 8)) : -1;
                    if (s1 == 0) key0 = k1; else key1 = k1;
                    if (s2 == 1) key1 = k2; else key2 = k2;
                    s = s1;
                    PSTAMP(1); PCNT(9);
                    wait_done();
                    PSTAMP(0);
                }
                PSTAMP(1); PCNT(10);
                const int nv = min(64 - sh, kFrameSyms - j0);                   // instants of this mask word
                const int nvote = min(nv, (kDiscOut - p) >> 1);                 // ... whose vote tick is inside the block
                const unsigned long long okm = (nvote >= 64) ? ~0ull : ((1ull << nvote) - 1ull);
                const int mb = (s * kParSegs + seg) * 4;
                const unsigned long long umw = (unsigned long long)(unsigned)__builtin_amdgcn_readlane(mv, mb) |
                                               ((unsigned long long)(unsigned)__builtin_amdgcn_readlane(mv, mb + 1) << 32);
                const unsigned long long dmw = (unsigned long long)(unsigned)__builtin_amdgcn_readlane(mv, mb + 2) |
                                               ((unsigned long long)(unsigned)__builtin_amdgcn_readlane(mv, mb + 3) << 32);
                const unsigned long long um = (umw >> sh) & okm, dm = (dmw >> sh) & okm;
                const int nu = (int)__popcll(um), nd = (int)__popcll(dm);
                int naccept = nv, kl = -1, ts_ = 0;
                PSTAMP(2);
                if (thr + nu > thresh || thr - nd < -thresh) {
                    // a crossing is possible in this run: the counter after every tick, first crossing wins
                    const int pu = (int)__builtin_amdgcn_mbcnt_hi((unsigned)(um >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)um, 0u));
                    const int pd = (int)__builtin_amdgcn_mbcnt_hi((unsigned)(dm >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)dm, 0u));
                    const int own = (int)((um >> gl) & 1ull) - (int)((dm >> gl) & 1ull);
                    const int tk = thr + pu - pd + own;
                    const unsigned long long cr = __builtin_amdgcn_ballot_w64(tk > thresh || tk < -thresh) & okm;
                    if (cr) {
                        kl = (int)__builtin_ctzll(cr);
                        naccept = kl + 1;
                        ts_ = __builtin_amdgcn_readlane(tk, kl);
                    }
                }
                PSTAMP(3);
                // accepted sums are symbols (m17_rx_sync.cpp:84-86)
                if (gl < naccept && (m_idx + gl) >= 0) my.H[(hp + m_idx + gl) & RM] = my.sd[s][j0 + (gl < naccept ? gl : 0)].x;
                m_idx += naccept;
                PSTAMP(4);
                const int jlast = j0 + naccept - 1;
                if (kl >= 0) {
                    thr = 0; clk = 0;
                    prev_index = index;
                    if (ts_ > thresh) {
                        index = (index + 1 == kPhases) ? 0 : index + 1;
                        if (index == 0) { clk = 1; if (m_idx >= 0 && gl == 0) my.H[(hp + m_idx) & RM] = 0.0f; m_idx++; }
                    } else {
                        index = (index == 0) ? kPhases - 1 : index - 1;
                        if (index == kPhases - 1) { clk = 1; m_idx--; }
                    }
                    p = p + 2 * kl + 2;
                } else {
                    thr += nu - nd;
                    const int ilast = p + 2 * (nv - 1);
                    if (ilast + 1 < kDiscOut) { clk = 0; p = ilast + 2; }
                    else { clk = 1; p = kDiscOut; }                             // the last vote tick falls into the next block
                }
                // the next tick, if it is one on carried values, and the state behind the launch's last block
                if (clk == 1 || (p >= kDiscOut && lb == bcount - 1)) fetch_sd(s, jlast);
                PSTAMP(5);
            }
            n = m_idx > 0 ? m_idx : 0;
            // ---- the lock flag this block should have seen: after the framer of block lb - 1 (k_sync_frame_duo)
            int actual = known_lock;
            PSTAMP(1);
            if (lb > 0) {
                duo_wait_lds(&my.frm_blk, lb);
                PSTAMP(6);
                actual = uni(lds_peek(&my.lock_after[(b0 + lb - 1) & 3]));
            }
            if (actual == lockv) break;
            lockv = actual;                                                     // mispredicted (lock just changed): walk the block again
            clk = s_clk; thr = s_thr; index = s_index; sum = s_sum; dif = s_dif; prev_index = s_prev;
        }
        known_lock = lockv;
        if (gl == 0) my.nsym[(b0 + lb) & 3] = n;
        duo_post_lds(&my.tim_blk, lb + 1, gl);
        hp += n;
        PSTAMP(6);
    }
    post(PAR_EXIT);
#ifdef M17_STAMPS
    wave_fence();
    if (chan < 4096 && gl < 8) g_chan_stamps[chan][gl] = pstamps[gl < 7 ? gl : 10];
#endif
    // ---- timing state in the reference's layout; m_buff = the last 30 inputs of the launch's last block
    if (gl == 0) { cs.clk = clk; cs.thr = thr; cs.index = index; cs.sum = sum; cs.dif = dif; cs.buff[0] = 0.0f; }
    {
        const int r = (bcount - 1) % 3;
        if (gl < kTaps - 1) cs.buff[gl + 1] = my.xr[32 + kDiscOut * r + kDiscOut - (kTaps - 1) + gl];
    }
}

} // namespace m17dev
