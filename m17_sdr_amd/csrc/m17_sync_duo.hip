// m17_sync_duo.hip -- k_sync_frame_duo: timing recovery and framer of one channel on TWO waves
// of the same workgroup, decoupled by one block (64 lanes per channel; used up to 1,024 channels); as <1> with the
// channel's front end on a third (the FIR stage of small batches as one pipelined kernel).
//
// Reference: m17_rx_sync_samples (m17_rx_sync.cpp:77-99), m17_rx_sym (m17_rx_frame.cpp:126-177).
//
// Why: with 1,024 channels the lane-group kernel has exactly one wave per SIMD, a wave issues one
// instruction per ~8 cycles whatever it does, and a SIMD can issue twice that -- half the issue
// slots idle while every channel's blocks run strictly in order.  The only thing the timing loop
// takes from the framer is the lock flag (threshold 80 / 10, m17_rx_sync.cpp:50 via m17_rx_lock()),
// and that changes about twice per transmission.  So:
//   timing wave : runs block b under the lock flag it last saw (after the framer of b-2), then
//                 waits for the framer of b-1 -- normally long done -- and, only if the flag
//                 differs, restores its five state variables and runs block b again.  It then
//                 publishes the block's symbols (in the channel's LDS ring) and moves on.
//   framer wave : one block behind: symbol stream out, frame sync checks, records, frame symbols.
// Both waves are resident by construction (same workgroup), every wait is on a flag the other
// wave sets without waiting for anything later, and both run exactly `bcount` blocks: no wait can
// outlive the kernel.  Results are the serial results -- a misprediction is recomputed, not patched.
#pragma clang fp contract(off)

namespace m17dev {

constexpr int kDuoRing = 1024;                 // 193 (block being written) + 193 + 192 (frame the framer may still read) < 1024

struct DuoChan {                               // LDS of one channel
    float x[kTaps - 1 + kDiscOut + 2];         // delay-line history (30) + this block's 384 inputs   (timing wave)
    float H[kDuoRing];                         // symbol ring: timing wave writes, framer wave reads
    int   nsym[4];                             // symbols of block b at [b & 3]
    int   lock_after[4];                       // lock flag after the framer of block b at [b & 3]
    int   tim_blk, frm_blk;                    // blocks published / framed since b0
    int   fe_rows;                             // PIPE: blocks since b0 whose discriminator rows the front-end wave has stored
    int   pad[5];
};

// Mailbox words are read and written as LDS words (ds_read / ds_write), not through generic pointers: a
// volatile generic access compiles to flat_load ... sc0 sc1 + s_waitcnt vmcnt(0), several hundred cycles each.
typedef __attribute__((address_space(3))) volatile int lds_vint;
__device__ __forceinline__ int lds_peek(const int *p) { return *(const lds_vint *)p; }
__device__ __forceinline__ void lds_poke(int *p, int v) { *(lds_vint *)p = v; }

// the same wait with an LDS-only acquire, for a waiter that reads only LDS behind the flag and has global loads in
// flight (the timing wave's prefetch): a full workgroup-scope fence would also drain vmcnt
template <int SLEEP = 1>
__device__ __forceinline__ void duo_wait_lds(const int *flag, int need)
{
    while (lds_peek(flag) < need) __builtin_amdgcn_s_sleep(SLEEP);
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}
template <int SLEEP = 1>
__device__ __forceinline__ void duo_wait(const int *flag, int need)
{
    // the poll shares the SIMD's issue slots with the other wave: the framer wave, which has ~2,700 cycles of
    // slack per block, sleeps longer between looks than the timing wave
    while (lds_peek(flag) < need) __builtin_amdgcn_s_sleep(SLEEP);
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}
__device__ __forceinline__ void duo_post(int *flag, int value, int lane)
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    if (lane == 0) lds_poke(flag, value);
}

// publish with an LDS-only release: everything the reader takes from this wave is in LDS
__device__ __forceinline__ void duo_post_lds(int *flag, int value, int lane)
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    if (lane == 0) lds_poke(flag, value);
}

// A block's DC offset.  PIPE: the offsets of thirty-two blocks share a 128-byte line, i.e. two front-end tiles do: a line this
// wave read while only its first half had been written must not be served again from the CU's vector cache when the
// second half is there -- the load is an agent-scope one (past that cache).  The discriminator rows need nothing of the
// kind: a row is twelve whole lines, written before it is announced and read only after.
template <int PIPE>
__device__ __forceinline__ float duo_offs(const float *p)
{
    if constexpr (PIPE) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else return *p;
}

// PIPE (round 5; k_rx_trio, option fir_impl 5): a THIRD wave per channel is the channel's front end -- sixteen of its
// blocks per tile (frontend_tile), rows stored to the workspace and announced through fe_rows -- so the front end of
// blocks 16 .. runs under the timing loop of blocks 0 .. instead of in a kernel in front of it.  At 1,024 channels
// both kernels are latency-bound (one channel per SIMD slot, its blocks strictly in order): what the stand-alone front
// end takes (0.10 of 0.24 ms at 1,024 x 50) is time in which no timing wave runs.  Same flag discipline as above: the
// front-end wave waits for nothing, the timing wave waits only for rows the front-end wave is on its way to.
template <int PIPE>
__global__ __launch_bounds__(PIPE ? 768 : 512)
void k_sync_frame_duo(const float *__restrict__ disc,     // [C][nblk][384]
                      const float *__restrict__ offs,     // [C][nblk] or null (already DC-free)
                      ChanState *__restrict__ st, int C, int nblk, int mode,
                      m17gpu_rec_dev *__restrict__ recs, int rec_cap, int32_t *__restrict__ counts,
                      float *__restrict__ syms, int32_t *__restrict__ nsyms,
                      float *__restrict__ fsym, int b0, int bcount,
                      const uint4 *__restrict__ iq = nullptr, float *__restrict__ disc_w = nullptr, float *__restrict__ offs_w = nullptr)
{
    constexpr int LPC = 64;
    __shared__ __attribute__((aligned(16))) float taps[kPhases * 64];      // (matched, derivative) pairs per branch
    __shared__ __attribute__((aligned(16))) DuoChan chs[4];
    __shared__ __attribute__((aligned(16))) uint32_t fe_raw[PIPE ? 4 : 1][PIPE ? 16 * FQ_STRIDE : 4];
    __shared__ __attribute__((aligned(16))) float fe_out[PIPE ? 4 : 1][PIPE ? 16 * FQ_STRIDE : 4];
    const int wave = (int)(threadIdx.x >> 6), gl = lane_id();
    const int w = wave & 3;
    const bool is_framer = (wave >> 2) == 1;
    for (int q = (int)threadIdx.x; q < kPhases * 32; q += (PIPE ? 768 : 512)) {
        taps[2 * q] = (&c_tab.mf[0][0])[q];
        taps[2 * q + 1] = (&c_tab.md[0][0])[q];
    }
    if (threadIdx.x < 4) { chs[threadIdx.x].tim_blk = 0; chs[threadIdx.x].frm_blk = 0; chs[threadIdx.x].fe_rows = 0; }
    __syncthreads();                                    // the only workgroup barrier
    const int chan = (int)blockIdx.x * 4 + w;
    if (chan >= C) return;                              // all waves of the channel leave together
    DuoChan &my = chs[w];
    ChanState &cs = st[chan];
    const int bend = b0 + bcount;
    int *tim_blk = &my.tim_blk, *frm_blk = &my.frm_blk;
    if constexpr (PIPE) {
        auto rows_from = [&](int r0) {
            return [=](int i, bool &valid) { valid = r0 + i < bcount; return chan * nblk + b0 + (valid ? r0 + i : bcount - 1); };
        };
        // Short calls (up to eight blocks): four-row tiles (frontend_quick4p: 23 us on a lone wave against 40 for a
        // sixteen-row tile, whatever that holds), rows 0-3 on the front-end wave and rows 4-7 on the framer wave, which has
        // nothing to frame yet.  Longer calls gain nothing from a quick start -- the timing wave would catch up with the
        // front-end wave inside its first sixteen-row tile and wait there (measured: three four-row tiles in front of the
        // sixteen-row ones +4-5 % at 12-50 blocks, profiles/r05_three_wave_fir_stage_1024.txt).
        const bool short_call = bcount <= 8;
        if (wave >= 8) {
            // =========================== front-end wave ===========================
            if (short_call) {
                frontend_quick4p(iq, st, disc_w, offs_w, nblk, 1, rows_from(0), gl);
                duo_post(&my.fe_rows, min(4, bcount), gl);
                return;
            }
            for (int r0 = 0; r0 < bcount; r0 += 16) {
                frontend_tile(iq, st, disc_w, offs_w, nblk, 1, rows_from(r0), fe_raw[w], fe_out[w], gl);
                duo_post(&my.fe_rows, min(r0 + 16, bcount), gl);
            }
            return;
        }
        if (is_framer && short_call && bcount > 4) {
            frontend_quick4p(iq, st, disc_w, offs_w, nblk, 1, rows_from(4), gl);
            duo_wait(&my.fe_rows, 4);                   // one counter, two writers: this one second
            duo_post(&my.fe_rows, bcount, gl);
        }
    }
    int fe_seen = PIPE ? 0 : 0x7FFFFFFF;               // rows known to be there (timing wave)

#ifdef M17_STAMPS
    unsigned long long acc_[12] = {0,0,0,0,0,0,0,0,0,0,0,0}, last_ = __builtin_amdgcn_s_memtime();
    __builtin_amdgcn_s_waitcnt(0xC07F);
#endif
    if (!is_framer) {
        // =========================== timing wave ===========================
        // the control state of a channel is wave-uniform: read into scalar registers, so that the branches on it are scalar
        // branches and its arithmetic the scalar unit's (as loaded -- per lane -- every compare below was an exec-mask branch)
        int clk = uni(cs.clk), thr = uni(cs.thr), index = uni(cs.index);
        float sum = unif(cs.sum), dif = unif(cs.dif);
        int known_lock = uni(cs.flock);
        int hp = 256;
        for (int q = gl; q < kTaps - 1; q += LPC) my.x[q] = cs.buff[q + 1];
        const float *dsrc = disc + (size_t)chan * nblk * kDiscOut;
        const float *osrc = offs ? offs + (size_t)chan * nblk : nullptr;
        if constexpr (PIPE) { duo_wait(&my.fe_rows, 1); fe_seen = uni(lds_peek(&my.fe_rows)); }
        {
            const float off = osrc ? duo_offs<PIPE>(&osrc[b0]) : 0.0f;
            for (int q = gl; q < kDiscOut; q += LPC) {
                float v = dsrc[(size_t)b0 * kDiscOut + q];
                if (osrc) v = v - off;                               // out[i] - offset (m17_dsp.cpp:217-219)
                my.x[kTaps - 1 + q] = v;
            }
        }
        wave_fence();
        float4 tp[16];                                                // 32 tap pairs of the current branch, kept across blocks
        int tap_index = -1;
        bool calm = false, crossed = false;                           // the last block / this block saw a branch step
        for (int b = b0; b < bend; ++b) {
            // next block's input: loads issued now, committed at the end of the block
            constexpr int PF = kDiscOut / LPC;
            float pf[PF];
            float noff = 0.0f;
            if (b + 1 < bend) {
                if constexpr (PIPE) {
                    if (b + 1 - b0 >= fe_seen) { duo_wait(&my.fe_rows, b + 2 - b0); fe_seen = uni(lds_peek(&my.fe_rows)); }
                }
                const float *nx = dsrc + (size_t)(b + 1) * kDiscOut;
                noff = osrc ? duo_offs<PIPE>(&osrc[b + 1]) : 0.0f;
#pragma unroll
                for (int r = 0; r < PF; ++r) pf[r] = __builtin_nontemporal_load(&nx[gl + LPC * r]);   // read once
            }
            const int s_clk = clk, s_thr = thr, s_index = index;
            const float s_sum = sum, s_dif = dif;
            int lockv = known_lock, n = 0;
            for (int attempt = 0; attempt < 2; ++attempt) {
                // ---- timing recovery in rounds of 64 instants; x[i .. i+30] is the delay line at input i
                const int thresh = lockv ? M17_LIT_THRESH_LOCKED : M17_LIT_THRESH_UNLOCKED;
                int p = 0, m_idx = 0;
                crossed = false;
                // vote tick on the carried sum/dif (sync_update :38-42, m17_sync_adjust :45-72): the first input of a block
                // whose predecessor ended on a filter instant, and the input behind a wrap of the branch
                auto tick = [&]() {
                    clk = 0;
                    const float d0 = (sum < 0.0f) ? -dif : dif;
                    if (d0 > 0.0f) thr++;
                    if (d0 < 0.0f) thr--;
                    if (thr > thresh) {
                        index = (index + 1 == kPhases) ? 0 : index + 1; thr = 0; crossed = true;
                        if (index == 0) { clk = 1; if (m_idx >= 0 && gl == 0) my.H[(hp + m_idx) & (kDuoRing - 1)] = 0.0f; m_idx++; }
                    }
                    if (thr < -thresh) {
                        thr = 0; index = (index == 0) ? kPhases - 1 : index - 1; crossed = true;
                        if (index == kPhases - 1) { clk = 1; m_idx--; }
                    }
                    p++;
                };
                while (clk == 1 && p < kDiscOut) tick();
                // A calm block at once.  When the previous block went by without a branch step, the loop most likely
                // sits on its branch for this one too: its three rounds are filtered back to back -- symbols stored as
                // they come, only the vote counts kept -- and committed together if the counter stays inside
                // [-thresh, thresh] throughout (per round the two scalar compares of the rounds below, the per-tick counts
                // only if those leave a doubt).  If it does not, nothing has happened yet: the rounds below start over
                // from the block's first instant and overwrite the symbols.  (This wave is alone on its SIMD slot: what it
                // saves in instructions it saves in time; k_sync_frame_wave, bound otherwise, loses with the same path.)
                if (calm && p <= 1 && m_idx == 0) {
                    if (tap_index != index) {
                        const float4 *t4 = reinterpret_cast<const float4 *>(&taps[64 * index]);
#pragma unroll
                        for (int q = 0; q < 16; ++q) tp[q] = t4[q];
                        tap_index = index;
                    }
                    int t = thr;
                    bool ok = true;
                    v2f a = {0.0f, 0.0f};
#pragma unroll 1
                    for (int r = 0; r < 3 && ok; ++r) {
                        a = fir_pair(my.x + p + 128 * r + 2 * gl, tp);
                        const float dd = (a.x < 0.0f) ? -a.y : a.y;
                        // with the instants on the odd inputs, the vote of input 383's instant is cast in the next block
                        const unsigned long long okm = (r == 2 && p != 0) ? 0x7FFFFFFFFFFFFFFFull : ~0ull;
                        const unsigned long long um = __builtin_amdgcn_ballot_w64(dd > 0.0f) & okm;
                        const unsigned long long dm = __builtin_amdgcn_ballot_w64(dd < 0.0f) & okm;
                        const int nu = (int)__popcll(um), nd = (int)__popcll(dm);
                        if (t + nu > thresh || t - nd < -thresh) {
                            const int pu = (int)__builtin_amdgcn_mbcnt_hi((unsigned)(um >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)um, 0u));
                            const int pd = (int)__builtin_amdgcn_mbcnt_hi((unsigned)(dm >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)dm, 0u));
                            const int own = (int)((um >> gl) & 1ull) - (int)((dm >> gl) & 1ull);
                            const int tk = t + pu - pd + own;
                            if (__builtin_amdgcn_ballot_w64(tk > thresh || tk < -thresh) & okm) ok = false;
                        }
                        t += nu - nd;
                        my.H[(hp + 64 * r + gl) & (kDuoRing - 1)] = a.x;
                    }
                    if (ok) {
                        thr = t;
                        m_idx = kFrameSyms;
                        sum = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(a.x), 63));
                        dif = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(a.y), 63));
                        clk = p;                                      // odd inputs: the last vote tick falls into the next block
                        p = kDiscOut;
                    }
                }
                while (p < kDiscOut) {
                    if (tap_index != index) {
                        const float4 *t4 = reinterpret_cast<const float4 *>(&taps[64 * index]);
#pragma unroll
                        for (int q = 0; q < 16; ++q) tp[q] = t4[q];
                        tap_index = index;
                    }
                    // lane g takes the instant at input p + 2g; near the end of the block the upper lanes run past it into
                    // the ring that follows x[] in the channel's LDS, and nothing of theirs is used (k_sync_frame_wave)
                    const v2f a = fir_pair(my.x + p + 2 * gl, tp);
                    const float s = a.x, d = a.y;
                    const int rem = kDiscOut - p;                     // >= 1
                    const int nv = min(LPC, (rem + 1) >> 1);          // filter instants of this round
                    const int nvote = min(LPC, rem >> 1);             // ... whose vote tick p + 2g + 1 is inside the block
                    const unsigned long long okm = (nvote >= 64) ? ~0ull : ((1ull << nvote) - 1ull);
                    const float dd = (s < 0.0f) ? -d : d;
                    const unsigned long long um = __builtin_amdgcn_ballot_w64(dd > 0.0f) & okm;
                    const unsigned long long dm = __builtin_amdgcn_ballot_w64(dd < 0.0f) & okm;
                    const int nu = (int)__popcll(um), nd = (int)__popcll(dm);
                    int naccept = nv, kl = -1, ts = 0;
                    // the counter can only leave [-thresh, thresh] in this round if all the votes of one sign could
                    // carry it there: otherwise the round is accepted whole on two scalar compares (k_sync_frame_wave)
                    if (uni(thr) + nu > thresh || uni(thr) - nd < -thresh) {
                        const int pu = (int)__builtin_amdgcn_mbcnt_hi((unsigned)(um >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)um, 0u));
                        const int pd = (int)__builtin_amdgcn_mbcnt_hi((unsigned)(dm >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)dm, 0u));
                        const int own = (int)((um >> gl) & 1ull) - (int)((dm >> gl) & 1ull);
                        const int tk = thr + pu - pd + own;
                        const unsigned long long cr = __builtin_amdgcn_ballot_w64(tk > thresh || tk < -thresh) & okm;
                        if (cr) {
                            kl = (int)__builtin_ctzll(cr);
                            naccept = kl + 1;
                            ts = __builtin_amdgcn_readlane(tk, kl);
                        }
                    }
                    if (gl < naccept && (m_idx + gl) >= 0) my.H[(hp + m_idx + gl) & (kDuoRing - 1)] = s;
                    m_idx += naccept;
                    // (wave-uniform lane numbers: v_readlane, not the LDS crossbar of a shuffle)
                    sum = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(s), uni(naccept - 1)));
                    dif = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(d), uni(naccept - 1)));
                    if (kl >= 0) {
                        thr = 0; clk = 0; crossed = true;
                        if (ts > thresh) {
                            index = (index + 1 == kPhases) ? 0 : index + 1;
                            if (index == 0) { clk = 1; if (m_idx >= 0 && gl == 0) my.H[(hp + m_idx) & (kDuoRing - 1)] = 0.0f; m_idx++; }
                        } else {
                            index = (index == 0) ? kPhases - 1 : index - 1;
                            if (index == kPhases - 1) { clk = 1; m_idx--; }
                        }
                        p = p + 2 * kl + 2;
                        while (clk == 1 && p < kDiscOut) tick();       // a wrap: the next input is a vote tick again
                    } else {
                        thr += nu - nd;
                        p += 2 * nv;                                   // behind the last instant's vote tick ...
                        clk = p > kDiscOut ? 1 : 0;                    // ... which falls into the next block when that instant is input 383
                        p = min(p, kDiscOut);
                    }
                }
                n = m_idx > 0 ? m_idx : 0;
                // ---- the lock flag this block should have seen: after the framer of block b-1
                int actual = known_lock;
#ifdef M17_STAMPS
                const unsigned long long before_ = acc_[0];
#endif
                STAMP(0);
#ifdef M17_STAMPS
                // per-block work of the first 64 channels' timing waves (scripts/exp_stamps_duo.py)
                if (chan < 64 && gl == 0 && b - b0 < 64) {
                    g_chan_stamps[chan * 64 + b - b0][6] = acc_[0] - before_;
                    g_chan_stamps[chan * 64 + b - b0][7] = (unsigned long long)(calm ? 1 : 0) | ((unsigned long long)(crossed ? 1 : 0) << 1) | ((unsigned long long)lockv << 2);
                }
#endif
                if (b > b0) {
                    duo_wait_lds(frm_blk, b - b0);
                    STAMP(1);
                    actual = uni(lds_peek(&my.lock_after[(b - 1) & 3]));
                }
                if (actual == lockv) break;
                lockv = actual;                                       // mispredicted (lock just changed): run the block again
                clk = s_clk; thr = s_thr; index = s_index; sum = s_sum; dif = s_dif;
            }
            known_lock = lockv;
            calm = !crossed;
            if (gl == 0) my.nsym[b & 3] = n;
            duo_post_lds(tim_blk, b - b0 + 1, gl);
            hp += n;
            // delay line: last 30 inputs; then the prefetched block moves in
            {
                const float keep_x = (gl < kTaps - 1) ? my.x[kDiscOut + gl] : 0.0f;
                wave_fence();
                if (gl < kTaps - 1) my.x[gl] = keep_x;
                if (b + 1 < bend) {
#pragma unroll
                    for (int r = 0; r < PF; ++r)
                        my.x[kTaps - 1 + gl + LPC * r] = osrc ? (pf[r] - noff) : pf[r];     // out[i] - offset
                }
            }
            wave_fence();
        }
#ifdef M17_STAMPS
        if (chan == 0 && gl == 0) { g_stamps[0] = acc_[0]; g_stamps[1] = acc_[1]; }
        if (chan < 4096 && gl == 0) { g_chan_stamps[chan][5] = acc_[0]; g_chan_stamps[chan][4] = acc_[1]; }
#endif
        if (gl == 0) { cs.clk = clk; cs.thr = thr; cs.index = index; cs.sum = sum; cs.dif = dif; cs.buff[0] = 0.0f; }
        for (int q = gl; q < kTaps - 1; q += LPC) cs.buff[q + 1] = my.x[q];
        return;
    }

    // =========================== framer wave ===========================
    m17gpu_rec_dev *crecs = recs ? recs + (size_t)chan * rec_cap : nullptr;
    if (!recs) rec_cap = 0;
    int flock = uni(cs.flock), fclk = uni(cs.fclk), ferr = uni(cs.ferr);
    uint32_t block_count = (uint32_t)uni((int)cs.block_count);
    int nrec = (b0 == 0) ? 0 : uni(counts[chan]);
    int sym_total = (b0 == 0) ? 0 : uni(cs.sym_total);
    int hp = 256;
    RegroupLane<LPC> rg;
    rg.load(gl);
    const unsigned sgn = sync_sign_mask(gl);
    // m_f_sym[0 .. fclk) is the frame in progress: ring [hp - fclk, hp); m_sync is the last 8 symbols.
    // Only positions below hp are written here: hp upward belongs to the timing wave from the start.
    if (flock) { for (int q = gl; q < fclk; q += LPC) my.H[(hp - fclk + q) & (kDuoRing - 1)] = cs.fsym[q]; }
    else if (gl < 8) my.H[(hp - 8 + gl) & (kDuoRing - 1)] = cs.sync[gl];
    float *sym_out = syms ? syms + (size_t)chan * M17_SYM_STRIDE(nblk) + sym_total : nullptr;
    wave_fence();
    for (int b = b0; b < bend; ++b) {
        STAMP(2);
        duo_wait<8>(tim_blk, b - b0 + 1);
        STAMP(3);
        const int n = uni(lds_peek(&my.nsym[b & 3]));
        if (sym_out) {
#pragma unroll
            for (int r = 0; r < (193 + LPC - 1) / LPC; ++r) {
                const int q = gl + LPC * r;
                if (q < n) sym_out[q] = my.H[(hp + q) & (kDuoRing - 1)];
            }
            sym_out += n;
        }
        if (nsyms && gl == 0) nsyms[(size_t)chan * nblk + b] = n;
        sym_total += n;
        // ---- framer (m17_rx_frame.cpp:126-177) over ring symbols hp .. hp+n-1
        int pos = 0;
        while (pos < n) {
            if (flock) {
                const int cnt = min(kFrameSyms - fclk, n - pos);
                fclk += cnt; pos += cnt;
                if (fclk == kFrameSyms) {
                    fclk = 0;
                    const int fs = hp + pos - kFrameSyms;               // the frame sits in the ring, in place
                    const SyncResult r = sync_check_lanes8(my.H[(fs + (gl & 7)) & (kDuoRing - 1)], sgn);
                    uint32_t flags = 0;
                    bool parse = false, unlock = false;
                    if (r.type == 5) { flags |= M17_F_EOT; unlock = true; }
                    else if (sync_accept(r, true)) { flags |= M17_F_SYNC_OK; parse = true; ferr = 0; }
                    else {
                        ferr++;
                        if (ferr > M17_LIT_N_FERROR) { flags |= M17_F_LOST; unlock = true; }
                        else parse = true;
                    }
                    if (parse && mode == 1) flags |= M17_F_PARSED;
                    const uint32_t w0 = (uint32_t)r.type | ((uint32_t)r.votes << 8) | ((uint32_t)(ferr & 0xFF) << 24);
                    emit_record_wave(crecs, rec_cap, nrec, gl, w0, flags, r.variance, block_count, (uint32_t)(pos - 1));
                    if ((flags & M17_F_PARSED) && nrec < rec_cap && r.type >= 1 && r.type <= 3) {
                        float *fd = fsym + ((size_t)chan * rec_cap + nrec) * kSlotFloats;
                        store_frame_slot<LPC>(fd, r.type, gl, rg, [&](int q) { return my.H[(fs + q) & (kDuoRing - 1)]; });
                    }
                    nrec++;
                    if (unlock) {
                        flock = 0;
                        // reset_sync(): the next hunt windows must see zeros behind them
                        wave_fence();
                        if (gl < 8) my.H[(hp + pos - 8 + gl) & (kDuoRing - 1)] = 0.0f;
                        wave_fence();
                    }
                }
            } else {
                // hunt: candidate symbol j = pos + lane, window = ring [hp + j - 7, hp + j]
                SyncResult r;
                const int l = hunt_pass(pos, n, gl, [&](int i) { return my.H[(hp + i) & (kDuoRing - 1)]; }, r);
                if (l >= 0) {
                    const int js = pos + l;
                    // copy_sync(); m_fclk = 8; lock; m17_aos(): the window already is the head of the frame
                    fclk = M17_LIT_FCLK_AFTER_SYNC; ferr = 0; flock = 1;
                    emit_record_wave(crecs, rec_cap, nrec, gl, (uint32_t)r.type | ((uint32_t)r.votes << 8), M17_F_AOS, r.variance,
                                    block_count, (uint32_t)js);
                    nrec++;
                    pos = js + 1;
                } else {
                    pos = min(n, pos + LPC);
                }
            }
        }
        hp += n;
        block_count++;
        if (gl == 0) my.lock_after[b & 3] = flock;
        duo_post(frm_blk, b - b0 + 1, gl);
    }
#ifdef M17_STAMPS
    if (chan == 0 && gl == 0) { g_stamps[2] = acc_[2]; g_stamps[3] = acc_[3]; }
#endif
    // ---- store state in the reference's layout
    if (gl == 0) {
        cs.flock = flock; cs.fclk = fclk; cs.ferr = ferr; cs.block_count = block_count; cs.sym_total = sym_total;
        if (counts) counts[chan] = nrec;
    }
    if (flock) { for (int q = gl; q < kFrameSyms; q += LPC) cs.fsym[q] = my.H[(hp - fclk + q) & (kDuoRing - 1)]; }
    else if (gl < 8) cs.sync[gl] = my.H[(hp - 8 + gl) & (kDuoRing - 1)];
}

} // namespace m17dev
