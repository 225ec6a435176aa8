// m17_sync_ring.hip -- k_sync_frame_ring<LPC>: the lane-group timing/framer kernel
// (m17_sync_grp.hip) with the symbol stream kept in ONE ring per channel.
//
// Reference: m17_rx_sync_samples (m17_rx_sync.cpp:77-99), m17_rx_sym (m17_rx_frame.cpp:126-177).
//
// The reference moves every symbol twice: into the hunt window m_sync (update_sync, :106-110)
// or the frame buffer m_f_sym (:141), and copy_sync (:111-114) moves the window into the frame
// on lock.  k_sync_frame_grp mirrored that with h[] -> f[] copies, which made the framer 27 %
// of the stage's instructions (s_memtime stamps, profiles/r01_c_stamps_sync_grp.txt).  Here
// the timing loop writes symbol number k of the channel to ring[k & 511] and that is the only
// copy: the frame in progress is the ring span [hp - fclk, hp), a hunt window ending at
// symbol k is [k - 7, k], a completed frame is read in place for the sync check and the
// frame-symbol output, and the optional symbol stream goes to HBM straight from the round
// that made it.  512 >= 192 (frame) + 193 (block) + 8 (window).
// Per-channel state in HBM keeps the reference's layout (m_f_sym, m_sync): converted at load
// and store.
#pragma clang fp contract(off)

namespace m17dev {

constexpr int kRing = 512;

struct RingChan {                          // LDS of one channel
    float x[kTaps - 1 + kDiscOut + 2];     // delay-line history (30) + this block's 384 inputs
    float H[kRing];                        // symbol ring
};

template <int LPC>
__global__ __launch_bounds__(256)
void k_sync_frame_ring(const float *__restrict__ disc,     // [C][nblk][384]
                       const float *__restrict__ offs,     // [C][nblk] or null (already DC-free)
                       ChanState *__restrict__ st, int C, int nblk, int mode, int ext_lock,
                       m17gpu_rec_dev *__restrict__ recs, int rec_cap, int32_t *__restrict__ counts,
                       float *__restrict__ syms, int32_t *__restrict__ nsyms,
                       float *__restrict__ fsym, int b0, int bcount)
{
    using Cfg = GrpCfg<LPC>;
    __shared__ __attribute__((aligned(16))) float taps[kPhases * 64];      // (matched, derivative) pairs per branch
    __shared__ __attribute__((aligned(16))) RingChan chs[Cfg::CPW];
    const int wave = (int)(threadIdx.x >> 6), lane = lane_id();
    const int grp = lane / LPC, gl = lane % LPC, gbase = lane - gl, gshift = grp * LPC;
    for (int q = (int)threadIdx.x; q < kPhases * 32; q += 256) {
        taps[2 * q] = (&c_tab.mf[0][0])[q];
        taps[2 * q + 1] = (&c_tab.md[0][0])[q];
    }
    __syncthreads();                                    // the only workgroup barrier
    const int chan = ((int)blockIdx.x * 4 + wave) * Cfg::G + grp;
    if (chan >= C) return;
    RingChan &my = chs[wave * Cfg::G + grp];
    ChanState &cs = st[chan];
    m17gpu_rec_dev *crecs = recs ? recs + (size_t)chan * rec_cap : nullptr;
    if (!recs) rec_cap = 0;
    const unsigned long long incl = (gl == 63) ? ~0ull : ((2ull << gl) - 1ull);

    // group-uniform control state, one copy per lane
    int clk = cs.clk, thr = cs.thr, index = cs.index;
    float sum = cs.sum, dif = cs.dif;
    int flock = cs.flock, fclk = cs.fclk, ferr = cs.ferr;
    uint32_t block_count = cs.block_count;
    int nrec = (b0 == 0) ? 0 : counts[chan];
    int sym_total = (b0 == 0) ? 0 : cs.sym_total;
    int hp = 256;                                        // ring position of the next symbol
    for (int q = gl; q < kTaps - 1; q += LPC) my.x[q] = cs.buff[q + 1];
    // m_f_sym[0 .. fclk) is the frame in progress: ring [hp - fclk, hp); m_sync is the last 8 symbols
    if (flock) { for (int q = gl; q < kFrameSyms; q += LPC) my.H[(hp - fclk + q) & (kRing - 1)] = cs.fsym[q]; }
    else if (gl < 8) my.H[(hp - 8 + gl) & (kRing - 1)] = cs.sync[gl];
    float *sym_out = syms ? syms + (size_t)chan * M17_SYM_STRIDE(nblk) + sym_total : nullptr;
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
#endif
    for (int b = b0; b < bend; ++b) {
        STAMP(7);
        // ---- timing recovery in rounds of LPC instants; x[i .. i+30] is the delay line at input i
        const int lockv = (ext_lock >= 0) ? ext_lock : flock;
        const int thresh = lockv ? 80 : 10;
        int p = 0, m_idx = 0, wmax = 0;
        {
            float4 tp[16];                                            // 32 tap pairs of the current branch
            int tap_index = -1;
            while (p < kDiscOut) {
                if (clk == 1) {
                    // vote tick on the carried sum/dif (sync_update :38-42, m17_sync_adjust :45-72)
                    clk = 0;
                    const float d0 = (sum < 0.0f) ? -dif : dif;
                    if (d0 > 0.0f) thr++;
                    if (d0 < 0.0f) thr--;
                    if (thr > thresh) {
                        index = (index + 1 == kPhases) ? 0 : index + 1; thr = 0;
                        if (index == 0) {
                            clk = 1;
                            if (m_idx >= 0 && gl == 0) { my.H[(hp + m_idx) & (kRing - 1)] = 0.0f; if (sym_out) sym_out[m_idx] = 0.0f; }
                            m_idx++;
                        }
                    }
                    if (thr < -thresh) {
                        thr = 0; index = (index == 0) ? kPhases - 1 : index - 1;
                        if (index == kPhases - 1) { clk = 1; m_idx--; }
                    }
                    p++;
                    continue;
                }
                if (tap_index != index) {
                    const float4 *t4 = reinterpret_cast<const float4 *>(&taps[64 * index]);
#pragma unroll
                    for (int q = 0; q < 16; ++q) tp[q] = t4[q];
                    tap_index = index;
                }
                // one round: lane gl = filter tick at input p + 2 gl and the vote tick after it
                const int rem = (kDiscOut - p + 1) >> 1;              // filter instants left in the block
                const int nv = rem < LPC ? rem : LPC;
                const v2f a = fir_pair(my.x + p + 2 * (gl < nv ? gl : 0), tp);
                const float s = a.x, d = a.y;
                const bool vote_ok = (gl < nv) && (p + 2 * gl + 1 < kDiscOut);
                const float dd = (s < 0.0f) ? -d : d;
                const unsigned long long um = (__builtin_amdgcn_ballot_w64(vote_ok && dd > 0.0f) >> gshift) & Cfg::MASK;
                const unsigned long long dm = (__builtin_amdgcn_ballot_w64(vote_ok && dd < 0.0f) >> gshift) & Cfg::MASK;
                const int tk = thr + (int)__popcll(um & incl) - (int)__popcll(dm & incl);
                const unsigned long long cr =
                    (__builtin_amdgcn_ballot_w64(vote_ok && (tk > thresh || tk < -thresh)) >> gshift) & Cfg::MASK;
                const int kl = cr ? (int)__ffsll((long long)cr) - 1 : 0;
                const int naccept = cr ? kl + 1 : nv;
                if (gl < naccept && (m_idx + gl) >= 0) {
                    my.H[(hp + m_idx + gl) & (kRing - 1)] = s;
                    if (sym_out) sym_out[m_idx + gl] = s;             // a slipped-back symbol is overwritten in order
                }
                m_idx += naccept;
                wmax = max(wmax, m_idx);
                sum = __shfl(s, gbase + naccept - 1, 64);
                dif = __shfl(d, gbase + naccept - 1, 64);
                if (cr) {
                    const int ts = __shfl(tk, gbase + kl, 64);
                    thr = 0; clk = 0;
                    if (ts > thresh) {
                        index = (index + 1 == kPhases) ? 0 : index + 1;
                        if (index == 0) {
                            clk = 1;
                            if (m_idx >= 0 && gl == 0) { my.H[(hp + m_idx) & (kRing - 1)] = 0.0f; if (sym_out) sym_out[m_idx] = 0.0f; }
                            m_idx++;
                        }
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
        }
        const int n = m_idx > 0 ? m_idx : 0;
        wave_fence();
        STAMP(0);

        // next block's input: loads issued now, committed after the framer
        constexpr int PF = kDiscOut / LPC;
        float pf[PF];
        float noff = 0.0f;
        if (b + 1 < bend) {
            const float *nx = dsrc + (size_t)(b + 1) * kDiscOut;
            noff = osrc ? osrc[b + 1] : 0.0f;
#pragma unroll
            for (int r = 0; r < PF; ++r) pf[r] = nx[gl + LPC * r];
        }
        if (nsyms && gl == 0) nsyms[(size_t)chan * nblk + b] = n;
        // a bit slip at the very end leaves one written symbol past the count: the stream buffer stays as
        // the reference's memcpy of n symbols would leave a zeroed buffer
        if (sym_out && gl < wmax - n) sym_out[n + gl] = 0.0f;
        if (sym_out) sym_out += n;
        sym_total += n;

        STAMP(1);
        // ---- framer (m17_rx_frame.cpp:126-177) over ring symbols hp .. hp+n-1
        int pos = (ext_lock >= 0) ? n : 0;
        while (pos < n) {
            if (flock) {
                const int cnt = min(kFrameSyms - fclk, n - pos);
                fclk += cnt; pos += cnt;
                if (fclk == kFrameSyms) {
                    fclk = 0;
                    const int fs = hp + pos - kFrameSyms;               // the frame sits in the ring, in place
                    float v[8];
#pragma unroll
                    for (int i = 0; i < 8; ++i) v[i] = my.H[(fs + i) & (kRing - 1)];
                    STAMP(2);
                    const SyncResult r = sync_check_grp<LPC>(v, gl, gbase, gshift);
                    STAMP(3);
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
                        for (int q = gl; q < kFrameSyms; q += LPC) fd[q] = my.H[(fs + q) & (kRing - 1)];
                    }
                    nrec++;
                    STAMP(4);
                    if (unlock) {
                        flock = 0;
                        // reset_sync(): the next hunt windows must see zeros behind them
                        wave_fence();
                        if (gl < 8) my.H[(hp + pos - 8 + gl) & (kRing - 1)] = 0.0f;
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
                for (int i = 0; i < 8; ++i) v[i] = my.H[(hp + jj - 7 + i) & (kRing - 1)];
                SyncResult r; r.type = 0; r.votes = 8; r.variance = 1.0f;
                if (cand && hunt_compatible(v)) r = sync_check(v);           // most windows are rejected by sign
                const unsigned long long hm = (__builtin_amdgcn_ballot_w64(cand && sync_accept(r, false)) >> gshift) & Cfg::MASK;
                if (hm) {
                    const int l = (int)__ffsll((long long)hm) - 1;
                    const int js = pos + l;
                    // copy_sync(); m_fclk = 8; lock; m17_aos(): the window already is the head of the frame
                    fclk = 8; ferr = 0; flock = 1;
                    const int ty = __shfl(r.type, gbase + l, 64), vo = __shfl(r.votes, gbase + l, 64);
                    const float va = __shfl(r.variance, gbase + l, 64);
                    emit_record_grp(crecs, rec_cap, nrec, gl, (uint32_t)ty | ((uint32_t)vo << 8), M17_F_AOS, va,
                                    block_count, (uint32_t)js);
                    nrec++;
                    pos = js + 1;
                } else {
                    pos = min(n, pos + LPC);
                }
            }
        }
        hp += n;
        STAMP(5);
        // delay line: last 30 inputs; then the prefetched block moves in
        {
            float keep_x[(kTaps - 1 + LPC - 1) / LPC];
#pragma unroll
            for (int r = 0; r < (kTaps - 1 + LPC - 1) / LPC; ++r)
                keep_x[r] = (gl + LPC * r < kTaps - 1) ? my.x[kDiscOut + gl + LPC * r] : 0.0f;
            wave_fence();
#pragma unroll
            for (int r = 0; r < (kTaps - 1 + LPC - 1) / LPC; ++r)
                if (gl + LPC * r < kTaps - 1) my.x[gl + LPC * r] = keep_x[r];
            if (b + 1 < bend) {
#pragma unroll
                for (int r = 0; r < PF; ++r)
                    my.x[kTaps - 1 + gl + LPC * r] = osrc ? (pf[r] - noff) : pf[r];     // out[i] - offset
            }
        }
        block_count++;
        wave_fence();
    }

#ifdef M17_STAMPS
    if (chan == 0 && gl == 0) for (int i = 0; i < 12; ++i) g_stamps[i] = acc_[i];
#endif
    // ---- store state in the reference's layout
    if (gl == 0) {
        cs.clk = clk; cs.thr = thr; cs.index = index; cs.sum = sum; cs.dif = dif;
        cs.flock = flock; cs.fclk = fclk; cs.ferr = ferr; cs.block_count = block_count;
        cs.buff[0] = 0.0f; cs.sym_total = sym_total;
        if (counts) counts[chan] = nrec;
    }
    for (int q = gl; q < kTaps - 1; q += LPC) cs.buff[q + 1] = my.x[q];
    if (flock) { for (int q = gl; q < kFrameSyms; q += LPC) cs.fsym[q] = my.H[(hp - fclk + q) & (kRing - 1)]; }
    else if (gl < 8) cs.sync[gl] = my.H[(hp - 8 + gl) & (kRing - 1)];
}

} // namespace m17dev
