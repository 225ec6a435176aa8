// m17_sync_wave.hip -- k_sync_frame_wave: timing recovery + sync correlator + framer
// with ONE WAVE per channel and no workgroup barrier in the block loop.
//
// Reference: m17_rx_sync_samples (m17_rx_sync.cpp:77-99) and m17_rx_sym
// (m17_rx_frame.cpp:126-177).  Same speculation as k_sync_frame_wg -- all symbol
// instants of a 1920-sample block are evaluated under the current polyphase branch
// and the vote counter is a popcount prefix over ballots -- but each lane takes
// three instants (k, k+64, k+128: six independent add chains in flight), the
// ballots never leave the scalar registers, and the wave owns its channel's LDS
// arrays outright, so the only synchronisation is the in-order LDS pipeline of one
// wave.  Measured at 1,024 channels (1 wave per SIMD) this beats the 128/256-thread
// workgroup variants, whose barriers and duplicated control dominated.
//
// LDS per workgroup of 4 waves: one shared copy of both tap tables (10 KB) and,
// per wave, the block's input with its 30-sample delay line twice (second copy
// shifted by one float: every (x[a], x[a+1]) pair is an aligned ds_read_b64), the
// symbol buffer and m_f_sym.
#pragma clang fp contract(off)

namespace m17dev {

constexpr int SW_WAVES = 4;

struct SwWave {
    float xa[kTaps - 1 + kDiscOut + 2];
    float xb[kTaps - 1 + kDiscOut + 2];
    float h[8 + 208];                      // m_sync (8) followed by the block's symbols
    float f[kFrameSyms];                   // m_f_sym
};
struct SwShared {
    float mf[kPhases][32];
    float md[kPhases][32];
    SwWave wv[SW_WAVES];
};

__device__ __forceinline__ void wave_fence()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront", "local");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront", "local");
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
    constexpr unsigned tn[4] = {0xB0, 0x4F, 0xF2, 0x0D};     // bit i set: template symbol i is -1
    bool ok = false;
#pragma unroll
    for (int k = 0; k < 4; ++k) ok = ok || (((pos & tn[k]) == 0u) && ((neg & (~tn[k] & 0xFFu)) == 0u));
    return ok;
}

__global__ __launch_bounds__(64 * SW_WAVES)
void k_sync_frame_wave(const float *__restrict__ disc,     // [C][nblk][384]
                       const float *__restrict__ offs,     // [C][nblk] or null (already DC-free)
                       ChanState *__restrict__ st, int C, int nblk, int mode, int ext_lock,
                       m17gpu_rec_dev *__restrict__ recs, int rec_cap, int32_t *__restrict__ counts,
                       float *__restrict__ syms, int32_t *__restrict__ nsyms,
                       float *__restrict__ fsym, int32_t *__restrict__ work, int32_t *__restrict__ nwork,
                       int b0, int bcount)          // this launch: blocks b0 .. b0+bcount-1 of a call of nblk blocks
{
    __shared__ __attribute__((aligned(16))) SwShared sh;
    const int wave = (int)(threadIdx.x >> 6), lane = lane_id();
    for (int q = (int)threadIdx.x; q < kPhases * 32; q += 64 * SW_WAVES) {
        // taps interleaved as (matched, derivative) pairs: row idx = 64 floats starting at &sh.mf[0][0] + 64 idx
        (&sh.mf[0][0])[2 * q] = (&c_tab.mf[0][0])[q];
        (&sh.mf[0][0])[2 * q + 1] = (&c_tab.md[0][0])[q];
    }
    __syncthreads();                                    // the only workgroup barrier
    const int chan = (int)blockIdx.x * SW_WAVES + wave;
    if (chan >= C) return;
    SwWave &my = sh.wv[wave];
    ChanState &cs = st[chan];
    m17gpu_rec_dev *crecs = recs ? recs + (size_t)chan * rec_cap : nullptr;

    int clk = uni(cs.clk), thr = uni(cs.thr), index = uni(cs.index);
    float sum = unif(cs.sum), dif = unif(cs.dif);
    int flock = uni(cs.flock), fclk = uni(cs.fclk), ferr = uni(cs.ferr);
    uint32_t block_count = (uint32_t)uni((int)cs.block_count);
    if (lane < kTaps - 1) {
        const float v = cs.buff[lane + 1];
        my.xa[lane] = v;
        if (lane >= 1) my.xb[lane - 1] = v;
    }
    if (lane < 8) my.h[lane] = cs.sync[lane];
#pragma unroll
    for (int r = 0; r < 3; ++r) my.f[lane + 64 * r] = cs.fsym[lane + 64 * r];
    int nrec = (b0 == 0) ? 0 : uni(counts[chan]);
    int sym_total = (b0 == 0) ? 0 : uni(cs.sym_total);
    const size_t sym_base = (size_t)chan * M17_SYM_STRIDE(nblk);
    const float *dsrc = disc + (size_t)chan * nblk * kDiscOut;
    const float *osrc = offs ? offs + (size_t)chan * nblk : nullptr;

    // block 0 input (DC removed, m17_dsp.cpp:217-219); later blocks are prefetched
    float pf[6];
    {
        const float off = osrc ? osrc[b0] : 0.0f;
#pragma unroll
        for (int r = 0; r < 6; ++r) {
            float v = dsrc[(size_t)b0 * kDiscOut + lane + 64 * r];
            if (osrc) v = v - off;
            my.xa[kTaps - 1 + lane + 64 * r] = v;
            my.xb[kTaps - 2 + lane + 64 * r] = v;
        }
    }
    wave_fence();

#ifdef M17_STAMPS
    unsigned long long acc_[12] = {0,0,0,0,0,0,0,0,0,0,0,0}, last_ = __builtin_amdgcn_s_memtime();
    __builtin_amdgcn_s_waitcnt(0xC07F);
    const int t = lane;
#endif
    const int bend = b0 + bcount;
    for (int b = b0; b < bend; ++b) {
        STAMP(0);
        float noff = 0.0f;
        if (b + 1 < bend) {
            const float *nx = dsrc + (size_t)(b + 1) * kDiscOut;
            noff = osrc ? osrc[b + 1] : 0.0f;
#pragma unroll
            for (int r = 0; r < 6; ++r) pf[r] = nx[lane + 64 * r];
        }

        // ---- timing recovery: x[i .. i+30] is the delay line at input i; symbols go to h[8+..]
        const int lockv = (ext_lock >= 0) ? ext_lock : flock;
        const int thresh = lockv ? 80 : 10;
        const int width = lockv ? 192 : 64;              // hunting: a crossing every >= 11 instants
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
                    if (index == 0) { clk = 1; if (m_idx >= 0 && lane == 0) my.h[8 + m_idx] = 0.0f; m_idx++; }
                }
                if (thr < -thresh) {
                    thr = 0; index = (index + kPhases - 1) % kPhases;
                    if (index == kPhases - 1) { clk = 1; m_idx--; }
                }
                p++;
                continue;
            }
            // one pass: instant k (filter tick at input p+2k, vote tick after it), lane takes k = lane + 64 r
            const int nf = min(width, (kDiscOut - p + 1) >> 1);
            const float4 *mf4 = reinterpret_cast<const float4 *>(&sh.mf[0][0] + 64 * index);
            const float4 *md4 = reinterpret_cast<const float4 *>(sh.md[index]);
            float sv[3] = {0.0f, 0.0f, 0.0f}, dv[3] = {0.0f, 0.0f, 0.0f};
            unsigned long long U[3] = {0, 0, 0}, D[3] = {0, 0, 0};
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                if (64 * r < nf) {                                   // uniform
                    const int k = lane + 64 * r;
                    const bool have = k < nf;
                    const int a = p + 2 * (have ? k : 0);
                    const float *xs = (a & 1) ? (my.xb + (a - 1)) : (my.xa + a);
                    fir_instant(xs, mf4, md4, sv[r], dv[r]);
                    const bool vote_ok = have && (p + 2 * k + 1 < kDiscOut);
                    const float dd = (sv[r] < 0.0f) ? -dv[r] : dv[r];
                    U[r] = __ballot(vote_ok && dd > 0.0f);
                    D[r] = __ballot(vote_ok && dd < 0.0f);
                }
            }
            STAMP(1);
            const unsigned long long incl = (lane == 63) ? ~0ull : ((2ull << lane) - 1ull);
            int running = thr, kstar = -1, ts = 0;
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                if (64 * r < nf) {
                    const int tk = running + __popcll(U[r] & incl) - __popcll(D[r] & incl);
                    const int k = lane + 64 * r;
                    const bool ok = (k < nf) && (p + 2 * k + 1 < kDiscOut);
                    const unsigned long long cr = __ballot(ok && (tk > thresh || tk < -thresh));
                    if (kstar < 0 && cr) {
                        const int kl = __ffsll((long long)cr) - 1;
                        kstar = 64 * r + kl;
                        ts = bcast_lane_i(tk, kl);
                    }
                    running += (int)__popcll(U[r]) - (int)__popcll(D[r]);
                }
            }
            const int naccept = (kstar >= 0) ? kstar + 1 : nf;
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                const int k = lane + 64 * r;
                if (k < naccept && (m_idx + k) >= 0) my.h[8 + m_idx + k] = sv[r];
            }
            m_idx += naccept;
            {
                const int last = naccept - 1, lr = last >> 6, ll = last & 63;
                const float s0 = bcast_lane(sv[0], ll), s1 = bcast_lane(sv[1], ll), s2 = bcast_lane(sv[2], ll);
                const float d0 = bcast_lane(dv[0], ll), d1 = bcast_lane(dv[1], ll), d2 = bcast_lane(dv[2], ll);
                sum = (lr == 0) ? s0 : (lr == 1 ? s1 : s2);
                dif = (lr == 0) ? d0 : (lr == 1 ? d1 : d2);
            }
            if (kstar >= 0) {
                thr = 0; clk = 0;
                if (ts > thresh) {
                    index = (index + 1) % kPhases;
                    if (index == 0) { clk = 1; if (m_idx >= 0 && lane == 0) my.h[8 + m_idx] = 0.0f; m_idx++; }
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
            STAMP(2);
        }
        const int n = m_idx > 0 ? m_idx : 0;
        wave_fence();
        STAMP(3);

        // symbols out (optional)
        if (syms) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int q = lane + 64 * r;
                if (q < n) syms[sym_base + sym_total + q] = my.h[8 + q];
            }
        }
        if (nsyms && lane == 0) nsyms[(size_t)chan * nblk + b] = n;
        sym_total += n;

        STAMP(4);
        // ---- framer (m17_rx_frame.cpp:126-177)
        int pos = (ext_lock >= 0) ? n : 0;
        while (pos < n) {
            pos = uni(pos); fclk = uni(fclk); flock = uni(flock);
            if (flock) {
                const int cnt = min(kFrameSyms - fclk, n - pos);
                for (int q = lane; q < cnt; q += 64) my.f[fclk + q] = my.h[8 + pos + q];
                fclk += cnt; pos += cnt;
                wave_fence();
                if (fclk == kFrameSyms) {
                    fclk = 0;
                    float v[8];
#pragma unroll
                    for (int i = 0; i < 8; ++i) v[i] = my.f[i];
                    const SyncResult r = sync_check_wave(v);
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
                    emit_record(crecs, rec_cap, nrec, w0, flags, r.variance, block_count, (uint32_t)(pos - 1));
                    if ((flags & M17_F_PARSED) && nrec < rec_cap && r.type >= 1 && r.type <= 3) {
                        float *fd = fsym + ((size_t)chan * rec_cap + nrec) * kFrameSyms;
#pragma unroll
                        for (int q = 0; q < 3; ++q) fd[lane + 64 * q] = my.f[lane + 64 * q];
                        if (work && lane == 0) work[atomicAdd(nwork, 1)] = chan * rec_cap + nrec;
                    }
                    nrec++;
                    if (unlock) {
                        flock = 0;
                        // reset_sync(): the next hunt windows must see zeros behind them
                        if (lane < 8) { my.h[pos + lane] = 0.0f; cs.sync[lane] = 0.0f; }
                        wave_fence();
                    }
                }
            } else {
                // hunt: candidate j = pos+lane, window = m_sync after shifting symbol j in
                const int jc = pos + lane;
                const bool cand = jc < n;
                const int jj = cand ? jc : pos;
                float v[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) v[i] = my.h[jj + 1 + i];
                unsigned long long hm = 0;
                SyncResult r; r.type = 0; r.votes = 8; r.variance = 1.0f;
                if (__ballot(cand && hunt_compatible(v)) != 0ull) {      // uniform: most windows are rejected by sign
                    r = sync_check(v);
                    hm = __ballot(cand && sync_accept(r, false));
                }
                if (hm) {
                    const int l = __ffsll((long long)hm) - 1;
                    const int js = pos + l;
                    // copy_sync(); m_fclk = 8; lock; m17_aos()
                    float wv = 0.0f;
                    if (lane < 8) wv = my.h[js + 1 + lane];
                    wave_fence();
                    if (lane < 8) { my.f[lane] = wv; cs.sync[lane] = wv; }
                    fclk = 8; ferr = 0; flock = 1;
                    const int ty = bcast_lane_i(r.type, l), vo = bcast_lane_i(r.votes, l);
                    const float va = bcast_lane(r.variance, l);
                    emit_record(crecs, rec_cap, nrec, (uint32_t)ty | ((uint32_t)vo << 8), M17_F_AOS, va,
                                block_count, (uint32_t)js);
                    nrec++;
                    pos = js + 1;
                    wave_fence();
                } else {
                    pos = min(n, pos + 64);
                }
            }
        }
        STAMP(5);
        // m_sync for the next block while hunting: last 8 entries of h; delay line: last 30
        // inputs; then the prefetched block moves in
        {
            float keep_h = 0.0f, keep_x = 0.0f;
            if (lane < 8) keep_h = my.h[n + lane];
            if (lane < kTaps - 1) keep_x = my.xa[kDiscOut + lane];
            wave_fence();
            if (!flock && lane < 8) { my.h[lane] = keep_h; cs.sync[lane] = keep_h; }
            if (lane < kTaps - 1) {
                my.xa[lane] = keep_x;
                if (lane >= 1) my.xb[lane - 1] = keep_x;
            }
            if (b + 1 < bend) {
#pragma unroll
                for (int r = 0; r < 6; ++r) {
                    const float v = osrc ? (pf[r] - noff) : pf[r];        // out[i] - offset (m17_dsp.cpp:217-219)
                    my.xa[kTaps - 1 + lane + 64 * r] = v;
                    my.xb[kTaps - 2 + lane + 64 * r] = v;
                }
            }
        }
        block_count++;
        wave_fence();
        STAMP(6);
    }
#ifdef M17_STAMPS
    if (chan == 0 && lane == 0) for (int i = 0; i < 12; ++i) g_stamps[i] = acc_[i];
#endif

    // ---- store state
    if (lane == 0) {
        cs.clk = clk; cs.thr = thr; cs.index = index; cs.sum = sum; cs.dif = dif;
        cs.flock = flock; cs.fclk = fclk; cs.ferr = ferr; cs.block_count = block_count;
        cs.buff[0] = 0.0f; cs.sym_total = sym_total;
        if (counts) counts[chan] = nrec;
    }
    if (lane < kTaps - 1) cs.buff[lane + 1] = my.xa[lane];
#pragma unroll
    for (int r = 0; r < 3; ++r) cs.fsym[lane + 64 * r] = my.f[lane + 64 * r];
}

} // namespace m17dev
