// m17_sync_wave.hip -- k_sync_frame_wave: timing recovery + sync correlator + framer, ONE WAVE PER
// CHANNEL with the whole control flow on the scalar unit and the filter taps in SGPRs.
//
// Reference: m17_rx_sync_samples (m17_rx_sync.cpp:77-99), m17_rx_sym (m17_rx_frame.cpp:126-177).
//
// Why (round-2 measurements): the lane-group kernel it replaces (four channels per wave,
// 16 lanes each, taps in 62 VGPRs; retired in round 3) ran at two waves per SIMD -- 241 VGPRs, 63 KB of LDS per workgroup --
// and at one instruction per ~9 cycles per wave: bound by latency, not by its 890 instructions per
// channel-block.  Its control variables are equal within a lane group but differ between the groups of
// a wave, so every branch is EXEC-mask divergence and every wave executes the union of its four
// channels' paths.  Here a wave owns one channel:
//   * every control variable (m_clk, m_thr, m_index, the symbol count, lock, frame clock) is
//     wave-uniform: SGPRs, scalar ALU, scalar branches; vote masks come straight out of v_cmp;
//   * the 31 (matched, derivative) tap pairs of the current polyphase branch are 62 SGPRs, loaded
//     with s_load from the constant table at the head of every round, and are the scalar operand of the
//     packed multiplies: no tap VGPRs, no tap LDS;
//   * what is left per lane is the 31-sample window and the accumulators: ~70 VGPRs and 3.7 KB of LDS
//     per wave, i.e. 6-7 waves per SIMD, which is what hides the LDS and issue latency.
// A round takes 64 consecutive symbol instants under the current branch; a threshold crossing ends
// the round early and the next round starts behind it (nothing past a crossing is ever used).
// Symbols go into a per-channel LDS ring in which frames and hunt windows are read in place.
#pragma clang fp contract(off)

namespace m17dev {

constexpr int kWvRing = 512;                   // 191 (frame in progress) + 193 (block being written) < 512

struct WvChan {                                // LDS of one channel: 4 KB, the ring 2 KB-aligned (ring addresses by AND/OR)
    float x[kTaps - 1 + kDiscOut + 2];         // delay-line history (30) + this block's 384 inputs
    float pad[1024 - kWvRing - (kTaps - 1 + kDiscOut + 2)];
    float H[kWvRing];                          // symbol ring; it follows x[]: the last round of a block reads up to 124 floats
                                               // past x[] (lanes whose instants lie beyond the block; nothing of theirs is used)
};
static_assert(sizeof(WvChan) == 4096, "WvChan layout");

// The filter statement below loads the current branch's 31 (matched, derivative) tap pairs into s[36:97] itself and
// declares them clobbered: tap pair j is s[36+2j : 37+2j].  (Kept there for the whole kernel -- loaded only when the
// branch changed, the allocator held below s40 with amdgpu_num_sgpr(46) -- the control state around the rounds lived
// in VGPR lanes: 100+ spills; same-box A/B 0.294 -> 0.290 ms for the reload in every round.)
#define M17_TAP_CLOBBERS "s36","s37","s38","s39","s40","s41","s42","s43","s44","s45","s46","s47","s48","s49","s50", \
    "s51","s52","s53","s54","s55","s56","s57","s58","s59","s60","s61","s62","s63","s64","s65","s66","s67","s68","s69", \
    "s70","s71","s72","s73","s74","s75","s76","s77","s78","s79","s80","s81","s82","s83","s84","s85","s86","s87","s88", \
    "s89","s90","s91","s92","s93","s94","s95","s96","s97"

// rx_sync_filter (m17_rx_sync.cpp:25-31) for both filters as one packed (s, d) chain, ascending order, bare first
// product, separate multiply and add, the tap pair as the scalar source operand of v_pk_mul_f32.  The whole round --
// six scalar tap loads, sixteen aligned 8-byte window reads, one wait, 31 multiplies and 30 adds -- is ONE asm statement (text in
// m17_fir_sgpr.inc, written by scripts/gen_fir_asm.py), so no compiler-visible register ever holds a window value
// that is still in flight.  ds_read_b64 with lane stride 8 bytes covers 64 consecutive dwords per half-wave:
// conflict-free, 2 LDS cycles each; the ds_read2_b32 form of the first version (lane stride 2 dwords, 32 banks:
// 2-way conflicts, 8 cycles each) made the LDS pipe the limit of the kernel (~600 of its cycles per channel-block).
//   row      : the branch's row of DevTables.tap_pairs
//   lds_pair : LDS byte address of the aligned pair that holds the window's first sample (x[w & ~1])
//   odd      : the window starts on the pair's second element (wave-uniform)
#include "m17_fir_sgpr.inc"
#define M17_FIR_OPERANDS                                                                                                 \
    [acc] "=&v"(acc), [p] "=&v"(P), [q] "=&v"(Q), [x0] "=&v"(x0), [x1] "=&v"(x1), [x2] "=&v"(x2), [x3] "=&v"(x3),        \
    [x4] "=&v"(x4), [x5] "=&v"(x5), [x6] "=&v"(x6), [x7] "=&v"(x7), [x8] "=&v"(x8), [x9] "=&v"(x9), [x10] "=&v"(x10),   \
    [x11] "=&v"(x11), [x12] "=&v"(x12), [x13] "=&v"(x13), [x14] "=&v"(x14), [x15] "=&v"(x15)
#define M17_TAP_CLOBBERS_H "s36","s37","s38","s39","s40","s41","s42","s43","s44","s45","s46","s47","s48","s49","s50", \
    "s51","s52","s53","s54","s55","s56","s57","s58","s59","s60","s61","s62","s63","s64","s65","s66","s67"
#define M17_FIR_OPERANDS_H                                                                                               \
    [acc] "=&v"(acc), [p] "=&v"(P), [q] "=&v"(Q), [x0] "=&v"(x0), [x1] "=&v"(x1), [x2] "=&v"(x2), [x3] "=&v"(x3),        \
    [x4] "=&v"(x4), [x5] "=&v"(x5), [x6] "=&v"(x6), [x7] "=&v"(x7)
// HALF: the window through eight registers, read twice (m17_fir_sgpr.inc, *_H): 16 VGPRs less per lane, one more wait on LDS per round
template <int HALF = 0>
__device__ __forceinline__ v2f fir_window_s(const float *row, unsigned lds_pair, bool odd)
{
    if (HALF) {
        v2f acc, P, Q, x0, x1, x2, x3, x4, x5, x6, x7;
        if (!odd) asm volatile(M17_FIR_SGPR_EVEN_H : M17_FIR_OPERANDS_H : [a] "v"(lds_pair), [row] "s"(row) : "memory", M17_TAP_CLOBBERS_H);
        else      asm volatile(M17_FIR_SGPR_ODD_H : M17_FIR_OPERANDS_H : [a] "v"(lds_pair), [row] "s"(row) : "memory", M17_TAP_CLOBBERS_H);
        return acc;
    }
    v2f acc, P, Q, x0, x1, x2, x3, x4, x5, x6, x7, x8, x9, x10, x11, x12, x13, x14, x15;
    if (!odd) asm volatile(M17_FIR_SGPR_EVEN : M17_FIR_OPERANDS : [a] "v"(lds_pair), [row] "s"(row) : "memory", M17_TAP_CLOBBERS);
    else      asm volatile(M17_FIR_SGPR_ODD : M17_FIR_OPERANDS : [a] "v"(lds_pair), [row] "s"(row) : "memory", M17_TAP_CLOBBERS);
    return acc;
}
typedef const __attribute__((address_space(3))) float *lds_cfp;

__device__ __forceinline__ float readlane_f(float v, int l)
{
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l));
}

typedef __attribute__((address_space(3))) float lds_f;
// ring element i (any integer) of the channel whose ring starts at LDS byte address hb (2 KB-aligned)
__device__ __forceinline__ unsigned ring_addr(int i, unsigned hb) { return (((unsigned)i << 2) & (4u * (kWvRing - 1))) | hb; }
__device__ __forceinline__ float ring_ld(int i, unsigned hb) { return *(const lds_f *)(uintptr_t)ring_addr(i, hb); }
__device__ __forceinline__ void ring_st(int i, unsigned hb, float v) { *(lds_f *)(uintptr_t)ring_addr(i, hb) = v; }

// A completed frame, in place in the ring from element fs, into its slot of the decoder's workspace (layout:
// m17_dev.h kSlotFloats; store_frame_slot, m17_sync_common.hip).  rgw = the lane's packed regroup bytes.
__device__ __forceinline__ void store_frame_slot_wave(float *__restrict__ fd, int type, int gl, const uint32_t (&rgw)[2],
                                                      int fs, unsigned hb)
{
    if (type == 2) {
        if (gl < 8) fd[gl] = ring_ld(fs + gl, hb);
        float4 t[2];
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            float e[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) e[c] = ring_ld(fs + 8 + (int)((rgw[r] >> (8 * c)) & 0xFFu), hb);
            t[r] = make_float4(e[0], e[1], e[2], e[3]);
        }
        *reinterpret_cast<float4 *>(fd + 8 + 4 * gl) = t[0];
        if (gl < kRegroup / 4 - 64) *reinterpret_cast<float4 *>(fd + 8 + 4 * (gl + 64)) = t[1];
    } else if (gl < kFrameSyms / 4) {
        const float4 t = make_float4(ring_ld(fs + 4 * gl, hb), ring_ld(fs + 4 * gl + 1, hb), ring_ld(fs + 4 * gl + 2, hb),
                                     ring_ld(fs + 4 * gl + 3, hb));
        *reinterpret_cast<float4 *>(fd + 4 * gl) = t;
    }
}

constexpr int WV_WAVES = 4;                    // channels (waves) per workgroup; the waves never synchronise

#ifdef M17_STAMPS
#define WSTAMP(i) do { __builtin_amdgcn_sched_barrier(0); const unsigned now_ = (unsigned)__builtin_amdgcn_s_memtime(); \
    __builtin_amdgcn_sched_barrier(0); if (gl == 0) wst[i] += now_ - t.last_; t.last_ = now_; } while (0)
#define WCNT(i) do { if (gl == 0) wst[i] += 1; } while (0)
#else
#define WSTAMP(i) do {} while (0)
#define WCNT(i) do {} while (0)
#endif

// Wave-uniform control state of one channel: the reference's file statics of the timing loop (m17_rx_sync.cpp:6-9,78)
// and of the framer (m17_rx_frame.cpp:16-18) plus this call's counters.  Scalar registers in both kernels that use it.
struct WvCtl {
    int clk, thr, index;                       // m_clk, m_thr, m_index
    float cs_, cd_;                            // carried (sum, dif): lane `clane` of these
    int clane;
    int flock, fclk, ferr;                     // m_flock, m_fclk, m_frame_errors
    uint32_t block_count;
    int nrec, sym_total;
    int hp;                                    // ring position of the block's first symbol
#ifdef M17_STAMPS
    unsigned last_;
#endif
};

// m17_rx_sync_samples (m17_rx_sync.cpp:77-99) over ONE block of 384 inputs: x[i .. i+30] is the delay line at input i,
// xb the LDS byte address of x[0] (8-byte aligned; up to 124 floats behind x[413] are read and never used), hb that of
// the channel's symbol ring.  Rounds of 64 instants; symbols go into the ring from t.hp on.  Returns the symbol count.
template <int HALF = 0>
__device__ __forceinline__ int wv_timing_block(WvCtl &t, const unsigned xb, const unsigned hb, const int gl, const int lockv, unsigned *wst)
{
    constexpr int LPC = 64;
    const int thresh = lockv ? M17_LIT_THRESH_LOCKED : M17_LIT_THRESH_UNLOCKED;
    int p = 0, m_idx = 0;
    // vote tick on the carried sum/dif (sync_update :38-42, m17_sync_adjust :45-72): the first input of a block whose
    // predecessor ended on a filter instant, and the input behind a wrap of the branch
    auto tick = [&]() {
        t.clk = 0;
        const float sum = readlane_f(t.cs_, t.clane), dif = readlane_f(t.cd_, t.clane);
        const float d0 = (sum < 0.0f) ? -dif : dif;
        if (d0 > 0.0f) t.thr++;
        if (d0 < 0.0f) t.thr--;
        if (t.thr > thresh) {
            t.index = (t.index + 1 == kPhases) ? 0 : t.index + 1; t.thr = 0;
            if (t.index == 0) { t.clk = 1; if (m_idx >= 0 && gl == 0) ring_st(t.hp + m_idx, hb, 0.0f); m_idx++; }
        }
        if (t.thr < -thresh) {
            t.thr = 0; t.index = (t.index == 0) ? kPhases - 1 : t.index - 1;
            if (t.index == kPhases - 1) { t.clk = 1; m_idx--; }
        }
        p++;
    };
    while (t.clk == 1 && p < kDiscOut) tick();
    while (p < kDiscOut) {
        WCNT(8);
        WSTAMP(0);
        // Lane g takes the instant at input p + 2g.  Near the end of the block the upper lanes run past it: their
        // windows read what follows the block in the channel's LDS, and nothing of theirs is used -- no vote (okm),
        // no symbol (naccept <= nv), no carried value.
        const v2f a = fir_window_s<HALF>(&c_tab.tap_pairs[t.index][0], ((unsigned)gl << 3) + (xb + ((unsigned)(p & ~1) << 2)), (p & 1) != 0);
        WSTAMP(1);
        const float s = a.x, d = a.y;
        const int rem = kDiscOut - p;                     // >= 1
        const int nv = min(LPC, (rem + 1) >> 1);          // filter instants of this round
        const int nvote = min(LPC, rem >> 1);             // ... whose vote tick p + 2g + 1 is inside the block
        const unsigned long long okm = (nvote >= 64) ? ~0ull : ((1ull << nvote) - 1ull);
        const float dd = (s < 0.0f) ? -d : d;             // sync_update, m17_rx_sync.cpp:38-42
        const unsigned long long um = __builtin_amdgcn_ballot_w64(dd > 0.0f) & okm;
        const unsigned long long dm = __builtin_amdgcn_ballot_w64(dd < 0.0f) & okm;
        const int nu = (int)__popcll(um), nd = (int)__popcll(dm);
        int naccept = nv, kl = -1, ts_ = 0;
        if (t.thr + nu > thresh || t.thr - nd < -thresh) {
            // a crossing is possible in this round: the counter after every tick, first crossing wins
            const int pu = (int)__builtin_amdgcn_mbcnt_hi((unsigned)(um >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)um, 0u));
            const int pd = (int)__builtin_amdgcn_mbcnt_hi((unsigned)(dm >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)dm, 0u));
            const int own = (int)((um >> gl) & 1ull) - (int)((dm >> gl) & 1ull);
            const int tk = t.thr + pu - pd + own;
            const unsigned long long cr = __builtin_amdgcn_ballot_w64(tk > thresh || tk < -thresh) & okm;
            if (cr) {
                kl = (int)__builtin_ctzll(cr);
                naccept = kl + 1;
                ts_ = __builtin_amdgcn_readlane(tk, kl);
            }
        }
        if (gl < naccept && (m_idx + gl) >= 0) ring_st(t.hp + m_idx + gl, hb, s);
        m_idx += naccept;
        t.cs_ = s; t.cd_ = d; t.clane = naccept - 1;       // the carried sum/dif stay in their lane until a tick needs them
        if (kl >= 0) {
            t.thr = 0; t.clk = 0;
            if (ts_ > thresh) {
                t.index = (t.index + 1 == kPhases) ? 0 : t.index + 1;
                if (t.index == 0) { t.clk = 1; if (m_idx >= 0 && gl == 0) ring_st(t.hp + m_idx, hb, 0.0f); m_idx++; }
            } else {
                t.index = (t.index == 0) ? kPhases - 1 : t.index - 1;
                if (t.index == kPhases - 1) { t.clk = 1; m_idx--; }
            }
            p = p + 2 * kl + 2;
            while (t.clk == 1 && p < kDiscOut) tick();     // a wrap: the next input is a vote tick again
        } else {
            t.thr += nu - nd;
            p += 2 * nv;                                   // behind the last instant's vote tick ...
            t.clk = p > kDiscOut ? 1 : 0;                  // ... which falls into the next block when that instant is input 383
            p = min(p, kDiscOut);
        }
        WSTAMP(2);
    }
    return m_idx > 0 ? m_idx : 0;
}

// The block's n symbols, in the ring from t.hp on: optional symbol stream out, then the framer (m17_rx_sym,
// m17_rx_frame.cpp:126-177) -- records, frame slots for the decoder -- and the ring position of the next block.
struct WvOut {
    m17gpu_rec_dev *crecs; int rec_cap;
    float *sym_out; int32_t *nsyms_row;        // this channel's symbol stream (advances) / per-block counts, or null
    float *fsym_chan;                          // this channel's frame slots [rec_cap][kSlotFloats]
    int mode, ext_lock;
};
__device__ __forceinline__ void wv_framer_block(WvCtl &t, WvOut &o, const int n, const int b, const unsigned hb, const int gl,
                                                const RegroupLane<64> &rg, unsigned *wst)
{
    constexpr int LPC = 64;
    if (o.sym_out) {
#pragma unroll
        for (int r = 0; r < (193 + LPC - 1) / LPC; ++r) {
            const int q = gl + LPC * r;
            if (q < n) __builtin_nontemporal_store(ring_ld(t.hp + q, hb), &o.sym_out[q]);
        }
        o.sym_out += n;
    }
    if (o.nsyms_row && gl == 0) o.nsyms_row[b] = n;
    t.sym_total += n;

    WSTAMP(3);
    int pos = (o.ext_lock >= 0) ? n : 0;
    while (pos < n) {
        WCNT(10);
        if (t.flock) {
            const int cnt = min(kFrameSyms - t.fclk, n - pos);
            t.fclk += cnt; pos += cnt;
            if (t.fclk == kFrameSyms) {
                t.fclk = 0;
                const int fs = t.hp + pos - kFrameSyms;               // the frame sits in the ring, in place
                const SyncResult r = sync_check_lanes8(ring_ld(fs + (gl & 7), hb), sync_sign_mask(gl));
                uint32_t flags = 0;
                bool parse = false, unlock = false;
                if (r.type == 5) { flags |= M17_F_EOT; unlock = true; }
                else if (sync_accept(r, true)) { flags |= M17_F_SYNC_OK; parse = true; t.ferr = 0; }
                else {
                    t.ferr++;
                    if (t.ferr > M17_LIT_N_FERROR) { flags |= M17_F_LOST; unlock = true; }
                    else parse = true;
                }
                if (parse && (o.mode & 1)) flags |= M17_F_PARSED;
                const uint32_t w0 = (uint32_t)r.type | ((uint32_t)r.votes << 8) | ((uint32_t)(t.ferr & 0xFF) << 24);
                emit_record_wave(o.crecs, o.rec_cap, t.nrec, gl, w0, flags, r.variance, t.block_count, (uint32_t)(pos - 1));
                if ((flags & M17_F_PARSED) && t.nrec < o.rec_cap && r.type >= 1 && r.type <= 3) {
                    float *fd = o.fsym_chan + (size_t)t.nrec * kSlotFloats;
                    store_frame_slot_wave(fd, (o.mode & 16) ? 1 : r.type, gl, rg.w, fs, hb);     // mode bit 4 (slot_impl 1): every frame as its 192 symbols
                }
                t.nrec++;
                if (unlock) {
                    t.flock = 0;
                    // reset_sync(): the next hunt windows must see zeros behind them
                    wave_fence();
                    if (gl < 8) ring_st(t.hp + pos - 8 + gl, hb, 0.0f);
                    wave_fence();
                }
            }
        } else {
            // hunt: candidate symbol j = pos + lane, window = ring [hp + j - 7, hp + j]
            SyncResult r;
            const int l = hunt_pass(pos, n, gl, [&](int i) { return ring_ld(t.hp + i, hb); }, r);
            if (l >= 0) {
                const int js = pos + l;
                // copy_sync(); m_fclk = 8; lock; m17_aos(): the window already is the head of the frame
                t.fclk = M17_LIT_FCLK_AFTER_SYNC; t.ferr = 0; t.flock = 1;
                emit_record_wave(o.crecs, o.rec_cap, t.nrec, gl, (uint32_t)r.type | ((uint32_t)r.votes << 8), M17_F_AOS, r.variance,
                                 t.block_count, (uint32_t)js);
                t.nrec++;
                pos = js + 1;
            } else {
                pos = min(n, pos + LPC);
            }
        }
    }
    t.hp = (t.hp + n) & (kWvRing - 1);
    if (o.ext_lock < 0) t.block_count++;
    WSTAMP(4);
}

// channel state <-> control registers / ring (ChanState keeps the reference's layout: m_f_sym[0 .. fclk) is the frame
// in progress = ring [hp - fclk, hp); m_sync the last 8 symbols)
__device__ __forceinline__ void wv_load_state(WvCtl &t, const ChanState &cs, const int32_t *counts, int chan, int b0, unsigned hb, int gl)
{
    t.clk = uni(cs.clk); t.thr = uni(cs.thr); t.index = uni(cs.index);
    t.cs_ = cs.sum; t.cd_ = cs.dif; t.clane = 0;
    t.flock = uni(cs.flock); t.fclk = uni(cs.fclk); t.ferr = uni(cs.ferr);
    t.block_count = (uint32_t)uni((int)cs.block_count);
    t.nrec = (b0 == 0) ? 0 : uni(counts[chan]);
    t.sym_total = (b0 == 0) ? 0 : uni(cs.sym_total);
    t.hp = 256;
    if (t.flock) { for (int q = gl; q < t.fclk; q += 64) ring_st(t.hp - t.fclk + q, hb, cs.fsym[q]); }
    else if (gl < 8) ring_st(t.hp - 8 + gl, hb, cs.sync[gl]);
}
__device__ __forceinline__ void wv_store_state(const WvCtl &t, ChanState &cs, int32_t *counts, int chan, int ext_lock, unsigned hb, int gl)
{
    const float sum_out = readlane_f(t.cs_, t.clane), dif_out = readlane_f(t.cd_, t.clane);
    if (gl == 0) {
        cs.clk = t.clk; cs.thr = t.thr; cs.index = t.index; cs.sum = sum_out; cs.dif = dif_out; cs.buff[0] = 0.0f;
        if (ext_lock < 0) {
            cs.flock = t.flock; cs.fclk = t.fclk; cs.ferr = t.ferr; cs.block_count = t.block_count; cs.sym_total = t.sym_total;
            if (counts) counts[chan] = t.nrec;
        }
    }
    if (ext_lock < 0) {
        if (t.flock) { for (int q = gl; q < kFrameSyms; q += 64) cs.fsym[q] = ring_ld(t.hp - t.fclk + q, hb); }
        else if (gl < 8) cs.sync[gl] = ring_ld(t.hp - 8 + gl, hb);
    }
}

// the whole call of ONE channel by the calling wave; `my` = the channel's 4 KB of LDS (4 KB-aligned), wave = the wave's
// index in its workgroup (instrumented build only)
// OFFS_AGENT: the block offsets are read past the CU's caches (agent scope) -- for a caller whose rows were written in
// this kernel by ANOTHER wave of the workgroup (k_rx_chan6's shared tiles): thirty-two offsets share a 128-byte line, and a
// line this wave read for an earlier group must not be served again once a sibling has written the next group's part of it.
template <int HALF = 0, int OFFS_AGENT = 0>
__device__ __forceinline__ void sync_wave_channel(const float *__restrict__ disc, const float *__restrict__ offs,
                       ChanState *__restrict__ st, int C, int nblk, int mode, int ext_lock,
                       m17gpu_rec_dev *__restrict__ recs, int rec_cap, int32_t *__restrict__ counts,
                       float *__restrict__ syms, int32_t *__restrict__ nsyms,
                       float *__restrict__ fsym, int b0, int bcount, const int chan, WvChan &my, const int wave,
                       const int lane = lane_id())
{
    constexpr int LPC = 64;
#ifdef M17_STAMPS
    const unsigned t_entry_ = (unsigned)__builtin_amdgcn_s_memtime();
    const unsigned long long rt_entry_ = __builtin_amdgcn_s_memrealtime();
#endif
    const int gl = lane;
    if (chan >= C) return;
    const unsigned hb = (unsigned)uni((int)(unsigned)(uintptr_t)(lds_cfp)my.H);       // LDS byte address of the ring
    ChanState &cs = st[chan];
    if (!recs) rec_cap = 0;

    WvCtl t;
    wv_load_state(t, cs, counts, chan, b0, hb, gl);
    RegroupLane<LPC> rg;
    rg.load(gl);
    WvOut o;
    o.crecs = recs ? recs + (size_t)chan * rec_cap : nullptr; o.rec_cap = rec_cap;
    o.sym_out = syms ? syms + (size_t)chan * M17_SYM_STRIDE(nblk) + t.sym_total : nullptr;
    o.nsyms_row = nsyms ? nsyms + (size_t)chan * nblk : nullptr;
    o.fsym_chan = fsym + (size_t)chan * rec_cap * kSlotFloats;
    o.mode = mode; o.ext_lock = ext_lock;

    if (gl < kTaps - 1) my.x[gl] = cs.buff[gl + 1];
    const float *dsrc = disc + (size_t)chan * nblk * kDiscOut;
    const float *osrc = offs ? offs + (size_t)chan * nblk : nullptr;
    auto ld_off = [&](int b) {
        if constexpr (OFFS_AGENT) return __hip_atomic_load(&osrc[b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else return osrc[b];
    };
    {
        const float off = osrc ? ld_off(b0) : 0.0f;
#pragma unroll
        for (int r = 0; r < kDiscOut / LPC; ++r) {
            float v = __builtin_nontemporal_load(&dsrc[(size_t)b0 * kDiscOut + gl + LPC * r]);
            if (osrc) v = v - off;                               // out[i] - offset (m17_dsp.cpp:217-219)
            my.x[kTaps - 1 + gl + LPC * r] = v;
        }
    }
    wave_fence();

    const unsigned xb = (unsigned)uni((int)(unsigned)(uintptr_t)(lds_cfp)my.x);       // LDS byte address of x[]
    const int bend = b0 + bcount;
    unsigned *wst = nullptr;
#ifdef M17_STAMPS
    // phase accumulators in LDS (the scalar registers are spoken for): lane 0 adds the ticks since the last stamp
    __shared__ unsigned wstamps[16][12];
    if (gl < 12) wstamps[wave][gl] = 0;
    wst = wstamps[wave];
    t.last_ = (unsigned)__builtin_amdgcn_s_memtime();
    if (gl == 0) wstamps[wave][6] = t.last_ - t_entry_;       // prologue: state and first block in
#endif
    for (int b = b0; b < bend; ++b) {
        WSTAMP(5);
        // m17_rx_lock(): the framer's state after the previous block
        const int n = wv_timing_block<HALF>(t, xb, hb, gl, (ext_lock >= 0) ? ext_lock : t.flock, wst);
        wave_fence();
        // next block's input: requested here, behind the filter rounds (whose 40-odd window registers leave no room
        // for six more), and moved into x[] at the end of the block, behind the framer
        constexpr int PF = kDiscOut / LPC;
        float pf[PF];
        float noff = 0.0f;
        if (b + 1 < bend) {
            const float *nx = dsrc + (size_t)(b + 1) * kDiscOut;
            noff = osrc ? ld_off(b + 1) : 0.0f;
#pragma unroll
            for (int r = 0; r < PF; ++r) pf[r] = __builtin_nontemporal_load(&nx[gl + LPC * r]);   // read once
        }
        wv_framer_block(t, o, n, b, hb, gl, rg, wst);

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
    WSTAMP(5);
#endif
    // ---- store state in the reference's layout
    wv_store_state(t, cs, counts, chan, ext_lock, hb, gl);
    if (gl < kTaps - 1) cs.buff[gl + 1] = my.x[gl];
#ifdef M17_STAMPS
    // epilogue (state out, every store of the wave complete: what s_endpgm waits for), the wave's life on the chip-wide
    // 100 MHz clock and in s_memtime ticks (scripts/exp_stamps_wave.py, scripts/exp_clock.py)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    {
        const unsigned now_ = (unsigned)__builtin_amdgcn_s_memtime();
        if (gl == 0) wstamps[wave][9] = now_ - t.last_;
    }
    wave_fence();
    if (chan < 4096 && gl < 8) g_chan_stamps[chan][gl] = wstamps[wave][gl < 7 ? gl : 8];
    if (chan < 4096 && gl == 0) g_chan_stamps_x[chan][0] = wstamps[wave][9];
    if (chan < 16384 && gl == 0) {
        g_wave_span[chan][0] = rt_entry_; g_wave_span[chan][1] = __builtin_amdgcn_s_memrealtime();
        g_wave_span[chan][2] = (unsigned long long)__builtin_amdgcn_s_getreg((4 /*HW_ID*/) | (0 << 6) | (31 << 11)) |
                               ((unsigned long long)((unsigned)__builtin_amdgcn_s_memtime() - t_entry_) << 32);
    }
#endif
}

// Two builds of the kernel.  <0, 6>: the filter of a round as one statement with the branch's 62 tap registers and 16 window
// registers: 106 SGPRs, which admit six waves per SIMD (a seventh changes nothing, eight with registers in scratch are
// slower: rounds 3 and 4).  <1, 8> (round 5): taps and window through half the registers, twice per round
// (fir_window_s<1>): 78 SGPRs and 61 VGPRs, EIGHT waves per SIMD -- 8,192 wave slots, two even generations of 16,384 channels.
template <int HALF, int OCC>
__global__ __launch_bounds__(64 * WV_WAVES, OCC)
void k_sync_frame_wave(const float *__restrict__ disc,     // [C][nblk][384]
                       const float *__restrict__ offs,     // [C][nblk] or null (already DC-free)
                       ChanState *__restrict__ st, int C, int nblk, int mode, int ext_lock,
                       m17gpu_rec_dev *__restrict__ recs, int rec_cap, int32_t *__restrict__ counts,
                       float *__restrict__ syms, int32_t *__restrict__ nsyms,
                       float *__restrict__ fsym, int b0, int bcount)
{
    __shared__ __attribute__((aligned(4096))) WvChan chs[WV_WAVES];
    const int wave = uni((int)(threadIdx.x >> 6));
    sync_wave_channel<HALF>(disc, offs, st, C, nblk, mode, ext_lock, recs, rec_cap, counts, syms, nsyms, fsym, b0, bcount,
                            (int)blockIdx.x * WV_WAVES + wave, chs[wave], wave);
}

#ifdef M17_STAMPS
// EXPERIMENT (round 4, profiles/r04_pc_mock.txt): what a producer / consumer workgroup could reach at best -- twelve
// waves run the timing kernel's whole call for twelve channels, four run the front end's tiles for the same number of
// channel-blocks, with no hand-over between them (each role reads and writes HBM as its stand-alone kernel does): the
// same work as the two kernels of a step, co-resident at the register count of the larger role.
constexpr int PC_CONS = 12, PC_PROD = 4;
__global__ __launch_bounds__(64 * (PC_CONS + PC_PROD), 4)
void k_pc_mock(const uint4 *__restrict__ iq, float *__restrict__ disc_w, float *__restrict__ offs_w,
               const float *__restrict__ disc, const float *__restrict__ offs,
               ChanState *__restrict__ st, ChanState *__restrict__ st_fe, int C, int nblk,
               float *__restrict__ syms, int32_t *__restrict__ nsyms, float *__restrict__ fsym, int32_t *__restrict__ counts)
{
    __shared__ __attribute__((aligned(4096))) WvChan chs[PC_CONS];
    __shared__ __attribute__((aligned(16))) uint32_t tile[PC_PROD][16 * FQ_STRIDE];
    __shared__ __attribute__((aligned(16))) float otile[PC_PROD][16 * FQ_STRIDE];
    const int wave = uni((int)(threadIdx.x >> 6));
    if (wave < PC_CONS) {
        sync_wave_channel(disc, offs, st, C, nblk, 0, -1, nullptr, 0, counts, syms, nsyms, fsym, 0, nblk,
                          (int)blockIdx.x * PC_CONS + wave, chs[wave], wave);
    } else {
        // this workgroup's share of the front end: PC_CONS channels x nblk blocks = PC_CONS * nblk / 16 tiles, dealt to the producers in turn
        const int p = wave - PC_CONS, total = C * nblk;
        const int tiles_per_wg = (PC_CONS * nblk + 15) / 16;
        for (int k = p; k < tiles_per_wg; k += PC_PROD)
            frontend_d_tile(iq, st_fe, disc_w, offs_w, nblk, total, 0, ((int)blockIdx.x * tiles_per_wg + k) * 16, tile[p], otile[p]);
    }
}

#endif // M17_STAMPS (experiments stay out of the shipped library)

} // namespace m17dev
