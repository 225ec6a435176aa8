// m17_fused.hip -- k_rx_fused: the whole FIR stage of ONE CHANNEL IN ONE WAVE -- int16 IQ -> limiter -> FM
// discriminator -> /5 -> DC removal (dsp_short_to_float m17_dsp.cpp:136-141, dsp_limit :412-419, dsp_arctan_disc2
// :194-222), then the polyphase timing loop (m17_rx_sync_samples m17_rx_sync.cpp:77-99), the sync correlator and the
// framer (m17_rx_frame.cpp:47-177) -- with the discriminator samples handed over in LDS.
//
// Why: as two kernels the stage wrote the discriminator stream to HBM and read it back
// (1,536 + 1,536 B per channel-block next to 7,680 B of input: 27 % of the stage's HBM bytes), and its two halves sit
// on different roofs -- the front end on vector-memory throughput with three quarters of its issue slots idle, the
// timing loop on issue / dependency latency with the memory pipe idle -- which separate launches cannot overlap: the
// front end's 124 registers x 4 waves leave no room beside it on a SIMD.  Here every wave runs BOTH phases for its
// own channel, so at any moment some waves of a SIMD stream IQ while the others filter, and nothing but the IQ (in)
// and the frame slots / records / symbols (out) crosses HBM.
//
// What makes that possible is the row mapping of the front-end phase.  The 1920-term DC sum of a block is a strictly
// sequential fp32 chain (m17_dsp.cpp:211), affordable only with several chains side by side in the lanes of one
// instruction; the two-kernel front end found them in 16 (channel, block) rows of OTHER channels.  A channel's own
// blocks are just as independent of each other -- the only state the front end carries is z[0], z[1], the last two
// limited samples, and those are in the input -- so a wave takes FU_R = 4 consecutive blocks of ITS channel as four
// rows of 16 lanes:
//   load    : lane (r, l) of chunk c reads the uint4 that holds samples 64c + 4l .. + 3 of row r -- 256 contiguous
//             bytes per row per instruction; two register sets of FU_P chunks, one being computed while the other is
//             in flight (a rotating set made the compiler copy registers and drain vmcnt at every loop latch), the
//             first pass of the next group requested before the timing phase;
//   compute : the lane converts, limits and discriminates its own four samples, the two before them by DPP row_shr:1
//             (lane 0: the row's previous chunk, row_ror:1); no LDS transpose;
//   DC sum  : the chain runs lane by lane through the row as 16 x 4 dependent adds per chunk: lane l adds its four
//             values to lane l-1's sum (v_add_f32_dpp row_shr:1); lanes ahead of the front compute on stale input
//             and are overwritten when the front reaches them, lane 0 adds exact zeros behind its own step;
//   /5 pick : sample s is an output iff s % 5 == 4; with q = (c + l) % 5 the lane's pick is its sample q - 1
//             (none for q == 0) and goes straight to its place in the row's x[] in LDS;
//   then the row's offset is subtracted in place and the timing loop runs over the four blocks with the delay line
//   simply continuing from one row into the next.
// The chain costs 480 wave-instructions per channel-block here against 120 with 16 rows per wave: the price of
// having both phases in one wave.
//
// Measured (profiles/r04_fused_*): bit-exact; HBM-side traffic of the FIR stage 2.39 -> 1.73 GB
// per launch at 16,384 x 12; 0.669 ms against 0.580 ms for front end + timing kernel on the same box, and further behind at
// small channel counts.  A wave spends 17 k ticks per block in the front-end phase and 15 k in the timing phase, one after
// the other, at four waves per SIMD.  Not the default (option fir_impl 2); kept under the parity tests.
#pragma clang fp contract(off)

namespace m17dev {

constexpr int FU_R = 4;                        // blocks per group = rows of the front-end phase
constexpr int FU_P = 5;                        // chunks per pass: while a pass is computed, the next pass's input is in flight
constexpr int FU_NCHUNK = kBlockSamples / 64;  // 30 chunks of 64 samples per row
constexpr int FU_WAVES = 4;                    // channels (waves) per workgroup; the waves never synchronise
// x[] of one channel: [2 pad][30 delay line][FU_R x 384 block inputs][124 read past the last block, never used]
constexpr int FU_XF = 2 + (kTaps - 1) + FU_R * kDiscOut + 124 + 4;
static_assert(FU_XF % 4 == 0 && FU_NCHUNK % (2 * FU_P) == 0, "fused kernel layout");

// (dpp_keep, fu_left: m17_kernels.hip, next to frontend_quick4p, which shares them)

// The DC sum of one chunk (m17_dsp.cpp:211: offset += out, strictly in sample order): on entry `carry` holds, in lane
// 0 of each row, the row's sum so far; u0..u3 the lane's four values, a0..a3 the same with exact zeros in lane 0.
// Lane 0 finishes in the first four adds; after step j lanes 0..j hold their final sums (lane l <= j recomputes the
// same value from lane l-1's final one; lane 0 is disabled for the DPP add -- no source, bound_ctrl 0 -- and adds
// zeros, which is exact: a running sum that starts at +0 never is -0).  Leaves the row's new sum in lane 0 of `carry`.
// (s_nop 1: a VALU write followed by a DPP read of the same register needs two wait states on gfx9.)
#define FU_STEP "s_nop 1\n\tv_add_f32_dpp %0, %0, %6 row_shr:1 row_mask:0xf bank_mask:0xf\n\t" \
                "v_add_f32 %0, %0, %7\n\tv_add_f32 %0, %0, %8\n\tv_add_f32 %0, %0, %9\n\t"
__device__ __forceinline__ void fu_chain(float &carry, float u0, float u1, float u2, float u3, float a0, float a1, float a2, float a3)
{
    float T;
    asm volatile("v_add_f32 %0, %1, %2\n\tv_add_f32 %0, %0, %3\n\tv_add_f32 %0, %0, %4\n\tv_add_f32 %0, %0, %5\n\t"
                 FU_STEP FU_STEP FU_STEP FU_STEP FU_STEP FU_STEP FU_STEP FU_STEP FU_STEP FU_STEP FU_STEP FU_STEP FU_STEP FU_STEP FU_STEP
                 "s_nop 1\n\tv_mov_b32_dpp %1, %0 row_ror:1 row_mask:0xf bank_mask:0xf"
                 : "=&v"(T), "+v"(carry) : "v"(u0), "v"(u1), "v"(u2), "v"(u3), "v"(a0), "v"(a1), "v"(a2), "v"(a3));
}

__global__ __launch_bounds__(64 * FU_WAVES, 4)
void k_rx_fused(const uint4 *__restrict__ iq,              // [C][nblk][480] uint4 (4 IQ samples each)
                ChanState *__restrict__ st, int C, int nblk, int mode,
                m17gpu_rec_dev *__restrict__ recs, int rec_cap, int32_t *__restrict__ counts,
                float *__restrict__ syms, int32_t *__restrict__ nsyms, float *__restrict__ fsym)
{
    constexpr int LPC = 64;
    __shared__ __attribute__((aligned(2048))) float rings[FU_WAVES][kWvRing];
    __shared__ __attribute__((aligned(16))) float xs[FU_WAVES][FU_XF];
    const int wave = uni((int)(threadIdx.x >> 6)), gl = lane_id();
    const int chan = (int)blockIdx.x * FU_WAVES + wave;
    if (chan >= C) return;
    const unsigned hb = (unsigned)uni((int)(unsigned)(uintptr_t)(lds_cfp)rings[wave]);
    float *const X = &xs[wave][2];                                    // x[0..29] delay line, x[30 + 384 r + i] row r
    const unsigned xb = (unsigned)uni((int)(unsigned)(uintptr_t)(lds_cfp)X);
    ChanState &cs = st[chan];
    if (!recs) rec_cap = 0;

    WvCtl t;
    wv_load_state(t, cs, counts, chan, 0, hb, gl);
    RegroupLane<LPC> rg;
    rg.load(gl);
    WvOut o;
    o.crecs = recs ? recs + (size_t)chan * rec_cap : nullptr; o.rec_cap = rec_cap;
    o.sym_out = syms ? syms + (size_t)chan * M17_SYM_STRIDE(nblk) : nullptr;
    o.nsyms_row = nsyms ? nsyms + (size_t)chan * nblk : nullptr;
    o.fsym_chan = fsym + (size_t)chan * rec_cap * kSlotFloats;
    o.mode = mode; o.ext_lock = -1;
    if (gl < kTaps - 1) X[gl] = cs.buff[gl + 1];
    unsigned *wst = nullptr;
#ifdef M17_STAMPS
    __shared__ unsigned wstamps[FU_WAVES][12];
    if (gl < 12) wstamps[wave][gl] = 0;
    wst = wstamps[wave];
    t.last_ = (unsigned)__builtin_amdgcn_s_memtime();
#endif

    // ---- front-end phase set-up: lane (r, l)
    const int r = gl >> 4, l = gl & 15;
    const uint4 *const iqc = iq + (size_t)chan * nblk * (kBlockSamples / 4);
    // z[0], z[1] for the NEXT call: the limited last two samples of the channel's last block (m17_dsp.cpp:196,205-206)
    float n0re, n0im, n1re, n1im;
    fe_next_z(iq, chan * nblk, nblk, n0re, n0im, n1re, n1im);
    const float s0re = cs.z0re, s0im = cs.z0im, s1re = cs.z1re, s1im = cs.z1im;

    typedef uint32_t u4v __attribute__((ext_vector_type(4)));
    const int ngroups = (nblk + FU_R - 1) / FU_R;
    // the lane's uint4 of chunk cc (counted through the whole call: group cc / 30, chunk cc % 30); rows beyond the call
    // read the group's first row instead (their results are never used)
    auto load_chunk = [&](int cc) {
        const int g = cc / FU_NCHUNK, c = cc - g * FU_NCHUNK;
        int blk = g * FU_R + r;
        blk = blk < nblk ? blk : g * FU_R;
        return __builtin_nontemporal_load(reinterpret_cast<const u4v *>(iqc + (size_t)blk * (kBlockSamples / 4) + c * 16 + l));
    };
    const int total_chunks = ngroups * FU_NCHUNK;
    u4v wa[FU_P], wb[FU_P];
#pragma unroll
    for (int j = 0; j < FU_P; ++j) wa[j] = load_chunk(j);         // total_chunks >= 30 > FU_P

    for (int g = 0; g < ngroups; ++g) {
        WSTAMP(5);
        const int blk = g * FU_R + r;
        const bool valid = blk < nblk;
        // the two limited samples in front of the row: channel state for the call's first block, else the input itself
        float p3re, p3im, p2re, p2im;                       // "previous chunk" values: sample -1 (z0) and -2 (z1), in every lane
        if (blk == 0 || !valid) { p3re = s0re; p3im = s0im; p2re = s1re; p2im = s1im; }
        else {
            const uint32_t *pw = reinterpret_cast<const uint32_t *>(iqc) + (size_t)blk * kBlockSamples;
            const uint32_t a = pw[-2], b = pw[-1];
            p2re = s16_to_float((int)(short)(a & 0xFFFF)); p2im = s16_to_float((int)a >> 16);
            p3re = s16_to_float((int)(short)(b & 0xFFFF)); p3im = s16_to_float((int)b >> 16);
            limit(p2re, p2im);
            limit(p3re, p3im);
        }
        float carry = 0.0f;                                  // offset = 0 (m17_dsp.cpp:199)
        int q = l % 5;                                       // (c + l) % 5
        const unsigned rowb = xb + 4u * (unsigned)((kTaps - 1) + kDiscOut * r);       // LDS byte address of the row's x[30]
        const bool first = l == 0;
        // one chunk: the lane's four samples v of chunk c
        auto chunk = [&](const u4v v, const int c) {
            const uint32_t ww[4] = {v.x, v.y, v.z, v.w};
            float re[4], im[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                re[k] = s16_to_float((int)(short)(ww[k] & 0xFFFF));
                im[k] = s16_to_float((int)ww[k] >> 16);
                limit(re[k], im[k]);
            }
            const float m1re = fu_left(re[3], p3re), m1im = fu_left(im[3], p3im);     // sample -1 of this lane's run
            const float m2re = fu_left(re[2], p2re), m2im = fu_left(im[2], p2im);     // sample -2
            p3re = re[3]; p3im = im[3]; p2re = re[2]; p2im = im[2];
            float u[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                // dsp_arctan_disc2 (m17_dsp.cpp:194-222): z0 = sample k-1, z1 = sample k-2
                const float z0re = k >= 1 ? re[k >= 1 ? k - 1 : 0] : m1re, z0im = k >= 1 ? im[k >= 1 ? k - 1 : 0] : m1im;
                const float z1re = k >= 2 ? re[k >= 2 ? k - 2 : 0] : (k == 1 ? m1re : m2re);
                const float z1im = k >= 2 ? im[k >= 2 ? k - 2 : 0] : (k == 1 ? m1im : m2im);
                const float aa = z0im * (re[k] - z1re);
                const float bb = z0re * (im[k] - z1im);
                u[k] = (bb - aa) * 0.5f;
            }
            fu_chain(carry, u[0], u[1], u[2], u[3], first ? 0.0f : u[0], first ? 0.0f : u[1], first ? 0.0f : u[2], first ? 0.0f : u[3]);
            // count % 5 == 0 pick (m17_dsp.cpp:207-210; 1920 % 5 == 0 keeps the phase from block to block)
            const float pick = q == 1 ? u[0] : (q == 2 ? u[1] : (q == 3 ? u[2] : u[3]));
            const unsigned oidx = ((unsigned)(64 * c + 4 * l + q - 5) * 52429u) >> 18;        // (s - 4) / 5, s = 64c + 4l + q - 1
            if (q != 0 && valid) *(lds_f *)(uintptr_t)(rowb + 4u * oidx) = pick;
            q = (q == 4) ? 0 : q + 1;
        };
        for (int c0 = 0; c0 < FU_NCHUNK; c0 += 2 * FU_P) {
            // pass A: chunks c0 .. c0 + P - 1 from wa, pass B's input requested first
#pragma unroll
            for (int j = 0; j < FU_P; ++j) wb[j] = load_chunk(g * FU_NCHUNK + c0 + FU_P + j);
#pragma unroll
            for (int j = 0; j < FU_P; ++j) chunk(wa[j], c0 + j);
            // pass B: chunks c0 + P .. c0 + 2P - 1 from wb; the next pass A (of the next group behind the last one: it
            // stays in flight through the timing phase)
            {
                const int cc = g * FU_NCHUNK + c0 + 2 * FU_P;
                if (cc < total_chunks) {
#pragma unroll
                    for (int j = 0; j < FU_P; ++j) wa[j] = load_chunk(cc + j);
                }
            }
#pragma unroll
            for (int j = 0; j < FU_P; ++j) chunk(wb[j], c0 + FU_P + j);
        }
        // offset / len, out[i] -= offset (m17_dsp.cpp:213,217-219): the row's sum sits in lane 0 of `carry`
        const float off = __int_as_float(__builtin_amdgcn_ds_bpermute((gl & 48) << 2, __float_as_int(carry))) / (float)kBlockSamples;
        wave_fence();
        {
            float4 *row4 = reinterpret_cast<float4 *>(X + (kTaps - 1) + kDiscOut * r);
#pragma unroll
            for (int j = 0; j < kDiscOut / 64; ++j) {
                float4 d = row4[j * 16 + l];
                d.x = d.x - off; d.y = d.y - off; d.z = d.z - off; d.w = d.w - off;
                if (valid) row4[j * 16 + l] = d;
            }
        }
        wave_fence();
        WSTAMP(6);

        // ---- timing loop + framer over the group's blocks; row rr's delay line is the tail of row rr - 1
        const int nrows = min(FU_R, nblk - g * FU_R);
        for (int rr = 0; rr < nrows; ++rr) {
            const int n = wv_timing_block(t, xb + 4u * (unsigned)(kDiscOut * rr), hb, gl, t.flock, wst);
            wave_fence();
            wv_framer_block(t, o, n, g * FU_R + rr, hb, gl, rg, wst);
            wave_fence();
        }
        // delay line of the next group: the last 30 inputs
        {
            const float keep_x = (gl < kTaps - 1) ? X[kDiscOut * nrows + gl] : 0.0f;
            wave_fence();
            if (gl < kTaps - 1) X[gl] = keep_x;
            wave_fence();
        }
    }
#ifdef M17_STAMPS
    WSTAMP(5);
    if (chan < 4096 && gl < 8) g_chan_stamps[chan][gl] = wstamps[wave][gl < 7 ? gl : 8];
#endif
    wv_store_state(t, cs, counts, chan, -1, hb, gl);
    if (gl < kTaps - 1) cs.buff[gl + 1] = X[gl];
    if (gl == 0) { cs.z0re = n0re; cs.z0im = n0im; cs.z1re = n1re; cs.z1im = n1im; }
}


// ---------------------------------------------------------------------------------------------------------------
// k_rx_chan (round 5; option fir_impl 3): the FIR stage of ONE CHANNEL IN ONE WAVE with the two-kernel stage's own
// device functions -- frontend_d_tile over sixteen of the channel's OWN blocks as the tile's rows (the strictly
// sequential DC chain then serves sixteen rows per instruction, as in k_frontend_d; k_rx_fused above has four), the
// discriminator rows handed from the wave to itself through its slice of the workspace (written, drained, read back
// by the same wave: L2 / Infinity Cache traffic, no other CU involved, no fence), then sync_wave_channel over those
// blocks.  What it is for: a wave alternates between a phase that waits on memory and a phase that waits on issue
// slots, and the waves of a SIMD drift apart, so the two phases overlap on the chip -- which two launches cannot do
// (DESIGN.md section 6).  LDS: the front end's two tiles and the timing loop's WvChan share one region.
// ---------------------------------------------------------------------------------------------------------------
constexpr int RC_WAVES = 4;
constexpr int RC_LDS = 10240;                  // per wave: >= 2 x 16 x FQ_STRIDE x 4 = 8,704 B and >= sizeof(WvChan); a multiple of 2 KB (ring alignment)
static_assert(RC_LDS >= 2 * 16 * FQ_STRIDE * 4 && RC_LDS >= (int)sizeof(WvChan) && RC_LDS % 2048 == 0, "k_rx_chan LDS layout");

// the lane number as a value the compiler cannot see through: everything a phase derives from it (row and tile
// addresses, lane masks, the framer's regroup bytes) is then computed at the head of that phase instead of being
// hoisted in front of the group loop and held across the other phase (128 VGPRs and 24 dwords of scratch that way)
__device__ __forceinline__ int rc_lane()
{
    int l = lane_id();
    asm volatile("" : "+v"(l));
    return l;
}

// Measured alternatives (round 5, profiles/r05_rx_chan_variants.txt), none kept: several channels per wave so that twelve-block
// calls fill whole tiles (4 channels = 48 rows = 3 tiles: 4,096 waves, one generation, every wave in the same phase at the
// same time -- no faster than two kernels); the workgroup's waves sharing the tiles of its channels behind a barrier (the
// barrier puts the two waves of a SIMD into the same phase: +3 % at sixteen blocks); the upper half of a workgroup's
// waves running its tiles one group ahead (-1..2 % at 32 and 48 blocks, nothing at 16).
#ifdef M17_STAMPS
__device__ unsigned long long g_rc_stamps[16384][4];          // per channel: ticks in front-end tiles, in timing phases, realtime in / out
#endif
__global__ __launch_bounds__(64 * RC_WAVES, 4)
void k_rx_chan(const uint4 *__restrict__ iq, ChanState *__restrict__ st, float *__restrict__ disc, float *__restrict__ offs,
               int C, int nblk, int mode, m17gpu_rec_dev *__restrict__ recs, int rec_cap, int32_t *__restrict__ counts,
               float *__restrict__ syms, int32_t *__restrict__ nsyms, float *__restrict__ fsym)
{
    __shared__ __attribute__((aligned(4096))) unsigned char lds[RC_WAVES][RC_LDS];
    const int wave = uni((int)(threadIdx.x >> 6));
    const int chan = (int)blockIdx.x * RC_WAVES + wave;
    if (chan >= C) return;
    uint32_t *tile = reinterpret_cast<uint32_t *>(lds[wave]);
    float *otile = reinterpret_cast<float *>(lds[wave] + 16 * FQ_STRIDE * 4);
    WvChan &wc = *reinterpret_cast<WvChan *>(lds[wave]);
    const int row0 = chan * nblk, row_end = row0 + nblk;      // the channel's rows of the [C * nblk] row space
#ifdef M17_STAMPS
    unsigned long long t_fe = 0, t_tm = 0, t_last = __builtin_amdgcn_s_memtime();
    const unsigned long long rt_in = __builtin_amdgcn_s_memrealtime();
#endif
    for (int b0 = 0; b0 < nblk; b0 += 16) {
        // rows b0 .. b0 + 15 of this channel (rows past its last block are computed on its last row and never stored)
        frontend_d_tile(iq, st, disc, offs, nblk, row_end, 1, row0 + b0, tile, otile, rc_lane());
        // the rows must be in memory before this wave reads them back (same wave, same addresses: its own stores are
        // ordered behind its vmcnt; the loads of the timing phase are non-temporal, served by L2)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        wave_lds_sync();
#ifdef M17_STAMPS
        { const unsigned long long now = __builtin_amdgcn_s_memtime(); t_fe += now - t_last; t_last = now; }
#endif
        sync_wave_channel<0>(disc, offs, st, C, nblk, mode, -1, recs, rec_cap, counts, syms, nsyms, fsym, b0, min(16, nblk - b0),
                             chan, wc, wave, rc_lane());
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // channel state out before the next group reads it
        wave_lds_sync();
#ifdef M17_STAMPS
        { const unsigned long long now = __builtin_amdgcn_s_memtime(); t_tm += now - t_last; t_last = now; }
#endif
    }
#ifdef M17_STAMPS
    if (chan < 16384 && lane_id() == 0) {
        g_rc_stamps[chan][0] = t_fe; g_rc_stamps[chan][1] = t_tm; g_rc_stamps[chan][2] = rt_in; g_rc_stamps[chan][3] = __builtin_amdgcn_s_memrealtime();
    }
#endif
}


// k_rx_chan6 (round 5; option fir_impl 4; the library's choice from 10,000 channels on for calls of at least twelve
// blocks): k_rx_chan built for SIX waves per SIMD -- frontend_lite_tile (32-sample chunks, ~60 VGPRs, 4.6 KB of LDS) and the
// timing loop with taps and window through half the registers (sync_wave_channel<1>).  Measured: the
// same time as k_rx_chan at four waves per SIMD -- the stage is bound by the vector ALU (~76 % busy) and by its traffic
// past L2 (4.7-5.1 TB/s), not by what more waves would cover; it is the default of the two for the registers and LDS it
// leaves.
constexpr int RC6_LDS = 6144;                  // per wave: the two 2,304-byte tiles / the timing loop's WvChan (4 KB); a multiple of 2 KB (ring alignment)
static_assert(RC6_LDS >= 2 * FL_TILE_BYTES && RC6_LDS >= (int)sizeof(WvChan) && RC6_LDS % 2048 == 0, "k_rx_chan6 LDS layout");
__global__ __launch_bounds__(64 * RC_WAVES, 6)
void k_rx_chan6(const uint4 *__restrict__ iq, ChanState *__restrict__ st, float *__restrict__ disc, float *__restrict__ offs,
                int C, int nblk, int mode, m17gpu_rec_dev *__restrict__ recs, int rec_cap, int32_t *__restrict__ counts,
                float *__restrict__ syms, int32_t *__restrict__ nsyms, float *__restrict__ fsym)
{
    __shared__ __attribute__((aligned(4096))) unsigned char lds[RC_WAVES][RC6_LDS];
    const int wave = uni((int)(threadIdx.x >> 6));
    const int chan0 = (int)blockIdx.x * RC_WAVES;      // the workgroup's first channel (always < C)
    const int chan = chan0 + wave;
    const bool live = chan < C;                         // no early exit: a group of fewer than sixteen blocks has a workgroup barrier
    uint32_t *tile = reinterpret_cast<uint32_t *>(lds[wave]);
    float *otile = reinterpret_cast<float *>(lds[wave] + FL_TILE_BYTES);
    WvChan &wc = *reinterpret_cast<WvChan *>(lds[wave]);
    const int row0 = chan * nblk;
#ifdef M17_STAMPS
    unsigned long long t_fe = 0, t_tm = 0, t_last = __builtin_amdgcn_s_memtime();
    const unsigned long long rt_in = __builtin_amdgcn_s_memrealtime();
#endif
    for (int b0 = 0; b0 < nblk; b0 += 16) {
        const int bc = min(16, nblk - b0);
        float *const dw = disc, *const ow = offs;
        if (bc == 16) {
            // rows b0 .. b0 + 15 of this channel: written and read back by the same wave
            if (live) {
                frontend_lite_tile(iq, st, dw, ow, nblk, 1, [&](int i, bool &valid) { valid = true; return row0 + b0 + i; }, tile, otile, rc_lane());
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // the rows are in memory before this wave reads them back
                wave_lds_sync();
            }
        } else {
            // A last group of fewer than sixteen blocks: a tile costs the same whatever it holds, so the 4 x bc rows of the
            // workgroup's four channels are packed into ceil(bc / 4) tiles on its first waves -- row j of that list is block
            // b0 + j % bc of channel chan0 + j / bc -- and every wave reads its channel's rows back behind a workgroup barrier
            // (same CU: the rows are whole cache lines nobody has read yet; the offsets are read at agent scope).
            const int ntile = (RC_WAVES * bc + 15) >> 4;
            if (wave < ntile) {
                frontend_lite_tile(iq, st, dw, ow, nblk, 1,
                                   [&](int i, bool &valid) {
                                       const int j = 16 * wave + i, cj = j / bc, ch = chan0 + cj;
                                       valid = cj < RC_WAVES && ch < C;
                                       return valid ? ch * nblk + b0 + (j - cj * bc) : chan0 * nblk + b0;
                                   }, tile, otile, rc_lane());
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            __syncthreads();
        }
#ifdef M17_STAMPS
        { const unsigned long long now = __builtin_amdgcn_s_memtime(); t_fe += now - t_last; t_last = now; }
#endif
        if (live) {
            sync_wave_channel<1, 1>(dw, ow, st, C, nblk, mode, -1, recs, rec_cap, counts, syms, nsyms, fsym, b0, bc,
                                    chan, wc, wave, rc_lane());
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // channel state out before the next group reads it
            wave_lds_sync();
        }
#ifdef M17_STAMPS
        { const unsigned long long now = __builtin_amdgcn_s_memtime(); t_tm += now - t_last; t_last = now; }
#endif
    }
#ifdef M17_STAMPS
    if (chan < 16384 && lane_id() == 0) {
        g_rc_stamps[chan][0] = t_fe; g_rc_stamps[chan][1] = t_tm; g_rc_stamps[chan][2] = rt_in; g_rc_stamps[chan][3] = __builtin_amdgcn_s_memrealtime();
    }
#endif
}

} // namespace m17dev
