// m17_fused.hip -- k_rx_chan6: the whole FIR stage of ONE CHANNEL IN ONE WAVE -- int16 IQ -> limiter -> FM discriminator ->
// /5 -> DC removal (dsp_short_to_float m17_dsp.cpp:136-141, dsp_limit :412-419, dsp_arctan_disc2 :194-222), then the
// polyphase timing loop (m17_rx_sync_samples m17_rx_sync.cpp:77-99), the sync correlator and the framer
// (m17_rx_frame.cpp:47-177).
//
// Why one wave runs both phases (DESIGN.md sections 5 and 6): as two kernels the stage's two halves sit on different roofs
// -- the front end on vector-memory throughput with most of its issue slots idle, the timing loop on issue / dependency
// latency with the memory pipe idle -- and separate launches cannot overlap them.  Here every wave alternates between a
// front-end tile over sixteen of ITS channel's blocks (frontend_lite_tile: the strictly sequential DC chain of
// m17_dsp.cpp:211 serves sixteen rows per instruction, as in the stand-alone front end) and the timing loop + framer over
// those blocks, and the waves of a SIMD drift apart, so at any moment some stream IQ while the others filter.  The
// sixteen discriminator rows travel from the wave to itself through its channel's rows of the workspace (24.5 KB per wave
// do not fit LDS at six waves per SIMD): written, drained and read back by the same wave -- no other CU, no fence.
//
// History: round 4's k_rx_fused (four blocks at a time through LDS: no rows in HBM, but a four-row DC chain at four times
// the instructions per block: 15-18 % slower) and round 5's k_rx_chan (this kernel on 64-sample chunks at four waves per
// SIMD: the same time) were removed in round 6; their measurements are in DESIGN.md section 6 and profiles/r04_fused_*,
// profiles/r05_rx_chan_variants.txt.
#pragma clang fp contract(off)

namespace m17dev {

constexpr int RC_WAVES = 4;                    // channels (waves) per workgroup

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

// k_rx_chan6 (round 5; option fir_impl 4; the library's choice from 10,000 channels on for calls of at least twelve
// blocks): frontend_lite_tile (32-sample chunks, ~60 VGPRs, 4.6 KB of LDS) and the timing loop with taps and window through
// half the registers (sync_wave_channel<1>) -- six waves per SIMD.  Measured: the stage is bound by the
// vector ALU (~76 % busy) and by its traffic past L2 (4.7-5.1 TB/s), not by what more waves would cover.
//
// The tail (round 6).  16,384 waves over 6,144 wave slots are 2.67 generations: for the last third of the launch's span the
// slots empty and its last waves run alone on their SIMDs, latency-bound (residency: profiles/r05_stamps_rx_chan.txt).
// Three ways to shorten or fill it were built, all bit-exact, none faster, none kept (the implementations are commits
// 4cfa6db and 6dc0fdc): dispatching the channels by the work their last call took, heaviest first (a wave's lifetime is set
// by WHEN it runs, not by what it has to do: -1.4 %, profiles/r06_dispatch_order_ab.txt); the call split into two channel
// parts with the decoder of the first on an internal stream beside the FIR stage of the second (+4...8 %:
// profiles/r06_split_call_ab.txt); and the decoder of the launch's first channels started behind a gate kernel while the
// last waves still run, the framer's outputs written through L2 for it (the write-through stores alone cost 8 %:
// profiles/r06_gated_decoder_ab.txt).
constexpr int RC6_LDS = 6144;                  // per wave: the two 2,304-byte tiles / the timing loop's WvChan (4 KB); a multiple of 2 KB (ring alignment)
static_assert(RC6_LDS >= 2 * FL_TILE_BYTES && RC6_LDS >= (int)sizeof(WvChan) && RC6_LDS % 2048 == 0, "k_rx_chan6 LDS layout");
// TAIL: the call's block count is no multiple of sixteen (the host picks the build): only then is there a last group whose
// tiles the workgroup's channels share, with its barrier.
template <int TAIL>
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
        if (!TAIL || bc == 16) {
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
    if (live && chan < 16384 && lane_id() == 0) {
        g_rc_stamps[chan][0] = t_fe; g_rc_stamps[chan][1] = t_tm; g_rc_stamps[chan][2] = rt_in; g_rc_stamps[chan][3] = __builtin_amdgcn_s_memrealtime();
    }
#endif
}

} // namespace m17dev
