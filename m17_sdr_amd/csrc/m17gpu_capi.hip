// m17gpu_capi.hip -- the C-ABI of include/m17gpu.h on top of the gfx950 kernels.
// No CPU fallback: every compute entry point needs a HIP device.
#include "m17_host.h"
#include "m17_kernels.hip"
#include "m17_sync_common.hip"
#include "m17_sync_duo.hip"
#include "m17_sync_wave.hip"
#include "m17_fused.hip"
#include "m17_decode_quad.hip"
#include "m17_book.hip"
#include "m17_pack.hip"
#include "m17_pluto.hip"
#include "m17_gen.hip"
#include "m17_host.h"
#include <cmath>
#include <algorithm>
#include "../../include/m17gpu.h"
#include <string>
#include <algorithm>
#include <vector>
#include <cstring>
#include <cstdio>
#include <cstdlib>
#include <new>
#include <memory>
#include <mutex>
#include <dlfcn.h>
// The few RCCL types the fan-out entries need, as rccl.h declares them (ABI-stable NCCL 2 values): the library is
// resolved in the process at first use, so neither its header nor the library is a build requirement.
typedef struct ncclComm *ncclComm_t;
typedef enum { ncclSuccess = 0 } ncclResult_t;
typedef enum { ncclInt8 = 0, ncclChar = 0, ncclInt32 = 2 } ncclDataType_t;

using namespace m17dev;

static_assert(sizeof(m17gpu_rec) == sizeof(m17gpu_rec_dev), "record layouts must agree");
static_assert(offsetof(m17gpu_rec, data) == offsetof(m17gpu_rec_dev, data), "record layouts must agree");
static_assert(offsetof(m17gpu_rec, variance) == 8 && offsetof(m17gpu_rec, block) == 12 &&
              offsetof(m17gpu_rec, sym_pos) == 16 && offsetof(m17gpu_rec, data) == 20, "record word layout");

struct m17gpu_ctx {
    int device = 0, C = 0, max_blocks = 0, rec_cap_max = 0;
    ChanState *d_state = nullptr;
    float *d_disc = nullptr, *d_offs = nullptr, *d_fsym = nullptr;
    int32_t *d_work = nullptr, *d_nwork = nullptr, *d_counts = nullptr;
    uint16_t *d_genc = nullptr, *d_gerr = nullptr, *d_crc_basis = nullptr;
    uint32_t *d_dec_hist = nullptr;          // [C][32] history of the wide-band decimator
    uint8_t *d_net = nullptr;                // optional network sink [C][net_rec_cap][56] (m17gpu_set_net_output); not owned
    int net_rec_cap = 0;                     // its records per channel: a mode-1 call must use exactly this rec_cap
    const uint16_t *d_stream_ids = nullptr;  // optional [C] stream-id base per channel; not owned
    unsigned long long dst_override = 0;     // 48-bit destination callsign written into every net frame (0 = keep the LSF's)
    int afc = 0;                             // 1 = AFC on (radio_set_afc_on): block-sequential front end
    bool profiling = false;
    int fe_impl = 0;                         // stand-alone front end: 0 | 2 = four lanes per channel-block, LDS chain (k_frontend_q, the default);
                                             // 3 = the sixteen-row tile of k_sync_frame_duo<1> as a kernel (k_frontend_d: DC chain and picks in registers);
                                             // 4 = the tile of k_rx_chan6 as a kernel (k_frontend_l: 32-sample chunks)
    int sync_impl = 0;                       // 0 | 6 = timing wave + framer wave per channel up to 1,024 channels, wave per channel beyond (default);
                                             // 8 = wave per channel at every size
    int fe_debug = 0;                        // instrumented build only (scripts/exp_fe_bound.py)
    int book_impl = 0;                       // bookkeeping kernel: 0 = by batch (a lane per channel from 8,192 channels on, else a wave per channel), 1 = a wave
                                             // per channel (k_book_chan; always with the network sink attached), 2 = a lane per channel (k_book_lanes)
    int slot_impl = 0;                       // stream frame slots: 1 = plain (768 B, regrouped in the decoder's LDS), 2 = regrouped by the framer (1,600 B),
                                             // 0 = by path: plain behind the wave-per-channel FIR stage, regrouped behind front end + timing kernel
    int32_t *d_flags = nullptr;              // [n_flags] verdict words of m17gpu_shard_gather_packed
    int n_flags = 0;
    int fir_impl = 0;                        // 0 = by call (fir_choice); 1 = front end + timing kernel; 4 = wave per channel over sixteen-row
                                             // tiles of its own blocks, rows through the workspace, six waves per SIMD (k_rx_chan6);
                                             // 5 = three waves per channel, the front end one of them (k_sync_frame_duo<1>, up to 1,024 channels)
    bool nwork_dirty = false;                // a full-chain call got past k_worklist but not to the bookkeeping kernel that zeroes the counters
    int last_path[4] = {0, 0, 0, 0};         // what the last m17gpu_rx_blocks call ran: FIR stage (fir_choice), plain slots, bookkeeping kernel, (reserved)
    std::vector<hipEvent_t> ev_pool;         // 7 events per profiled call: 5 stage marks + call start / end
    std::vector<int> ev_mode;                // mode of each profiled call
};

namespace {

thread_local std::string g_err;

int fail(int code, const std::string &msg) { g_err = msg; return code; }

#define HIPCHK(expr)                                                                   \
    do {                                                                               \
        hipError_t e_ = (expr);                                                        \
        if (e_ != hipSuccess)                                                          \
            return fail(M17GPU_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_)); \
    } while (0)

// Every entry point runs on the context's device whatever the caller's current device is, and
// leaves the caller's device selection as it found it (the __constant__ tables and all buffers of a
// context live on ctx->device only).
struct DeviceScope {
    int prev = -1, want = -1;
    bool ok = true;
    explicit DeviceScope(int dev) : want(dev)
    {
        if (hipGetDevice(&prev) != hipSuccess) { ok = false; return; }
        if (prev != want && hipSetDevice(want) != hipSuccess) ok = false;
    }
    ~DeviceScope() { if (ok && prev != want) (void)hipSetDevice(prev); }
};
#define ON_CTX_DEVICE(ctx)                                                             \
    DeviceScope dev_scope_((ctx)->device);                                             \
    if (!dev_scope_.ok) return fail(M17GPU_ERR_HIP, "cannot select the context's device")

int upload_tables(m17gpu_ctx *ctx)
{
    const m17::Tables &T = m17::tables();
    // per-call scratch on the heap (too large for the stack of some callers; a function-static one made two host
    // threads creating contexts for two GPUs race)
    std::unique_ptr<DevTables> hp(new (std::nothrow) DevTables);
    if (!hp) return fail(M17GPU_ERR_NOMEM, "upload_tables: out of host memory");
    DevTables &h = *hp;
    std::memset(&h, 0, sizeof h);
    for (int p = 0; p < kPhases; ++p)
        for (int j = 0; j < kTaps; ++j) {
            h.mf[p][j] = T.mf[p][j]; h.md[p][j] = T.md[p][j];
            h.tap_pairs[p][2 * j] = T.mf[p][j]; h.tap_pairs[p][2 * j + 1] = T.md[p][j];
        }
    for (int t = 0; t < 4; ++t) {
        h.glen[t] = T.glen[t];
        for (int k = 0; k < 488; ++k) {
            int16_t g = T.gather[t][k];
            if (t == 0 || k >= T.glen[t]) g = -1;
            else if (g >= 0 && T.gsign[t][k] < 0) g = (int16_t)(g | 0x4000);
            h.gather[t][k] = g;
        }
    }
    for (int j = 0; j < 96; ++j)
    {
        h.lich[j] = (int16_t)(T.lich_src[j] | (T.lich_sign[j] < 0 ? 0x4000 : 0));
        h.lich_q[j] = dq_lich_entry((int)h.lich[j]);
    }
    for (int i = 0; i < kRegroup; ++i) {
        const int g = (i < 96) ? (int)h.lich[i] : (int)h.gather[2][i - 96];
        h.regroup[i] = (uint8_t)(g < 0 ? 0 : ((g & 0x3FF) >> 1));
    }
    std::memcpy(h.bm_even, T.bm_even, 16);
    std::memcpy(h.bm_odd, T.bm_odd, 16);
    std::memcpy(h.crc, T.crc, sizeof h.crc);
    HIPCHK(hipMemcpyToSymbol(HIP_SYMBOL(c_tab), &h, sizeof h));
    HIPCHK(hipMalloc(&ctx->d_genc, 4096 * sizeof(uint16_t)));
    HIPCHK(hipMalloc(&ctx->d_gerr, 4096 * sizeof(uint16_t)));
    HIPCHK(hipMemcpy(ctx->d_genc, T.golay_enc, 4096 * sizeof(uint16_t), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(ctx->d_gerr, T.golay_err, 4096 * sizeof(uint16_t), hipMemcpyHostToDevice));
    // CRC-16 is linear over GF(2): crc(msg) = crc(len zero bytes) xor XOR over set bits of E[distance from the end][bit].
    // Layout of d_crc_basis: [240] the 30-byte case as (byte i, bit k) | [801] Z[L] = crc of L zero bytes |
    // [800][8] E[p][k] for messages up to the 800-byte packet buffer (m17_rx_parse.cpp:6).
    std::vector<uint16_t> basis(240 + 801 + 6400);
    {
        std::vector<uint8_t> msg(801, 0);
        uint16_t *Z = &basis[240], *E = &basis[240 + 801];
        for (int L = 0; L <= 800; ++L) Z[L] = m17::crc16(msg.data(), L);
        for (int p_ = 0; p_ < 800; ++p_)
            for (int k = 0; k < 8; ++k) {
                msg[0] = (uint8_t)(1u << k);
                E[p_ * 8 + k] = (uint16_t)(m17::crc16(msg.data(), p_ + 1) ^ Z[p_ + 1]);
            }
        msg[0] = 0;
        for (int i = 0; i < 30; ++i)
            for (int k = 0; k < 8; ++k) basis[i * 8 + k] = E[(29 - i) * 8 + k];
    }
    int16_t dec[32] = {0};
    m17::build_pluto_dec_filter(dec);
    HIPCHK(hipMemcpyToSymbol(HIP_SYMBOL(c_dec), dec, sizeof dec));
    HIPCHK(hipMalloc(&ctx->d_crc_basis, basis.size() * sizeof(uint16_t)));
    HIPCHK(hipMemcpy(ctx->d_crc_basis, basis.data(), basis.size() * sizeof(uint16_t), hipMemcpyHostToDevice));
    return M17GPU_OK;
}

inline hipStream_t S(void *s) { return reinterpret_cast<hipStream_t>(s); }
inline int cdiv(long long a, long long b) { return (int)((a + b - 1) / b); }

// Channel range [c0, c0 + cn) of the context (cn < 0: all): every array of the path is channel-major, so a range
// is the same launch on offset pointers.
int launch_frontend(m17gpu_ctx *ctx, const int16_t *d_iq, int nblk, float *disc, float *offs,
                    int update_state, hipStream_t st, int c0 = 0, int cn = -1)
{
    if (cn < 0) cn = ctx->C;
    const int total = cn * nblk;
    d_iq += (size_t)c0 * nblk * kBlockSamples * 2;
    disc += (size_t)c0 * nblk * kDiscOut;
    offs += (size_t)c0 * nblk;
    ChanState *state = ctx->d_state + c0;
    // four lanes per channel-block (k_frontend_q) at every size; 3 / 4 run the tiles of the fused kernels as kernels of
    // their own (the lane-per-row kernel of round 1, fe_impl 1, was removed in round 6: slower at every size)
    if (ctx->fe_impl == 4)
        hipLaunchKernelGGL(k_frontend_l, dim3(cdiv(total, 16)), dim3(64), 0, st,
                           reinterpret_cast<const uint4 *>(d_iq), state, disc, offs, nblk, total, update_state);
    else if (ctx->fe_impl == 3)
        hipLaunchKernelGGL(k_frontend_d, dim3(cdiv(total, 16 * FQ_WAVES)), dim3(64 * FQ_WAVES), 0, st,
                           reinterpret_cast<const uint4 *>(d_iq), state, disc, offs, nblk, total, update_state);
    else
        hipLaunchKernelGGL(k_frontend_q, dim3(cdiv(total, 16 * FQ_WAVES)), dim3(64 * FQ_WAVES), 0, st,
                           reinterpret_cast<const uint4 *>(d_iq), state, disc, offs, nblk, total, update_state | (ctx->fe_debug << 1));
    HIPCHK(hipGetLastError());
    return M17GPU_OK;
}

bool plain_slots(const m17gpu_ctx *ctx, int nblk);
int launch_sync_frame(m17gpu_ctx *ctx, const float *disc, const float *offs, int nblk, int mode,
                      m17gpu_rec *d_recs, int rec_cap, int32_t *d_counts, float *d_syms, int32_t *d_nsyms,
                      hipStream_t st, int ext_lock = -1, int b0 = 0, int bcount = -1, int c0 = 0, int cn = -1,
                      const int16_t *pipe_iq = nullptr)
{
    if (bcount < 0) bcount = nblk;
    if (cn < 0) cn = ctx->C;
    disc += (size_t)c0 * nblk * kDiscOut;
    if (offs) offs += (size_t)c0 * nblk;
    ChanState *state = ctx->d_state + c0;
    m17gpu_rec_dev *recs = d_recs ? reinterpret_cast<m17gpu_rec_dev *>(d_recs) + (size_t)c0 * rec_cap : nullptr;
    int32_t *counts = (d_counts ? d_counts : ctx->d_counts) + c0;
    float *syms = d_syms ? d_syms + (size_t)c0 * M17_SYM_STRIDE(nblk) : nullptr;
    int32_t *nsyms = d_nsyms ? d_nsyms + (size_t)c0 * nblk : nullptr;
    float *fsym = ctx->d_fsym + (size_t)c0 * rec_cap * kSlotFloats;
    // two-wave kernel (timing wave + framer wave per channel, m17_sync_duo.hip: one 8-wave workgroup per CU) up to
    // 1,024 channels -- a second workgroup per CU does not fit its registers; beyond that, and for the lock-forced
    // stage entry, which has no framer, one wave per channel with scalar control (m17_sync_wave.hip)
    const bool duo = (ctx->sync_impl == 0 || ctx->sync_impl == 6) && ctx->C <= 1024 && ext_lock < 0 && ctx->slot_impl != 1;
    const int kmode = mode | ((!duo && plain_slots(ctx, nblk)) ? 16 : 0);       // bit 4: plain frame slots (wave kernels only)
    if (duo && pipe_iq)
        // front end, timing loop and framer of a channel on three waves of one workgroup (k_sync_frame_duo<1>)
        hipLaunchKernelGGL(k_sync_frame_duo<1>, dim3(cdiv(cn, 4)), dim3(768), 0, st,
                           disc, offs, state, cn, nblk, mode, recs, recs ? rec_cap : 0,
                           counts, syms, nsyms, fsym, b0, bcount,
                           reinterpret_cast<const uint4 *>(pipe_iq + (size_t)c0 * nblk * kBlockSamples * 2),
                           const_cast<float *>(disc), const_cast<float *>(offs));
    else if (duo)
        hipLaunchKernelGGL(k_sync_frame_duo<0>, dim3(cdiv(cn, 4)), dim3(512), 0, st,
                           disc, offs, state, cn, nblk, mode, recs, recs ? rec_cap : 0,
                           counts, syms, nsyms, fsym, b0, bcount, nullptr, nullptr, nullptr);
    else {
        // 0 / 6 beyond 1,024 channels, and 8: taps and window through half the registers, eight waves per SIMD (round 3's
        // form with all 62 tap registers at six waves per SIMD, and the half-register form at six, were removed in round 6)
        hipLaunchKernelGGL((k_sync_frame_wave<1, 8>), dim3(cdiv(cn, WV_WAVES)), dim3(64 * WV_WAVES), 0, st,
                           disc, offs, state, cn, nblk, kmode, ext_lock, recs, recs ? rec_cap : 0,
                           counts, syms, nsyms, fsym, b0, bcount);
    }
    HIPCHK(hipGetLastError());
    return M17GPU_OK;
}

// Which FIR stage runs.  fir_impl 0 (default): the wave-per-channel kernel k_rx_chan6 where it wins -- calls of whole
// sixteen-block groups (its front-end tiles are sixteen of a channel's own blocks) on batches of at least 10,000 channels
// (its 6,144 wave slots want well over one generation of waves: full chain at 16 blocks per call against front end +
// timing kernel: +3 % at 8,192 channels, -2...-5 % at 10,240, -6 % at 12,288, -9 % at 16,384, -7 % at 32,768,
// profiles/r05_channel_count_crossover.txt) -- and front end + timing kernel otherwise.
int fir_choice(const m17gpu_ctx *ctx, int nblk);
// Stream frames as their 192 symbols (slot_impl 1) or regrouped into decoder order by the framer (2)?  Measured at 16,384
// channels (profiles/r05_plain_frame_slots.txt): the regrouped store costs the framer more than it saves the decoder when
// framer and front end share their waves (k_rx_chan6: -6 % on that kernel, +15 % on the decoder, -1.3 % on the step) and
// is worth its cost behind the stand-alone timing kernel (step +-1 %).  Only the wave-per-channel framers write plain slots.
bool plain_slots(const m17gpu_ctx *ctx, int nblk)
{
    const int fir = fir_choice(ctx, nblk);
    if (fir == 5) return false;                     // the two-wave framer writes regrouped slots only
    if (ctx->slot_impl) return ctx->slot_impl == 1;
    return fir == 4;
}
// 5: front end, timing loop and framer of a channel on three waves of one workgroup (k_sync_frame_duo<1>) -- small
// batches, where one channel per SIMD slot leaves both kernels latency-bound: up to 1,024 channels.  Short calls (under
// sixteen blocks; up to eight start on four-row tiles) at every channel count: one launch instead of two, 10-25 % less
// from 1 x 1 to 1,024 x 12.  Longer calls from 512 channels on: below, the stand-alone front end spreads a channel's
// blocks over the idle chip while the channel's own front-end wave takes them tile after tile (256 x 50: +1 %, 1 x 50:
// +5 %; profiles/r05_three_wave_fir_stage_1024.txt).
int fir_choice(const m17gpu_ctx *ctx, int nblk)
{
    if (ctx->afc) return 1;
    const bool trio_ok = (ctx->sync_impl == 0 || ctx->sync_impl == 6) && ctx->C <= 1024 && ctx->slot_impl != 1;
    if (ctx->fir_impl == 5) return trio_ok ? 5 : 1;
    if (ctx->fir_impl != 0) return ctx->fir_impl;                   // 1 | 4
    if (trio_ok && (nblk < 16 || ctx->C >= 512)) return 5;
    // 4: the wave-per-channel stage works in tiles of sixteen of the channel's blocks; a last group of fewer blocks is packed
    // four channels to a workgroup's tiles (k_rx_chan6), so any call from twelve blocks on is served at the cost its rows
    // need (measured at 16,384 channels: 12 blocks -3 %, 20: -6 %, 24: -6 %, 40: -7 %, 8: even, 10: +3 %;
    // profiles/r05_channel_count_crossover.txt)
    return (ctx->C >= 10000 && nblk >= 12) ? 4 : 1;
}
// the wave-per-channel FIR stage (k_rx_chan6)
int launch_fused(m17gpu_ctx *ctx, const int16_t *d_iq, int nblk, int mode, m17gpu_rec *d_recs, int rec_cap,
                 int32_t *d_counts, float *d_syms, int32_t *d_nsyms, hipStream_t st)
{
    mode |= plain_slots(ctx, nblk) ? 16 : 0;                         // bit 4: plain frame slots
    hipLaunchKernelGGL((nblk % 16) ? k_rx_chan6<1> : k_rx_chan6<0>, dim3(cdiv(ctx->C, RC_WAVES)), dim3(64 * RC_WAVES), 0, st,
                       reinterpret_cast<const uint4 *>(d_iq), ctx->d_state, ctx->d_disc, ctx->d_offs, ctx->C, nblk, mode,
                       reinterpret_cast<m17gpu_rec_dev *>(d_recs), d_recs ? rec_cap : 0, d_counts ? d_counts : ctx->d_counts,
                       d_syms, d_nsyms, ctx->d_fsym);
    HIPCHK(hipGetLastError());
    return M17GPU_OK;
}

// demap / gather / Viterbi / Golay of the frames the framer queued, then the per-channel in-order bookkeeping,
// for the channel range [c0, c0 + cn): its own work lists and counters (slot numbers are relative to the range)
int launch_decode(m17gpu_ctx *ctx, m17gpu_rec *d_recs, int rec_cap, int32_t *d_counts, hipStream_t st,
                  int c0, int cn, int chunk, hipEvent_t ev_mid, bool plain)
{
    m17gpu_rec_dev *recs = reinterpret_cast<m17gpu_rec_dev *>(d_recs) + (size_t)c0 * rec_cap;
    int32_t *cnt = (d_counts ? d_counts : ctx->d_counts) + c0;
    float *fsym = ctx->d_fsym + (size_t)c0 * rec_cap * kSlotFloats;
    int32_t *work = ctx->d_work + 3 * (size_t)c0 * ctx->rec_cap_max;
    int32_t *nwork = ctx->d_nwork + 4 * chunk;
    const long long slots = (long long)cn * rec_cap;
    ctx->nwork_dirty = true;                        // until the bookkeeping kernel, which zeroes the counters, is enqueued
    hipLaunchKernelGGL(k_worklist, dim3(cdiv(slots, 1024)), dim3(1024), 0, st, recs, rec_cap, cnt, cn,
                       work, nwork, (int)slots);
    // one launch for all three lists (k_decode_lists): stream frames, the bulk of any traffic, on five workgroups of
    // four waves per CU (30.5 KB of LDS and 96 VGPRs each); link-setup and packet frames on up to 512 leading workgroups
    const int tasks = cdiv(slots, DQ_FRAMES) + 2;
    int n_other = cdiv(tasks, 2);
    if (n_other > 512) n_other = 512;
    int grid = cdiv(tasks, 4);
    if (grid > 256 * 5) grid = 256 * 5;             // (four or three workgroups per CU: 0.177-0.180 -> 0.196-0.200 / 0.191 ms, round 4)
    if (plain) {
        int gp = cdiv(tasks, 4);
        if (gp > 256 * 2) gp = 256 * 2;             // 76 KB of LDS per workgroup: two per CU
        hipLaunchKernelGGL(k_decode_lists_p, dim3(gp + n_other), dim3(256), 0, st, fsym, work, nwork, (int)slots, recs,
                           ctx->d_genc, ctx->d_gerr, kSlotFloats, n_other);
    } else
    hipLaunchKernelGGL(k_decode_lists, dim3(grid + n_other), dim3(256), 0, st, fsym, work, nwork, (int)slots, recs,
                       ctx->d_genc, ctx->d_gerr, kSlotFloats, n_other);
    HIPCHK(hipGetLastError());
    if (ev_mid) HIPCHK(hipEventRecord(ev_mid, st));
    // bookkeeping: a lane per channel (k_book_lanes) for large batches -- it is ~25 us whatever the batch, the wave-per-channel
    // kernel 19 us at 4,096 channels x 16 blocks, 28 at 8,192, 42 at 16,384, 80 at 32,768 -- unless the network sink is attached
    // (its frames are formatted by the wave-per-channel kernel); book_impl 1 / 2 force one
    // (its LDS holds four words of every record of its eight channels: calls of more than ~190 blocks stay with the wave kernel)
    const bool lanes = !ctx->d_net && (size_t)BL_CH * rec_cap * 16 <= 48 * 1024 && (ctx->book_impl == 2 || (ctx->book_impl == 0 && cn >= 8192));
    ctx->last_path[2] = lanes ? 2 : 1;
    if (lanes)
        hipLaunchKernelGGL(k_book_lanes, dim3(cdiv(cn, BL_CH)), dim3(64), (size_t)BL_CH * rec_cap * 16, st, ctx->d_state + c0, recs, rec_cap, cnt, ctx->d_crc_basis, nwork, cn);
    else
    hipLaunchKernelGGL(k_book_chan, dim3(cn), dim3(64), 0, st, ctx->d_state + c0, recs, rec_cap, cnt, ctx->d_crc_basis,
                       ctx->d_net, ctx->d_stream_ids, ctx->dst_override, c0, nwork);
    HIPCHK(hipGetLastError());
    return M17GPU_OK;
}

} // namespace

extern "C" {

const char *m17gpu_last_error(void) { return g_err.c_str(); }

int m17gpu_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int m17gpu_channels(const m17gpu_ctx *ctx) { return ctx ? ctx->C : 0; }

int m17gpu_create(m17gpu_ctx **out, int n_channels, int max_blocks, int device)
{
    if (!out || n_channels <= 0 || max_blocks <= 0) return fail(M17GPU_ERR_ARG, "m17gpu_create: bad argument");
    if (m17gpu_device_count() <= 0)
        return fail(M17GPU_ERR_NO_DEVICE, "m17gpu_create: no HIP device visible (there is no CPU fallback)");
    if (device < 0 || device >= m17gpu_device_count()) return fail(M17GPU_ERR_ARG, "m17gpu_create: no such device");
    DeviceScope dev_scope_(device);                 // the caller's current device is left as it was
    if (!dev_scope_.ok) return fail(M17GPU_ERR_HIP, "m17gpu_create: cannot select the device");
    m17gpu_ctx *ctx = new (std::nothrow) m17gpu_ctx;
    if (!ctx) return fail(M17GPU_ERR_NOMEM, "m17gpu_create: out of host memory");
    ctx->device = device; ctx->C = n_channels; ctx->max_blocks = max_blocks;
    ctx->rec_cap_max = 2 * max_blocks + 2;
    const size_t cb = (size_t)n_channels * max_blocks;
    int rc = upload_tables(ctx);
    if (rc != M17GPU_OK) { m17gpu_destroy(ctx); return rc; }
#define ALLOC(ptr, bytes)                                                               \
    do {                                                                               \
        hipError_t e_ = hipMalloc(reinterpret_cast<void **>(&(ptr)), (bytes));          \
        if (e_ != hipSuccess) { m17gpu_destroy(ctx);                                     \
            return fail(M17GPU_ERR_NOMEM, std::string("hipMalloc " #ptr ": ") + hipGetErrorString(e_)); } \
    } while (0)
    ALLOC(ctx->d_state, sizeof(ChanState) * (size_t)n_channels);
    ALLOC(ctx->d_disc, sizeof(float) * cb * kDiscOut);
    ALLOC(ctx->d_offs, sizeof(float) * cb);
    ALLOC(ctx->d_fsym, sizeof(float) * (size_t)n_channels * ctx->rec_cap_max * kSlotFloats);
    ALLOC(ctx->d_work, sizeof(int32_t) * 3 * (size_t)n_channels * ctx->rec_cap_max);   // one list per frame type (decode_impl 2)
    ALLOC(ctx->d_nwork, sizeof(int32_t) * 8);
    ALLOC(ctx->d_counts, sizeof(int32_t) * (size_t)n_channels);
    ALLOC(ctx->d_dec_hist, sizeof(uint32_t) * 32 * (size_t)n_channels);
#undef ALLOC
    rc = m17gpu_reset(ctx, nullptr);
    if (rc != M17GPU_OK) { m17gpu_destroy(ctx); return rc; }
    {
        const hipError_t e_ = hipDeviceSynchronize();
        if (e_ != hipSuccess) {
            m17gpu_destroy(ctx);
            return fail(M17GPU_ERR_HIP, std::string("m17gpu_create: ") + hipGetErrorString(e_));
        }
    }
    *out = ctx;
    return M17GPU_OK;
}

void m17gpu_destroy(m17gpu_ctx *ctx)
{
    if (!ctx) return;
    DeviceScope dev_scope_(ctx->device);
    void *bufs[] = {ctx->d_state, ctx->d_disc, ctx->d_offs, ctx->d_fsym, ctx->d_work,
                    ctx->d_nwork, ctx->d_counts, ctx->d_genc, ctx->d_gerr, ctx->d_crc_basis, ctx->d_dec_hist, ctx->d_flags};
    for (void *p : bufs) (void)hipFree(p);
    for (hipEvent_t e : ctx->ev_pool) (void)hipEventDestroy(e);
    delete ctx;
}

int m17gpu_reset(m17gpu_ctx *ctx, void *stream)
{
    if (!ctx) return fail(M17GPU_ERR_ARG, "m17gpu_reset: null context");
    ON_CTX_DEVICE(ctx);
    const long long words = (long long)ctx->C * (long long)(sizeof(ChanState) / 4);
    hipLaunchKernelGGL(k_reset, dim3(cdiv(words, 256)), dim3(256), 0, S(stream), ctx->d_state, ctx->C);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemsetAsync(ctx->d_dec_hist, 0, sizeof(uint32_t) * 32 * (size_t)ctx->C, S(stream)));
    HIPCHK(hipMemsetAsync(ctx->d_nwork, 0, sizeof(int32_t) * 8, S(stream)));
    ctx->nwork_dirty = false;
    return M17GPU_OK;
}

int m17gpu_rx_blocks(m17gpu_ctx *ctx, const int16_t *d_iq, int nblk, int mode,
                     m17gpu_rec *d_recs, int rec_cap, int32_t *d_counts,
                     float *d_syms, int32_t *d_nsyms, void *stream)
{
    if (!ctx || !d_iq || nblk <= 0 || nblk > ctx->max_blocks || rec_cap < 0)
        return fail(M17GPU_ERR_ARG, "m17gpu_rx_blocks: bad argument (nblk must be 1..max_blocks)");
    if (mode != 0 && mode != 1) return fail(M17GPU_ERR_ARG, "m17gpu_rx_blocks: mode must be 0 (front end) or 1 (full chain)");
    ON_CTX_DEVICE(ctx);
    // mode 1: every framer event must get its record -- an event without one would also lose its frame symbols,
    // and with them the LICH / counter / packet bookkeeping the reference does for that frame
    if (mode == 1 && (!d_recs || rec_cap < 2 * nblk + 2 || rec_cap > ctx->rec_cap_max))
        return fail(M17GPU_ERR_ARG, "m17gpu_rx_blocks: mode 1 needs d_recs and 2*nblk+2 <= rec_cap <= 2*max_blocks+2");
    // the network sink is indexed [channel][record] with the call's rec_cap: a sink of another capacity would be
    // written out of bounds (smaller) or at the wrong rows (larger)
    if (mode == 1 && ctx->d_net && rec_cap != ctx->net_rec_cap)
        return fail(M17GPU_ERR_ARG, "m17gpu_rx_blocks: rec_cap differs from the capacity the network sink was attached with");
    hipStream_t st = S(stream);
    int rc;
    const bool full = mode == 1;
    // The work-list counters are zero here: m17gpu_reset zeroed them, and every full-chain call leaves them zeroed (the
    // bookkeeping kernel, the last one of the call).  A call that returned between k_worklist and that kernel's launch
    // did not: the next one zeroes them itself.  (All full-chain calls of a context are ordered on one stream: header.)
    if (full) {
        if (ctx->nwork_dirty) HIPCHK(hipMemsetAsync(ctx->d_nwork, 0, sizeof(int32_t) * 8, st));
        ctx->nwork_dirty = false;
    }
    const int fir = fir_choice(ctx, nblk);
    ctx->last_path[0] = fir; ctx->last_path[1] = plain_slots(ctx, nblk) ? 1 : 0; ctx->last_path[2] = 0;
    ctx->last_path[3] = 0;
    hipEvent_t *ev = nullptr;
    if (ctx->profiling && ctx->ev_mode.size() < 512) {
        const size_t base = ctx->ev_pool.size();
        for (int i = 0; i < 7; ++i) {
            hipEvent_t e;
            HIPCHK(hipEventCreate(&e));
            ctx->ev_pool.push_back(e);
        }
        ctx->ev_mode.push_back(mode);
        ev = &ctx->ev_pool[base];
    }
    if (ev) HIPCHK(hipEventRecord(ev[5], st));
    {
#define MARK(i) do { if (ev) HIPCHK(hipEventRecord(ev[i], st)); } while (0)
        MARK(0);
        if (ctx->afc) {
            // AFC closes a per-block loop through front end, timing loop and framer (k_frontend_afc): block by block
            for (int b = 0; b < nblk; ++b) {
                hipLaunchKernelGGL(k_frontend_afc, dim3(ctx->C), dim3(64), 0, st, reinterpret_cast<const uint32_t *>(d_iq),
                                   ctx->d_state, ctx->d_disc, ctx->d_offs, nblk, b);
                HIPCHK(hipGetLastError());
                if ((rc = launch_sync_frame(ctx, ctx->d_disc, ctx->d_offs, nblk, mode, d_recs, rec_cap, d_counts,
                                            d_syms, d_nsyms, st, -1, b, 1)) != M17GPU_OK) return rc;
            }
            MARK(1);
            MARK(2);
        } else if (fir == 5) {
            MARK(1);                             // no separate front end: a third wave per channel, under the timing loop
            if ((rc = launch_sync_frame(ctx, ctx->d_disc, ctx->d_offs, nblk, mode, d_recs, rec_cap, d_counts,
                                        d_syms, d_nsyms, st, -1, 0, -1, 0, -1, d_iq)) != M17GPU_OK) return rc;
            MARK(2);
        } else if (fir == 4) {
            MARK(1);                             // no separate front end: stage 0 reads as zero, stage 1 is the wave-per-channel kernel
            if ((rc = launch_fused(ctx, d_iq, nblk, mode, d_recs, rec_cap, d_counts, d_syms, d_nsyms, st)) != M17GPU_OK) return rc;
            MARK(2);
        } else {
            if ((rc = launch_frontend(ctx, d_iq, nblk, ctx->d_disc, ctx->d_offs, 1, st)) != M17GPU_OK) return rc;
            MARK(1);
            if ((rc = launch_sync_frame(ctx, ctx->d_disc, ctx->d_offs, nblk, mode, d_recs, rec_cap, d_counts,
                                        d_syms, d_nsyms, st)) != M17GPU_OK) return rc;
            MARK(2);
        }
        if (full) {
            if ((rc = launch_decode(ctx, d_recs, rec_cap, d_counts, st, 0, ctx->C, 0, ev ? ev[3] : nullptr, plain_slots(ctx, nblk))) != M17GPU_OK) return rc;
            ctx->nwork_dirty = false;
            MARK(4);
        }
#undef MARK
    }
    if (ev) HIPCHK(hipEventRecord(ev[6], st));
    return M17GPU_OK;
}

#ifdef M17_STAMPS
// EXPERIMENT (not in the header): k_pc_mock on the context's discriminator stream -- call m17gpu_rx_blocks(mode 0) on the
// same d_iq first, so that the producers rewrite what is already there.  One launch = the front end's and the timing
// kernel's work of one step, co-resident.
int m17gpu_debug_pc_mock(m17gpu_ctx *ctx, const int16_t *d_iq, int nblk, float *d_syms, int32_t *d_nsyms, void *stream)
{
    if (!ctx || !d_iq || nblk <= 0 || nblk > ctx->max_blocks) return fail(M17GPU_ERR_ARG, "m17gpu_debug_pc_mock: bad argument");
    ON_CTX_DEVICE(ctx);
    hipLaunchKernelGGL(k_pc_mock, dim3(cdiv(ctx->C, PC_CONS)), dim3(64 * (PC_CONS + PC_PROD)), 0, S(stream),
                       reinterpret_cast<const uint4 *>(d_iq), ctx->d_disc, ctx->d_offs, ctx->d_disc, ctx->d_offs,
                       ctx->d_state, ctx->d_state, ctx->C, nblk, d_syms, d_nsyms, ctx->d_fsym, ctx->d_counts);
    HIPCHK(hipGetLastError());
    return M17GPU_OK;
}

int m17gpu_debug_ptrs(m17gpu_ctx *ctx, unsigned long long *out /* [6] */)
{
    out[0] = (unsigned long long)ctx->d_state; out[1] = (unsigned long long)ctx->d_disc; out[2] = (unsigned long long)ctx->d_offs;
    out[3] = (unsigned long long)ctx->d_fsym; out[4] = (unsigned long long)ctx->d_counts; out[5] = (unsigned long long)ctx->d_work;
    return 0;
}
int m17gpu_debug_rc_stamps(unsigned long long *out /* [16384][4] */)
{
    HIPCHK(hipDeviceSynchronize());
    HIPCHK(hipMemcpyFromSymbol(out, HIP_SYMBOL(g_rc_stamps), sizeof(unsigned long long) * 16384 * 4));
    return 0;
}
int m17gpu_debug_stamps(unsigned long long *out)
{
    HIPCHK(hipDeviceSynchronize());
    HIPCHK(hipMemcpyFromSymbol(out, HIP_SYMBOL(g_stamps), sizeof(unsigned long long) * 16));
    return 0;
}
int m17gpu_debug_fe_stamps(unsigned long long *out /* [8] */)
{
    HIPCHK(hipDeviceSynchronize());
    HIPCHK(hipMemcpyFromSymbol(out, HIP_SYMBOL(g_fe_stamps), sizeof(unsigned long long) * 8));
    return 0;
}
int m17gpu_debug_fe_span(unsigned long long *out /* [16384][2] */)
{
    HIPCHK(hipDeviceSynchronize());
    HIPCHK(hipMemcpyFromSymbol(out, HIP_SYMBOL(g_fe_span), sizeof(unsigned long long) * 16384 * 2));
    return 0;
}
int m17gpu_debug_wave_span(unsigned long long *out /* [16384][3] */)
{
    HIPCHK(hipDeviceSynchronize());
    HIPCHK(hipMemcpyFromSymbol(out, HIP_SYMBOL(g_wave_span), sizeof(unsigned long long) * 16384 * 3));
    return 0;
}
int m17gpu_debug_chan_stamps_x(unsigned long long *out /* [4096][2] */)
{
    HIPCHK(hipDeviceSynchronize());
    HIPCHK(hipMemcpyFromSymbol(out, HIP_SYMBOL(g_chan_stamps_x), sizeof(unsigned long long) * 4096 * 2));
    return 0;
}
int m17gpu_debug_chan_stamps(unsigned long long *out /* [4096][8] */)
{
    HIPCHK(hipDeviceSynchronize());
    HIPCHK(hipMemcpyFromSymbol(out, HIP_SYMBOL(g_chan_stamps), sizeof(unsigned long long) * 4096 * 8));
    return 0;
}
#endif

// Exhaustive on-device check that the shortened exact-arithmetic sequences of the
// front end (int16 scaling, sqrt, reciprocal, composed limiter) return the same
// bits as the literal reference expressions.  h_bad[4] = mismatch counts.
int m17gpu_selftest(m17gpu_ctx *ctx, unsigned *h_bad)
{
    if (!ctx || !h_bad) return fail(M17GPU_ERR_ARG, "m17gpu_selftest: bad argument");
    ON_CTX_DEVICE(ctx);
    unsigned *d_bad = nullptr;
    HIPCHK(hipMalloc(&d_bad, 8 * sizeof(unsigned)));
    HIPCHK(hipMemset(d_bad, 0, 8 * sizeof(unsigned)));
    hipLaunchKernelGGL(k_selftest_scale, dim3(256), dim3(256), 0, nullptr, d_bad + 0);
    // a = re^2 + im^2 lies in [9e-10, 2]; sweep every float in [2^-32, 8)
    hipLaunchKernelGGL(k_selftest_sqrt, dim3(4096), dim3(256), 0, nullptr, 0x2F800000u, 0x41000000u, d_bad + 1);
    // m = sqrt(a) lies in [3e-5, 1.42]; sweep every float in [2^-17, 4)
    hipLaunchKernelGGL(k_selftest_rcp, dim3(4096), dim3(256), 0, nullptr, 0x37000000u, 0x40800000u, d_bad + 2);
    hipLaunchKernelGGL(k_selftest_limit, dim3(8192), dim3(256), 0, nullptr, d_bad + 3);
    HIPCHK(hipGetLastError());
    HIPCHK(hipDeviceSynchronize());
    HIPCHK(hipMemcpy(h_bad, d_bad, 4 * sizeof(unsigned), hipMemcpyDeviceToHost));

    (void)hipFree(d_bad);
    return M17GPU_OK;
}

// Implementation selectors, for A/B measurements and so that every kernel variant stays under the
// parity tests.  Every accepted value selects a kernel that is held to bit-exact parity; anything
// else is rejected.
int m17gpu_set_option(m17gpu_ctx *ctx, const char *name, int value)
{
    if (!ctx || !name) return fail(M17GPU_ERR_ARG, "m17gpu_set_option: bad argument");
    auto bad = [&]() { return fail(M17GPU_ERR_ARG, std::string("m17gpu_set_option: value out of range for ") + name); };
    if (!std::strcmp(name, "sync_impl")) { if (value != 0 && value != 6 && value != 8) return bad(); ctx->sync_impl = value; }
    else if (!std::strcmp(name, "fe_impl")) { if (value != 0 && (value < 2 || value > 4)) return bad(); ctx->fe_impl = value; }
    else if (!std::strcmp(name, "fir_impl")) { if (value != 0 && value != 1 && value != 4 && value != 5) return bad(); ctx->fir_impl = value; }
    else if (!std::strcmp(name, "afc")) { if (value != 0 && value != 1) return bad(); ctx->afc = value; }
    else if (!std::strcmp(name, "slot_impl")) { if (value < 0 || value > 2) return bad(); ctx->slot_impl = value; }
    else if (!std::strcmp(name, "book_impl")) { if (value < 0 || value > 2) return bad(); ctx->book_impl = value; }
#ifdef M17_STAMPS
    else if (!std::strcmp(name, "fe_debug")) { ctx->fe_debug = value; }      // instrumented build only: WRONG results
#endif
    else return fail(M17GPU_ERR_ARG, std::string("m17gpu_set_option: unknown option ") + name);
    return M17GPU_OK;
}

static_assert(sizeof(m17gpu_lsf_fields) == sizeof(LsfFieldsDev) && sizeof(m17gpu_lsf_fields) == 64 &&
              offsetof(m17gpu_lsf_fields, crc_ok) == offsetof(LsfFieldsDev, crc_ok) &&
              offsetof(m17gpu_lsf_fields, meta) == offsetof(LsfFieldsDev, meta), "LSF field layouts must agree");

// Network sink of the full chain (SURVEY 8f-3 on the device): while set, every m17gpu_rx_blocks(mode 1) call writes the
// 54-byte M17-over-IP frame of each DELIVERED record into d_net[channel][record index] (rows of 56 bytes).
int m17gpu_set_net_output(m17gpu_ctx *ctx, uint8_t *d_net, int net_rec_cap, const uint16_t *d_stream_ids, uint64_t dst_override)
{
    if (!ctx) return fail(M17GPU_ERR_ARG, "m17gpu_set_net_output: null context");
    if (dst_override >> 48) return fail(M17GPU_ERR_ARG, "m17gpu_set_net_output: dst_override is a 48-bit callsign");
    if (d_net && (net_rec_cap < 4 || net_rec_cap > ctx->rec_cap_max))
        return fail(M17GPU_ERR_ARG, "m17gpu_set_net_output: net_rec_cap must be a mode-1 rec_cap of this context (4 .. 2*max_blocks+2)");
    ctx->d_net = d_net; ctx->net_rec_cap = d_net ? net_rec_cap : 0;
    ctx->d_stream_ids = d_net ? d_stream_ids : nullptr; ctx->dst_override = dst_override;
    return M17GPU_OK;
}

int m17gpu_parse_lsf_batch(m17gpu_ctx *ctx, const uint8_t *d_lsf, m17gpu_lsf_fields *d_out, int n, void *stream)
{
    if (!ctx || !d_lsf || !d_out || n <= 0) return fail(M17GPU_ERR_ARG, "m17gpu_parse_lsf_batch: bad argument");
    ON_CTX_DEVICE(ctx);
    hipLaunchKernelGGL(k_parse_lsf, dim3(cdiv(n, 128)), dim3(128), 0, S(stream), d_lsf, reinterpret_cast<LsfFieldsDev *>(d_out), n);
    HIPCHK(hipGetLastError());
    return M17GPU_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// Multi-GPU fan-out for a C / C++ host: the path shards by channel with no data-path collective; the only
// exchanges are the IQ fan-out from an ingest rank and the gather of the 64-byte records, both point to point.  RCCL
// is taken from the process (the copy the host application or PyTorch already loaded, else librccl.so.1) at first
// use: the library itself does not link it, so single-GPU users need no RCCL at all.
// ---------------------------------------------------------------------------------------------------------------
} // extern "C" (the helper below has C++ linkage)
namespace {
struct RcclApi {
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    ncclResult_t (*Send)(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    bool ok = false;
};
std::mutex g_rccl_mu;
std::string g_rccl_path;            // m17gpu_shard_set_library: bind this library instead of the process's RCCL
bool g_rccl_bound = false;
const RcclApi &rccl()
{
    static RcclApi api;
    static std::once_flag once;
    std::call_once(once, [] {
        void *h = nullptr;
        {
            std::lock_guard<std::mutex> lk(g_rccl_mu);
            g_rccl_bound = true;
            if (!g_rccl_path.empty()) h = dlopen(g_rccl_path.c_str(), RTLD_NOW | RTLD_LOCAL);
            else {
                for (const char *name : {"librccl.so", "librccl.so.1"})
                    if (!h) h = dlopen(name, RTLD_NOW | RTLD_NOLOAD);             // whichever RCCL the process already runs on
                if (!h) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
                if (!h) h = dlopen("/opt/rocm/lib/librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
            }
        }
        if (!h) return;
        api.GroupStart = reinterpret_cast<decltype(api.GroupStart)>(dlsym(h, "ncclGroupStart"));
        api.GroupEnd = reinterpret_cast<decltype(api.GroupEnd)>(dlsym(h, "ncclGroupEnd"));
        api.Send = reinterpret_cast<decltype(api.Send)>(dlsym(h, "ncclSend"));
        api.Recv = reinterpret_cast<decltype(api.Recv)>(dlsym(h, "ncclRecv"));
        api.GetErrorString = reinterpret_cast<decltype(api.GetErrorString)>(dlsym(h, "ncclGetErrorString"));
        api.ok = api.GroupStart && api.GroupEnd && api.Send && api.Recv;
    });
    return api;
}
#define RCCLCHK(expr)                                                                                    \
    do {                                                                                                 \
        ncclResult_t r_ = (expr);                                                                        \
        if (r_ != ncclSuccess)                                                                           \
            return fail(M17GPU_ERR_HIP, std::string(#expr) + ": " + (R.GetErrorString ? R.GetErrorString(r_) : "RCCL error")); \
    } while (0)
// inside ncclGroupStart .. ncclGroupEnd: remember the first failure and keep going to the GroupEnd -- returning with the
// group open would nest every later RCCL call of this thread inside it
#define RCCLGRP(expr)                                                                                    \
    do {                                                                                                 \
        ncclResult_t r_ = (expr);                                                                        \
        if (r_ != ncclSuccess && grp_err.empty())                                                        \
            grp_err = std::string(#expr) + ": " + (R.GetErrorString ? R.GetErrorString(r_) : "RCCL error"); \
    } while (0)
} // namespace
extern "C" {

void m17gpu_shard_range(int rank, int world, int n_channels, int *lo, int *hi)
{
    const int base = n_channels / world, extra = n_channels % world;
    const int a = rank * base + (rank < extra ? rank : extra);
    if (lo) *lo = a;
    if (hi) *hi = a + base + (rank < extra ? 1 : 0);
}

int m17gpu_shard_scatter_iq(m17gpu_ctx *ctx, void *comm, int rank, int world, int src_rank,
                            const int16_t *d_iq_all, int n_channels_total, int nblk, int16_t *d_iq_mine, void *stream)
{
    if (!ctx || !comm || world <= 0 || rank < 0 || rank >= world || src_rank < 0 || src_rank >= world || nblk <= 0 ||
        n_channels_total <= 0 || !d_iq_mine || (rank == src_rank && !d_iq_all))
        return fail(M17GPU_ERR_ARG, "m17gpu_shard_scatter_iq: bad argument");
    int lo, hi;
    m17gpu_shard_range(rank, world, n_channels_total, &lo, &hi);
    if (hi - lo != ctx->C) return fail(M17GPU_ERR_ARG, "m17gpu_shard_scatter_iq: the context does not hold this rank's channel range");
    const RcclApi &R = rccl();
    if (!R.ok) return fail(M17GPU_ERR_HIP, "m17gpu_shard_scatter_iq: no RCCL library in this process (librccl.so.1)");
    ON_CTX_DEVICE(ctx);
    const size_t per = (size_t)nblk * kBlockSamples * 2 * sizeof(int16_t);       // bytes per channel
    hipStream_t st = S(stream);
    // one group: the sends to all peers are in flight together, each on its own xGMI link (a ring would be per-link bound)
    std::string grp_err;
    RCCLCHK(R.GroupStart());
    if (rank == src_rank) {
        for (int r = 0; r < world; ++r) {
            int a, b;
            m17gpu_shard_range(r, world, n_channels_total, &a, &b);
            if (r == src_rank || b <= a) continue;
            RCCLGRP(R.Send(reinterpret_cast<const char *>(d_iq_all) + (size_t)a * per, (size_t)(b - a) * per, ncclChar, r,
                           static_cast<ncclComm_t>(comm), st));
        }
    } else if (hi > lo) {
        RCCLGRP(R.Recv(d_iq_mine, (size_t)(hi - lo) * per, ncclChar, src_rank, static_cast<ncclComm_t>(comm), st));
    }
    RCCLGRP(R.GroupEnd());
    if (!grp_err.empty()) return fail(M17GPU_ERR_HIP, grp_err);
    if (rank == src_rank && hi > lo)
        HIPCHK(hipMemcpyAsync(d_iq_mine, reinterpret_cast<const char *>(d_iq_all) + (size_t)lo * per, (size_t)(hi - lo) * per,
                              hipMemcpyDeviceToDevice, st));
    return M17GPU_OK;
}

int m17gpu_shard_gather_records(m17gpu_ctx *ctx, void *comm, int rank, int world, int dst_rank,
                                const m17gpu_rec *d_recs_mine, const int32_t *d_counts_mine, int rec_cap,
                                int n_channels_total, m17gpu_rec *d_recs_all, int32_t *d_counts_all, void *stream)
{
    if (!ctx || !comm || world <= 0 || rank < 0 || rank >= world || dst_rank < 0 || dst_rank >= world || rec_cap <= 0 ||
        n_channels_total <= 0 || !d_recs_mine || !d_counts_mine || (rank == dst_rank && (!d_recs_all || !d_counts_all)))
        return fail(M17GPU_ERR_ARG, "m17gpu_shard_gather_records: bad argument");
    int lo, hi;
    m17gpu_shard_range(rank, world, n_channels_total, &lo, &hi);
    if (hi - lo != ctx->C) return fail(M17GPU_ERR_ARG, "m17gpu_shard_gather_records: the context does not hold this rank's channel range");
    const RcclApi &R = rccl();
    if (!R.ok) return fail(M17GPU_ERR_HIP, "m17gpu_shard_gather_records: no RCCL library in this process (librccl.so.1)");
    ON_CTX_DEVICE(ctx);
    const size_t per = (size_t)rec_cap * sizeof(m17gpu_rec);
    hipStream_t st = S(stream);
    std::string grp_err;
    RCCLCHK(R.GroupStart());
    if (rank == dst_rank) {
        for (int r = 0; r < world; ++r) {
            int a, b;
            m17gpu_shard_range(r, world, n_channels_total, &a, &b);
            if (r == dst_rank || b <= a) continue;
            RCCLGRP(R.Recv(reinterpret_cast<char *>(d_recs_all) + (size_t)a * per, (size_t)(b - a) * per, ncclChar, r,
                           static_cast<ncclComm_t>(comm), st));
            RCCLGRP(R.Recv(d_counts_all + a, (size_t)(b - a), ncclInt32, r, static_cast<ncclComm_t>(comm), st));
        }
    } else if (hi > lo) {
        RCCLGRP(R.Send(d_recs_mine, (size_t)(hi - lo) * per, ncclChar, dst_rank, static_cast<ncclComm_t>(comm), st));
        RCCLGRP(R.Send(d_counts_mine, (size_t)(hi - lo), ncclInt32, dst_rank, static_cast<ncclComm_t>(comm), st));
    }
    RCCLGRP(R.GroupEnd());
    if (!grp_err.empty()) return fail(M17GPU_ERR_HIP, grp_err);
    if (rank == dst_rank && hi > lo) {
        HIPCHK(hipMemcpyAsync(reinterpret_cast<char *>(d_recs_all) + (size_t)lo * per, d_recs_mine, (size_t)(hi - lo) * per,
                              hipMemcpyDeviceToDevice, st));
        HIPCHK(hipMemcpyAsync(d_counts_all + lo, d_counts_mine, (size_t)(hi - lo) * sizeof(int32_t), hipMemcpyDeviceToDevice, st));
    }
    return M17GPU_OK;
}

// Valid records only, channel-major (m17_pack.hip): d_offsets [C+1] (d_offsets[C] = number of records),
// d_packed [packed_cap] records.  Everything is enqueued on `stream`; nothing is read back.
int m17gpu_pack_records(m17gpu_ctx *ctx, const m17gpu_rec *d_recs, int rec_cap, const int32_t *d_counts,
                        m17gpu_rec *d_packed, int packed_cap, int32_t *d_offsets, void *stream)
{
    if (!ctx || !d_recs || !d_counts || !d_packed || !d_offsets || rec_cap <= 0 || packed_cap <= 0)
        return fail(M17GPU_ERR_ARG, "m17gpu_pack_records: bad argument");
    ON_CTX_DEVICE(ctx);
    hipStream_t st = S(stream);
    hipLaunchKernelGGL(k_pack_scan, dim3(1), dim3(1024), 0, st, d_counts, ctx->C, rec_cap, d_offsets);
    const long long pieces = (long long)ctx->C * rec_cap * 4;
    hipLaunchKernelGGL(k_pack_copy, dim3(cdiv(pieces, 256)), dim3(256), 0, st, reinterpret_cast<const uint4 *>(d_recs), ctx->C,
                       rec_cap, d_offsets, reinterpret_cast<uint4 *>(d_packed), packed_cap);
    HIPCHK(hipGetLastError());
    return M17GPU_OK;
}

int m17gpu_unpack_records(m17gpu_ctx *ctx, const m17gpu_rec *d_packed, const int32_t *d_offsets, int n_channels,
                          m17gpu_rec *d_recs, int rec_cap, int32_t *d_counts, void *stream)
{
    if (!ctx || !d_packed || !d_offsets || !d_recs || !d_counts || rec_cap <= 0 || n_channels <= 0)
        return fail(M17GPU_ERR_ARG, "m17gpu_unpack_records: bad argument");
    ON_CTX_DEVICE(ctx);
    const long long pieces = (long long)n_channels * rec_cap * 4;
    hipLaunchKernelGGL(k_unpack_copy, dim3(cdiv(pieces, 256)), dim3(256), 0, S(stream), reinterpret_cast<const uint4 *>(d_packed),
                       n_channels, rec_cap, d_offsets, reinterpret_cast<uint4 *>(d_recs), d_counts);
    HIPCHK(hipGetLastError());
    return M17GPU_OK;
}

// The gather of a step's records in packed form: every rank has run m17gpu_pack_records (d_packed_mine, d_offsets_mine
// [C+1]); dst_rank receives d_packed_all (all ranks' records, rank after rank, channel-major) and d_offsets_all
// [n_channels_total + 1] (global: channel c's records are rows d_offsets_all[c] .. d_offsets_all[c+1]).  What crosses the
// links is sum(counts) x 64 B + 4 B per channel -- about 1 MB per 16,384 channels x 12 blocks, where the unpacked
// [C][2 nblk + 2] array is 27 MB.  The row counts size the transfers, so this entry synchronises `stream` twice (each
// rank reads back its own total; the gathering rank the others' totals): it runs after the step, beside nothing.
// h_totals [world] (host, may be NULL) receives the per-rank record counts on dst_rank.
int m17gpu_shard_gather_packed(m17gpu_ctx *ctx, void *comm, int rank, int world, int dst_rank,
                               const m17gpu_rec *d_packed_mine, int packed_cap_mine, const int32_t *d_offsets_mine, int n_channels_total,
                               m17gpu_rec *d_packed_all, int packed_cap_all, int32_t *d_offsets_all, int32_t *h_totals, void *stream)
{
    // Only what makes the exchange itself impossible returns here, on this rank alone (a caller's programming error: its
    // peers then wait).  Everything else that is wrong on ONE rank -- no or too small a buffer, a context of another
    // channel count -- becomes that rank's "no" in the verdict every rank takes part in, so that ALL ranks return
    // M17GPU_ERR_ARG together and nothing is left unmatched.  Every rank of the communicator must make the call, also
    // one whose channel range is empty (world > channels: it passes any context, no buffers and capacity 0).
    if (!ctx || !comm || world <= 0 || rank < 0 || rank >= world || dst_rank < 0 || dst_rank >= world || n_channels_total <= 0)
        return fail(M17GPU_ERR_ARG, "m17gpu_shard_gather_packed: bad argument");
    int lo, hi;
    m17gpu_shard_range(rank, world, n_channels_total, &lo, &hi);
    const bool empty = hi == lo;
    std::string why_local;
    if (!empty && hi - lo != ctx->C) why_local = "the context does not hold this rank's channel range";
    else if (!empty && (!d_packed_mine || packed_cap_mine <= 0 || !d_offsets_mine)) why_local = "no or an empty packed buffer on a rank that has channels";
    else if (rank == dst_rank && (!d_packed_all || packed_cap_all <= 0)) why_local = "the gathering rank has no destination buffer";
    const RcclApi &R = rccl();
    if (!R.ok) return fail(M17GPU_ERR_HIP, "m17gpu_shard_gather_packed: no RCCL library in this process (librccl.so.1)");
    ON_CTX_DEVICE(ctx);
    hipStream_t st = S(stream);
    ncclComm_t cm = static_cast<ncclComm_t>(comm);
    std::string grp_err;
    // Whether the exchange can go through is decided by ALL ranks before any record moves: a rank that left alone
    // between the two exchanges -- its own capacity too small, or the gathering rank's -- would leave its peers with a
    // send nobody receives (round 4's form did, on the gathering rank).  So: every rank reads its own total and checks it
    // against its own buffer; the verdicts travel to the gathering rank with the offset tables; its verdict over
    // everything travels back; the records move only if that is "go", and otherwise every rank returns M17GPU_ERR_ARG.
    // flags[0] = this rank's verdict (out) / the gathering rank's verdict (in), flags[1 ..] = the peers' verdicts on dst_rank
    if (!ctx->d_flags || ctx->n_flags < world + 1) {
        (void)hipFree(ctx->d_flags);
        ctx->d_flags = nullptr; ctx->n_flags = 0;
        HIPCHK(hipMalloc(&ctx->d_flags, sizeof(int32_t) * (size_t)(world + 1)));
        ctx->n_flags = world + 1;
    }
    // the gathering rank receives the peers' offset tables even when it will refuse: into a scratch table if it was given none
    int32_t *offs_scratch = nullptr;
    struct Scratch { int32_t *&p; ~Scratch() { (void)hipFree(p); } } scratch_{offs_scratch};
    if (rank == dst_rank && !d_offsets_all) {
        HIPCHK(hipMalloc(&offs_scratch, sizeof(int32_t) * ((size_t)n_channels_total + 1)));
        d_offsets_all = offs_scratch;
        if (why_local.empty()) why_local = "the gathering rank has no offset table";
    }
    const bool can_read = !empty && d_offsets_mine && hi - lo == ctx->C;
    // ... and a peer that has channels but no table to send ("no" for that very reason) sends zeros of the right length: the
    // gathering rank's receive is posted by channel range
    int32_t *offs_dummy = nullptr;
    Scratch dummy_{offs_dummy};
    if (!empty && !can_read && rank != dst_rank) {
        HIPCHK(hipMalloc(&offs_dummy, sizeof(int32_t) * (size_t)(hi - lo + 1)));
        HIPCHK(hipMemsetAsync(offs_dummy, 0, sizeof(int32_t) * (size_t)(hi - lo + 1), st));
    }
    const int32_t *offs_src = can_read ? d_offsets_mine : offs_dummy;
    int32_t mine_total = 0;
    if (can_read) HIPCHK(hipMemcpyAsync(&mine_total, d_offsets_mine + (hi - lo), sizeof(int32_t), hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    // m17gpu_pack_records drops rows beyond its packed_cap: a total above the capacity is this rank's "no" as well
    int32_t my_ok = (why_local.empty() && mine_total >= 0 && mine_total <= packed_cap_mine) ? 1 : 0;
    if (!my_ok) mine_total = 0;
    HIPCHK(hipMemcpyAsync(ctx->d_flags, &my_ok, sizeof(int32_t), hipMemcpyHostToDevice, st));
    // leg 1: verdicts and local offset tables ([Cr] int32 each) to the gathering rank
    RCCLCHK(R.GroupStart());
    if (rank == dst_rank) {
        for (int r = 0; r < world; ++r) {
            int a, b;
            m17gpu_shard_range(r, world, n_channels_total, &a, &b);
            if (r == dst_rank) continue;
            RCCLGRP(R.Recv(ctx->d_flags + 1 + r, 1, ncclInt32, r, cm, st));
            if (b > a) RCCLGRP(R.Recv(d_offsets_all + a + 1, (size_t)(b - a), ncclInt32, r, cm, st));    // local offs[1 .. Cr]; shifted below
        }
    } else {
        RCCLGRP(R.Send(ctx->d_flags, 1, ncclInt32, dst_rank, cm, st));
        // (a rank that says "no" for want of an offset table still sends one: the gathering rank's receive is posted by range)
        if (hi > lo) RCCLGRP(R.Send(offs_src + 1, (size_t)(hi - lo), ncclInt32, dst_rank, cm, st));
    }
    RCCLGRP(R.GroupEnd());
    if (!grp_err.empty()) return fail(M17GPU_ERR_HIP, grp_err);
    std::vector<int32_t> totals((size_t)world, 0), oks((size_t)world, 1);
    int32_t go = 1;
    std::string why;
    if (rank == dst_rank) {
        if (hi > lo && can_read) HIPCHK(hipMemcpyAsync(d_offsets_all + lo + 1, d_offsets_mine + 1, (size_t)(hi - lo) * sizeof(int32_t), hipMemcpyDeviceToDevice, st));
        for (int r = 0; r < world; ++r) {                       // each rank's total = its last local offset
            int a, b;
            m17gpu_shard_range(r, world, n_channels_total, &a, &b);
            if (b > a && (r != dst_rank || can_read)) HIPCHK(hipMemcpyAsync(&totals[r], d_offsets_all + b, sizeof(int32_t), hipMemcpyDeviceToHost, st));
            if (r != dst_rank) HIPCHK(hipMemcpyAsync(&oks[r], ctx->d_flags + 1 + r, sizeof(int32_t), hipMemcpyDeviceToHost, st));
        }
        HIPCHK(hipStreamSynchronize(st));
        oks[rank] = my_ok;
        long long sum = 0;
        for (int r = 0; r < world; ++r) {
            sum += totals[r];
            if (!oks[r] && go) { go = 0; why = "rank " + std::to_string(r) + (r == rank && !why_local.empty() ? ": " + why_local : " refused: its packed buffer is missing, of another context or too small for its records"); }
        }
        if (go && sum > packed_cap_all) { go = 0; why = "d_packed_all is too small for this step's records"; }
        HIPCHK(hipMemcpyAsync(ctx->d_flags, &go, sizeof(int32_t), hipMemcpyHostToDevice, st));
    }
    // leg 1.5: the verdict back to every peer
    RCCLCHK(R.GroupStart());
    if (rank == dst_rank) {
        for (int r = 0; r < world; ++r)
            if (r != dst_rank) RCCLGRP(R.Send(ctx->d_flags, 1, ncclInt32, r, cm, st));
    } else {
        RCCLGRP(R.Recv(ctx->d_flags, 1, ncclInt32, dst_rank, cm, st));
    }
    RCCLGRP(R.GroupEnd());
    if (!grp_err.empty()) return fail(M17GPU_ERR_HIP, grp_err);
    if (rank != dst_rank) {
        HIPCHK(hipMemcpyAsync(&go, ctx->d_flags, sizeof(int32_t), hipMemcpyDeviceToHost, st));
        totals[rank] = mine_total;
    }
    HIPCHK(hipStreamSynchronize(st));              // (dst_rank: the verdict's source word stays untouched until the sends have read it)
    if (!go)
        return fail(M17GPU_ERR_ARG, "m17gpu_shard_gather_packed: refused on every rank, nothing moved: " +
                                    (!why_local.empty() ? why_local : why.empty() ? std::string("a packed buffer is too small for this step's records (see the gathering rank)") : why));
    // leg 2: the records themselves, sum(counts) rows per rank
    long long base = 0;
    RCCLCHK(R.GroupStart());
    if (rank == dst_rank) {
        for (int r = 0; r < world; ++r) {
            if (r != dst_rank && totals[r] > 0)
                RCCLGRP(R.Recv(reinterpret_cast<char *>(d_packed_all) + (size_t)base * sizeof(m17gpu_rec), (size_t)totals[r] * sizeof(m17gpu_rec),
                               ncclChar, r, cm, st));
            base += totals[r];
        }
    } else if (mine_total > 0) {
        RCCLGRP(R.Send(d_packed_mine, (size_t)mine_total * sizeof(m17gpu_rec), ncclChar, dst_rank, cm, st));
    }
    RCCLGRP(R.GroupEnd());
    if (!grp_err.empty()) return fail(M17GPU_ERR_HIP, grp_err);
    if (rank == dst_rank) {
        // own rows, and every rank's local offsets moved behind the ranks before it
        long long before = 0;
        HIPCHK(hipMemsetAsync(d_offsets_all, 0, sizeof(int32_t), st));
        for (int r = 0; r < world; ++r) {
            int a, b;
            m17gpu_shard_range(r, world, n_channels_total, &a, &b);
            if (r == dst_rank && totals[r] > 0)
                HIPCHK(hipMemcpyAsync(reinterpret_cast<char *>(d_packed_all) + (size_t)before * sizeof(m17gpu_rec), d_packed_mine,
                                      (size_t)totals[r] * sizeof(m17gpu_rec), hipMemcpyDeviceToDevice, st));
            if (b > a && before > 0)
                hipLaunchKernelGGL(k_offs_shift, dim3(cdiv(b - a, 256)), dim3(256), 0, st, d_offsets_all + a, b - a, (int)before);
            before += totals[r];
        }
        HIPCHK(hipGetLastError());
        if (h_totals) std::memcpy(h_totals, totals.data(), sizeof(int32_t) * (size_t)world);
    }
    return M17GPU_OK;
}

// Bind the fan-out entries to THIS library (any library with RCCL's ncclGroupStart / ncclGroupEnd / ncclSend / ncclRecv:
// a host application's own RCCL build, a test transport) instead of the RCCL the process already runs on.  Before
// the first fan-out call only: once bound, the choice stands for the life of the process.
int m17gpu_shard_set_library(const char *path)
{
    std::lock_guard<std::mutex> lk(g_rccl_mu);
    if (g_rccl_bound) return fail(M17GPU_ERR_ARG, "m17gpu_shard_set_library: the fan-out entries are already bound to a library");
    g_rccl_path = path ? path : "";
    return M17GPU_OK;
}

int m17gpu_set_profiling(m17gpu_ctx *ctx, int on)
{
    if (!ctx) return fail(M17GPU_ERR_ARG, "m17gpu_set_profiling: null context");
    ctx->profiling = on != 0;
    return M17GPU_OK;
}

int m17gpu_get_kernel_ms(m17gpu_ctx *ctx, float h_ms[4], int *h_calls)
{
    return m17gpu_get_call_ms(ctx, h_ms, nullptr, h_calls);
}

// As m17gpu_get_kernel_ms, plus the average duration of a whole m17gpu_rx_blocks call between two events on the
// caller's stream (h_call_ms).  With channel chunks pipelined over internal streams the stage figures are those of
// chunk 0 on its stream, beside the other chunks' kernels; the call duration is the number that counts.
int m17gpu_get_call_ms(m17gpu_ctx *ctx, float h_ms[4], float *h_call_ms, int *h_calls)
{
    if (!ctx || !h_ms) return fail(M17GPU_ERR_ARG, "m17gpu_get_kernel_ms: bad argument");
    ON_CTX_DEVICE(ctx);
    double acc[4] = {0, 0, 0, 0}, call = 0;
    int n[4] = {0, 0, 0, 0};
    const size_t calls = ctx->ev_mode.size();
    for (size_t k = 0; k < calls; ++k) {
        hipEvent_t *ev = &ctx->ev_pool[7 * k];
        const int last = ctx->ev_mode[k] == 1 ? 4 : 2;
        HIPCHK(hipEventSynchronize(ev[6]));
        for (int i = 0; i < last; ++i) {
            float ms = 0;
            HIPCHK(hipEventElapsedTime(&ms, ev[i], ev[i + 1]));
            acc[i] += ms; n[i]++;
        }
        float ms = 0;
        HIPCHK(hipEventElapsedTime(&ms, ev[5], ev[6]));
        call += ms;
    }
    for (int i = 0; i < 4; ++i) h_ms[i] = n[i] ? (float)(acc[i] / n[i]) : 0.0f;
    if (h_call_ms) *h_call_ms = calls ? (float)(call / calls) : 0.0f;
    if (h_calls) *h_calls = (int)calls;
    for (hipEvent_t e : ctx->ev_pool) (void)hipEventDestroy(e);
    ctx->ev_pool.clear();
    ctx->ev_mode.clear();
    return M17GPU_OK;
}

int m17gpu_frontend(m17gpu_ctx *ctx, const int16_t *d_iq, int nblk, float *d_disc, float *d_offset, void *stream)
{
    if (!ctx || !d_iq || !d_disc || nblk <= 0 || nblk > ctx->max_blocks)
        return fail(M17GPU_ERR_ARG, "m17gpu_frontend: bad argument");
    ON_CTX_DEVICE(ctx);
    hipStream_t st = S(stream);
    float *offs = d_offset ? d_offset : ctx->d_offs;
    int rc = launch_frontend(ctx, d_iq, nblk, d_disc, offs, 1, st);
    if (rc != M17GPU_OK) return rc;
    const long long n = (long long)ctx->C * nblk * kDiscOut;
    hipLaunchKernelGGL(k_dc_remove, dim3(cdiv(n, 256)), dim3(256), 0, st, d_disc, offs, ctx->C * nblk);
    HIPCHK(hipGetLastError());
    return M17GPU_OK;
}

int m17gpu_sync_frame(m17gpu_ctx *ctx, const float *d_disc, int nblk, m17gpu_rec *d_recs, int rec_cap,
                      int32_t *d_counts, float *d_syms, int32_t *d_nsyms, void *stream)
{
    if (!ctx || !d_disc || nblk <= 0) return fail(M17GPU_ERR_ARG, "m17gpu_sync_frame: bad argument");
    ON_CTX_DEVICE(ctx);
    return launch_sync_frame(ctx, d_disc, nullptr, nblk, 0, d_recs, rec_cap, d_counts, d_syms, d_nsyms, S(stream));
}

int m17gpu_pluto_decimate(m17gpu_ctx *ctx, const int16_t *d_in, int n_in, int16_t *d_out, void *stream)
{
    if (!ctx || !d_in || !d_out || n_in < 32 || (n_in & 7))
        return fail(M17GPU_ERR_ARG, "m17gpu_pluto_decimate: n_in must be a multiple of 8 and >= 32");
    ON_CTX_DEVICE(ctx);
    hipStream_t st = S(stream);
    const int M = n_in / 8, wpc = (M + 60) / 61;
    hipLaunchKernelGGL(k_pluto_decimate, dim3(cdiv((long long)ctx->C * wpc, 4)), dim3(256), 0, st,
                       reinterpret_cast<const uint32_t *>(d_in), ctx->d_dec_hist,
                       reinterpret_cast<uint32_t *>(d_out), n_in, wpc);
    HIPCHK(hipGetLastError());
    hipLaunchKernelGGL(k_pluto_hist, dim3(cdiv((long long)ctx->C * 32, 256)), dim3(256), 0, st,
                       reinterpret_cast<const uint32_t *>(d_in), ctx->d_dec_hist, n_in, ctx->C);
    HIPCHK(hipGetLastError());
    return M17GPU_OK;
}

// GPU-side signal source: same arguments and the same signal as m17gen_batch
// (stream mode), written to device memory.  Channels are generated in groups so that the fp32
// phase workspace stays below 8 GiB.
int m17gpu_gen_batch(m17gpu_ctx *ctx, uint64_t base_seed, int first_channel, int nblk, int n_stream_frames,
                     float ebn0_db, float noise_cutoff_hz, int16_t *d_iq, uint8_t *d_lsf, uint8_t *d_payload,
                     int max_payload_frames, int32_t *d_nframes, void *stream)
{
    return m17gpu_gen_batch_stages(ctx, base_seed, first_channel, nblk, n_stream_frames, ebn0_db, noise_cutoff_hz, d_iq, d_lsf,
                                   d_payload, max_payload_frames, d_nframes, nullptr, nullptr, stream);
}

// The same with the generator's intermediate stages copied out, so that each can be compared with the oracle's
// transmitter on its own: d_dibits [C][nblk + 1][192] (0..3, 255 = unmodulated carrier) as m17_fmt_add_* built them,
// d_phase [C][nblk * 1920] the modulator's m_acc after every sample (before the per-symbol wrap; 0 during the start delay).
int m17gpu_gen_batch_stages(m17gpu_ctx *ctx, uint64_t base_seed, int first_channel, int nblk, int n_stream_frames,
                            float ebn0_db, float noise_cutoff_hz, int16_t *d_iq, uint8_t *d_lsf, uint8_t *d_payload,
                            int max_payload_frames, int32_t *d_nframes, uint8_t *d_dibits, float *d_phase, void *stream)
{
    if (!ctx || !d_iq || nblk <= 0 || n_stream_frames < 0 || (d_payload && max_payload_frames <= 0))
        return fail(M17GPU_ERR_ARG, "m17gpu_gen_batch: bad argument");
    ON_CTX_DEVICE(ctx);
    hipStream_t st = S(stream);
    const long long want = (long long)nblk * kBlockSamples;
    // modulator taps and deviations exactly as the host generator builds them (m17_modulate.cpp:9,73-74)
    float taps[310];
    m17::build_rrc(taps, 0.5f, 310, 10);
    m17::set_filter_gain(taps, 10, 1, 310);
    GenArgs A;
    A.base_seed = base_seed; A.nblk = nblk; A.n_stream_frames = n_stream_frames; A.nslots = nblk + 1;
    m17::tx_deviation_lut(A.lut);
    NoiseArgs NZ;
    std::memset(&NZ, 0, sizeof NZ);
    if (ebn0_db < 100.0f) {
        const double esn0 = 2.0 * std::pow(10.0, ebn0_db / 10.0), a = 16383.0;
        NZ.on = 1;
        NZ.sigma = std::sqrt(a * a * 10 / (2.0 * esn0));
        if (noise_cutoff_hz > 0.0f) {
            constexpr int L = 63;
            double hs = 0.0;
            const double fc = (double)noise_cutoff_hz / 48000.0;
            for (int k = 0; k < L; ++k) {
                const int m = k - L / 2;
                const double sinc = (m == 0) ? 2.0 * fc : std::sin(2.0 * M_PI * fc * m) / (M_PI * m);
                NZ.h[k] = sinc * (0.54 - 0.46 * std::cos(2.0 * M_PI * k / (L - 1)));
                hs += NZ.h[k];
            }
            for (int k = 0; k < L; ++k) NZ.h[k] /= hs;
            NZ.taps = L;
        }
    }
    int group = (int)std::max(1ll, (8ll << 30) / (want * 4));   // the phase chain is one lane per channel: few, large groups
    if (group > ctx->C) group = ctx->C;
    float *d_taps = nullptr, *d_sum = nullptr;
    uint8_t *d_sym = nullptr;
    struct Workspace {                              // freed on every return path
        float *&a, *&b; uint8_t *&c;
        ~Workspace() { (void)hipFree(a); (void)hipFree(b); (void)hipFree(c); }
    } ws_{d_taps, d_sum, d_sym};
    HIPCHK(hipMalloc(&d_taps, sizeof taps));
    HIPCHK(hipMemcpyAsync(d_taps, taps, sizeof taps, hipMemcpyHostToDevice, st));
    HIPCHK(hipMalloc(&d_sum, sizeof(float) * (size_t)group * want));
    HIPCHK(hipMalloc(&d_sym, (size_t)group * A.nslots * 192));
    if (d_nframes) HIPCHK(hipMemsetAsync(d_nframes, 0, sizeof(int32_t) * (size_t)ctx->C, st));
    for (int c0 = 0; c0 < ctx->C; c0 += group) {
        const int cn = std::min(group, ctx->C - c0);
        A.first_channel = first_channel + c0; A.C = cn;
        hipLaunchKernelGGL(k_gen_symbols, dim3(cdiv((long long)cn * A.nslots, 64)), dim3(64), 0, st, A, ctx->d_genc, d_sym,
                           d_lsf ? d_lsf + (size_t)c0 * 30 : nullptr,
                           d_payload ? d_payload + (size_t)c0 * max_payload_frames * 16 : nullptr, max_payload_frames,
                           d_nframes ? d_nframes + c0 : nullptr);
        if (d_dibits)
            HIPCHK(hipMemcpyAsync(d_dibits + (size_t)c0 * A.nslots * 192, d_sym, (size_t)cn * A.nslots * 192, hipMemcpyDeviceToDevice, st));
        hipLaunchKernelGGL(k_gen_sum, dim3(cdiv((long long)cn * want, 256)), dim3(256), 0, st, A, d_sym, d_taps, d_sum);
        hipLaunchKernelGGL(k_gen_phase, dim3(cdiv(cn, 64)), dim3(64), 0, st, A, d_sum);
        if (d_phase)
            HIPCHK(hipMemcpyAsync(d_phase + (size_t)c0 * want, d_sum, sizeof(float) * (size_t)cn * want, hipMemcpyDeviceToDevice, st));
        const int segs = (int)((want + GEN_SEG - 1) / GEN_SEG);
        hipLaunchKernelGGL(k_gen_iq, dim3((unsigned)((long long)cn * segs)), dim3(GEN_SEG), 0, st, A, NZ, d_sum,
                           d_iq + (size_t)c0 * want * 2);
        HIPCHK(hipGetLastError());
    }
    HIPCHK(hipStreamSynchronize(st));                       // the workspace is freed on return
    return M17GPU_OK;
}

int m17gpu_sync_samples(m17gpu_ctx *ctx, const float *d_disc, int nblk, int lock, float *d_syms,
                        int32_t *d_nsyms, void *stream)
{
    if (!ctx || !d_disc || !d_syms || nblk <= 0) return fail(M17GPU_ERR_ARG, "m17gpu_sync_samples: bad argument");
    ON_CTX_DEVICE(ctx);
    return launch_sync_frame(ctx, d_disc, nullptr, nblk, 0, nullptr, 0, nullptr, d_syms, d_nsyms, S(stream),
                             lock ? 1 : 0);
}

int m17gpu_viterbi_decode(m17gpu_ctx *ctx, const float *d_soft, uint8_t *d_bits, int len, int n, void *stream)
{
    if (!ctx || !d_soft || !d_bits || len <= 0 || len > 488 || (len & 1) || n <= 0)
        return fail(M17GPU_ERR_ARG, "m17gpu_viterbi_decode: len must be even and <= 488");
    ON_CTX_DEVICE(ctx);
    hipLaunchKernelGGL(k_viterbi, dim3(cdiv(n, DEC_FRAMES_PER_WG)), dim3(256), 0, S(stream), d_soft, d_bits, len, n);
    HIPCHK(hipGetLastError());
    return M17GPU_OK;
}

int m17gpu_demap_frame(m17gpu_ctx *ctx, const float *d_sym, float *d_soft, int n, void *stream)
{
    if (!ctx || !d_sym || !d_soft || n <= 0) return fail(M17GPU_ERR_ARG, "m17gpu_demap_frame: bad argument");
    ON_CTX_DEVICE(ctx);
    hipLaunchKernelGGL(k_demap, dim3(cdiv(n, DEC_FRAMES_PER_WG)), dim3(256), 0, S(stream), d_sym, d_soft, n);
    HIPCHK(hipGetLastError());
    return M17GPU_OK;
}

int m17gpu_decode_frames(m17gpu_ctx *ctx, const float *d_sym, const uint8_t *d_type, m17gpu_rec *d_recs, int n, void *stream)
{
    if (!ctx || !d_sym || !d_type || !d_recs || n <= 0) return fail(M17GPU_ERR_ARG, "m17gpu_decode_frames: bad argument");
    ON_CTX_DEVICE(ctx);
    hipStream_t st = S(stream);
    HIPCHK(hipMemsetAsync(d_recs, 0, sizeof(m17gpu_rec) * (size_t)n, st));
    int grid = cdiv(cdiv(n, DQ_FRAMES), 4);
    if (grid > 256 * 2) grid = 256 * 2;
    hipLaunchKernelGGL(k_decode_quad<0>, dim3(grid), dim3(256), 0, st, d_sym, (const int32_t *)nullptr,
                       (const int32_t *)nullptr, 0, d_type, n, reinterpret_cast<m17gpu_rec_dev *>(d_recs),
                       ctx->d_genc, ctx->d_gerr, 0, kFrameSyms);
    HIPCHK(hipGetLastError());
    return M17GPU_OK;
}

int m17gpu_golay_decode(m17gpu_ctx *ctx, const uint32_t *d_words, uint16_t *d_out, int n, void *stream)
{
    if (!ctx || !d_words || !d_out || n <= 0) return fail(M17GPU_ERR_ARG, "m17gpu_golay_decode: bad argument");
    ON_CTX_DEVICE(ctx);
    hipLaunchKernelGGL(k_golay, dim3(cdiv(n, 256)), dim3(256), 0, S(stream), d_words, d_out, n, ctx->d_genc, ctx->d_gerr);
    HIPCHK(hipGetLastError());
    return M17GPU_OK;
}

static int fetch_state(m17gpu_ctx *ctx, std::vector<ChanState> &h)
{
    h.resize((size_t)ctx->C);
    HIPCHK(hipDeviceSynchronize());
    HIPCHK(hipMemcpy(h.data(), ctx->d_state, sizeof(ChanState) * (size_t)ctx->C, hipMemcpyDeviceToHost));
    return M17GPU_OK;
}

int m17gpu_get_lsf(m17gpu_ctx *ctx, uint8_t *h_lsf)
{
    if (!ctx || !h_lsf) return fail(M17GPU_ERR_ARG, "m17gpu_get_lsf: bad argument");
    ON_CTX_DEVICE(ctx);
    std::vector<ChanState> h; int rc = fetch_state(ctx, h); if (rc) return rc;
    for (int c = 0; c < ctx->C; ++c)
        for (int k = 0; k < 2; ++k) std::memcpy(h_lsf + ((size_t)c * 2 + k) * 30, h[c].lsf[k], 30);
    return M17GPU_OK;
}

int m17gpu_get_counters(m17gpu_ctx *ctx, uint32_t *h_cnt)
{
    if (!ctx || !h_cnt) return fail(M17GPU_ERR_ARG, "m17gpu_get_counters: bad argument");
    ON_CTX_DEVICE(ctx);
    std::vector<ChanState> h; int rc = fetch_state(ctx, h); if (rc) return rc;
    for (int c = 0; c < ctx->C; ++c) {
        h_cnt[4 * c + 0] = h[c].g_errors; h_cnt[4 * c + 1] = h[c].n_frames;
        h_cnt[4 * c + 2] = h[c].in_frame; h_cnt[4 * c + 3] = h[c].frame_id_epoch;
    }
    return M17GPU_OK;
}

int m17gpu_get_afc(m17gpu_ctx *ctx, float *h_delta)
{
    if (!ctx || !h_delta) return fail(M17GPU_ERR_ARG, "m17gpu_get_afc: bad argument");
    ON_CTX_DEVICE(ctx);
    std::vector<ChanState> h; int rc = fetch_state(ctx, h); if (rc) return rc;
    for (int c = 0; c < ctx->C; ++c) h_delta[c] = h[c].afc_delta;
    return M17GPU_OK;
}

int m17gpu_get_lock(m17gpu_ctx *ctx, uint8_t *h_lock)
{
    if (!ctx || !h_lock) return fail(M17GPU_ERR_ARG, "m17gpu_get_lock: bad argument");
    ON_CTX_DEVICE(ctx);
    std::vector<ChanState> h; int rc = fetch_state(ctx, h); if (rc) return rc;
    for (int c = 0; c < ctx->C; ++c) h_lock[c] = (uint8_t)(h[c].flock != 0);
    return M17GPU_OK;
}

// m_clk, m_thr, m_index (m17_rx_sync.cpp:6-9), m_flock, m_fclk, m_frame_errors (m17_rx_frame.cpp:16-18) | the carried sum and
// dif (m17_rx_sync.cpp:78), z[0] and z[1] of dsp_arctan_disc2 (m17_dsp.cpp:196) as re, im, and m_buff[1 .. 30], the timing
// loop's delay line (m17_rx_sync.cpp:11; m_buff[0] leaves the window with the next input and is not kept)
int m17gpu_get_timing_state(m17gpu_ctx *ctx, int32_t *h_int, float *h_flt)
{
    if (!ctx || !h_int || !h_flt) return fail(M17GPU_ERR_ARG, "m17gpu_get_timing_state: bad argument");
    ON_CTX_DEVICE(ctx);
    std::vector<ChanState> h; int rc = fetch_state(ctx, h); if (rc) return rc;
    for (int c = 0; c < ctx->C; ++c) {
        const ChanState &s = h[c];
        int32_t *a = h_int + 6 * (size_t)c;
        float *f = h_flt + 36 * (size_t)c;
        a[0] = s.clk; a[1] = s.thr; a[2] = s.index; a[3] = s.flock; a[4] = s.fclk; a[5] = s.ferr;
        f[0] = s.sum; f[1] = s.dif; f[2] = s.z0re; f[3] = s.z0im; f[4] = s.z1re; f[5] = s.z1im;
        std::memcpy(f + 6, s.buff + 1, 30 * sizeof(float));
    }
    return M17GPU_OK;
}

// what the last m17gpu_rx_blocks call of the context ran (the library picks its kernels by call: DESIGN.md section 5)
int m17gpu_get_last_path(const m17gpu_ctx *ctx, int h_path[4])
{
    if (!ctx || !h_path) return fail(M17GPU_ERR_ARG, "m17gpu_get_last_path: bad argument");
    for (int i = 0; i < 4; ++i) h_path[i] = ctx->last_path[i];
    return M17GPU_OK;
}

int m17gpu_get_taps(float *h_mf, float *h_md)
{
    const m17::Tables &T = m17::tables();
    if (h_mf) std::memcpy(h_mf, T.mf, sizeof T.mf);
    if (h_md) std::memcpy(h_md, T.md, sizeof T.md);
    return M17GPU_OK;
}

int m17gpu_get_golay_tables(uint16_t *h_enc, uint16_t *h_err)
{
    const m17::Tables &T = m17::tables();
    if (h_enc) std::memcpy(h_enc, T.golay_enc, sizeof T.golay_enc);
    if (h_err) std::memcpy(h_err, T.golay_err, sizeof T.golay_err);
    return M17GPU_OK;
}

} // extern "C"
