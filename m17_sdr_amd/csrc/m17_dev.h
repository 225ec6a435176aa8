// m17_dev.h -- device-side data layout shared by the kernels and the C-ABI host code.
#pragma once
#include <cstdint>
#include <cstddef>

namespace m17dev {

constexpr int kBlockSamples = 1920;   // m17defines.h:17
constexpr int kDiscOut      = 384;    // m17_dsp.cpp:463
constexpr int kFrameSyms    = 192;    // m17defines.h:66
// One frame slot of the decoder's workspace.  Link-setup and packet frames are stored as their 192 symbols.  A stream
// frame is stored in the order its decoder reads it (DevTables.regroup), so that the decoder's loads are contiguous --
// gathered from the plain 192 symbols, every soft bit cost an L2 round trip of a whole cache line:
//   [0,8) sync symbols; [8,104) the source symbols of the 96 LICH soft bits; [104,400) those of the 296 de-punctured
//   payload soft bits (an erasure's place holds some symbol, never used); [400,424) not written (the last
//   chunk reads it and masks it out).
constexpr int kSlotFloats   = 424;
constexpr int kRegroup      = 392;
constexpr int kSoftBits     = 368;
constexpr int kPhases       = 40;     // m17_rx_sync.cpp:3
constexpr int kTaps         = 31;     // m17_rx_sync.cpp:4

// The literals and control constants of the reference's streaming arithmetic, each under ONE name: the kernels and the
// host tables use these names where the reference has the literal, and m17gpu_get_constant("rx_literals") returns them in
// this order -- tests/test_ref_constants.py holds that array against the values tests/golden/extract_ref_constants.py
// found in the reference's source text (and against the oracle's own), so a mistyped threshold cannot hide.
#define M17_LIT_S16_SCALE           0.00003   /* m17_dsp.cpp:138-139  out = in * 0.00003 (double) */
#define M17_LIT_DEMAP_OFFSET        0.6666    /* m17_dsp.cpp:41       fabs(m) - 0.6666 (double) */
#define M17_LIT_DEMAP_COR_NUM       8.0       /* m17_dsp.cpp:88       cor = 8.0 / sum (double) */
#define M17_LIT_DEMAP_SYNC_SYMBOLS  8         /* m17_dsp.cpp:85       sum over the 8 sync symbols */
#define M17_LIT_DISC_C              0.5f      /* m17_dsp.cpp:199      u * c */
#define M17_LIT_DISC_DECIM          5         /* m17_dsp.cpp:207      count = (count + 1) % 5 */
#define M17_LIT_LIMIT_NUM           1.0       /* m17_dsp.cpp:415      g = 1.0 / m (double) */
#define M17_LIT_THRESH_UNLOCKED     10        /* m17_rx_sync.cpp:93 */
#define M17_LIT_THRESH_LOCKED       80        /* m17_rx_sync.cpp:95 */
#define M17_LIT_CLK_MODULUS         2         /* m17_rx_sync.cpp:82   m_clk = (m_clk + 1) % 2 */
#define M17_LIT_CLK_INIT            1         /* m17_rx_sync.cpp:123 */
#define M17_LIT_THR_INIT            0         /* m17_rx_sync.cpp:125 */
#define M17_LIT_INDEX_INIT          10        /* m17_rx_sync.cpp:126 */
#define M17_LIT_VOTES_UNLOCKED_MAX  0         /* m17_rx_frame.cpp:83  votes > 0 rejects */
#define M17_LIT_VAR_UNLOCKED        0.3       /* m17_rx_frame.cpp:87  variance < 0.3 (double) */
#define M17_LIT_VOTES_LOCKED_MAX    1         /* m17_rx_frame.cpp:94 */
#define M17_LIT_VAR_LOCKED          0.5       /* m17_rx_frame.cpp:98 */
#define M17_LIT_N_FERROR            5         /* m17_rx_frame.cpp:122 */
#define M17_LIT_FCLK_AFTER_SYNC     8         /* m17_rx_frame.cpp:166 */
#define M17_LIT_ACM0                1.0f      /* m17_conv.cpp:153     m_acm[0] = 1.0 */
#define M17_LIT_TRACEBACK_MASK      0x08      /* m17_conv.cpp:165     out = state & 0x08 */
#define M17_LIT_GOLAY_FILL_END      0xFFF     /* m17_golay.cpp:53     for (i = 0; i < 0xFFF; ...) -- entry 0xFFF stays 0 */
#define M17_LIT_GOLAY_UNRECOVERABLE 0x400     /* m17_golay.cpp:54 */
#define M17_LIT_GOLAY_MAX_BITS      5         /* m17_golay.cpp:61     bits < 5 */
#define M17_RX_LITERALS { M17_LIT_S16_SCALE, M17_LIT_DEMAP_OFFSET, M17_LIT_DEMAP_COR_NUM, M17_LIT_DEMAP_SYNC_SYMBOLS, M17_LIT_DISC_C, \
    M17_LIT_DISC_DECIM, M17_LIT_LIMIT_NUM, M17_LIT_THRESH_UNLOCKED, M17_LIT_THRESH_LOCKED, M17_LIT_CLK_MODULUS, M17_LIT_CLK_INIT,    \
    M17_LIT_THR_INIT, M17_LIT_INDEX_INIT, M17_LIT_VOTES_UNLOCKED_MAX, M17_LIT_VAR_UNLOCKED, M17_LIT_VOTES_LOCKED_MAX,               \
    M17_LIT_VAR_LOCKED, M17_LIT_N_FERROR, M17_LIT_FCLK_AFTER_SYNC, M17_LIT_ACM0, M17_LIT_TRACEBACK_MASK, M17_LIT_GOLAY_FILL_END,    \
    M17_LIT_GOLAY_UNRECOVERABLE, M17_LIT_GOLAY_MAX_BITS }
static_assert(kBlockSamples % M17_LIT_DISC_DECIM == 0 && kDiscOut == kBlockSamples / M17_LIT_DISC_DECIM, "the /5 pick keeps its phase from block to block");
static_assert(kDiscOut % M17_LIT_CLK_MODULUS == 0 && M17_LIT_CLK_MODULUS == 2 && M17_LIT_TRACEBACK_MASK == 0x08 && M17_LIT_DEMAP_SYNC_SYMBOLS == 8 &&
              M17_LIT_THR_INIT == 0 && M17_LIT_LIMIT_NUM == 1.0 && M17_LIT_DISC_C == 0.5f,
              "built into the kernels' structure: two inputs per symbol, bits[t] = state >> 3, eight sync symbols, reciprocal and halving");

#ifdef __HIPCC__
#define M17_HD __host__ __device__
#else
#define M17_HD
#endif
// The six sync templates sframe[6][8] (m17_rx_frame.cpp:5-12) as bit masks: bit i of row k set
// <=> template symbol i of class k is -1 (preamble, link setup, stream, packet, BERT, EOT).
// The one definition every kernel and the host accessor m17gpu_get_constant("sframe") use.
#define M17_SYNC_NEG_MASKS {0xAA, 0xB0, 0x4F, 0xF2, 0x0D, 0x40}
M17_HD constexpr unsigned sync_neg_mask(int k)
{
    constexpr unsigned m[6] = M17_SYNC_NEG_MASKS;
    return m[k];
}

#define M17_SYM_STRIDE(nblk) ((size_t)(nblk) * 193 + 8)

#define M17_F_SYNC_OK    0x0001u
#define M17_F_PARSED     0x0002u
#define M17_F_LICH_OK    0x0004u
#define M17_F_DELIVERED  0x0008u
#define M17_F_EOT        0x0010u
#define M17_F_LOST       0x0020u
#define M17_F_LSF_GATE   0x0040u
#define M17_F_PKT_VALID  0x0080u
#define M17_F_AOS        0x0100u

// binary-identical to m17gpu_rec (include/m17gpu.h)
struct m17gpu_rec_dev {
    uint8_t  type, votes, golay_errs, frame_errors;
    uint16_t flags, fn;
    float    variance;
    uint32_t block;
    uint16_t sym_pos, rsv0;
    uint8_t  data[32];
    uint8_t  rsv[12];
};
static_assert(sizeof(m17gpu_rec_dev) == 64, "record must be 64 bytes");

// Per-channel state in HBM: the reference's file-static variables of one
// receiver instance (SURVEY.md section 5 "stream continuity"), 1,920 bytes.
struct ChanState {
    // discriminator memory z[0], z[1]                      m17_dsp.cpp:196
    float    z0re, z0im, z1re, z1im;
    // timing loop                                           m17_rx_sync.cpp:6-9,78
    int32_t  clk, thr, index;
    float    sum, dif;
    // framer                                                m17_rx_frame.cpp:16-18
    int32_t  flock, fclk, ferr;
    uint32_t block_count;
    // m17_dbase.cpp:60-82 mirrors
    uint32_t g_errors, n_frames, in_frame, frame_id_epoch;
    int32_t  packet_idx;                                  // m17_rx_parse.cpp:7
    int32_t  sym_total;                                   // symbols emitted so far in the current call (pipelined launches)
    float    afc_delta;                                   // m_afc_delta                 radio.cpp:10   (AFC contexts only)
    double   afc_acc;                                     // NCO phase of dsp_nco_mixer  m17_dsp.cpp:391
    int32_t  pad[10];
    float    buff[32];                                    // m_buff[31]           m17_rx_sync.cpp:11
    float    sync[8];                                     // m_sync               m17_rx_frame.cpp:104
    float    fsym[kFrameSyms];                            // m_f_sym              m17_rx_frame.cpp:14
    uint8_t  lsf[2][32];                                  // m_lsf[2][30]         m17_rx_parse.cpp:5
    uint8_t  packet[800];                                 // m_packet             m17_rx_parse.cpp:6
};
static_assert(sizeof(ChanState) == 1920, "ChanState layout");

// one LICH soft-bit source re-coded for the decoder (m17_decode_quad.hip: dq_lich_entry)
struct DqLich { uint32_t e; float T; };

// constant tables (built on the host by m17::tables())
struct DevTables {
    float    mf[kPhases][32];        // matched filter taps, 31 used
    float    md[kPhases][32];        // derivative filter taps
    float    tap_pairs[kPhases][64]; // the same taps as (matched, derivative) pairs, 32 pairs per branch (pair 31 = 0)
    int16_t  gather[4][488];         // -1 erasure, else src | 0x4000 when negated
    int16_t  lich[96];
    DqLich   lich_q[96];             // lich[] as the decoder reads it
    alignas(4) uint8_t regroup[kRegroup];   // stream frame slot: payload symbol (0..183) behind slot entry 8 + i
    int16_t  glen[4];
    uint8_t  bm_even[16], bm_odd[16];
    uint16_t crc[256];
};

} // namespace m17dev
