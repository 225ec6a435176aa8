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
