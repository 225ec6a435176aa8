/*
 * m17_oracle.h -- CPU ORACLE (test infrastructure, NOT product code).
 *
 * A from-scratch plain-C restatement of the receive hot path of G4GUO/m17_sdr
 * (reference tree m17gismo/, tag v1).  Every function cites the reference
 * file:line whose arithmetic it follows.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load this library; the product path
 * (m17_sdr_amd/) never links, imports or calls it.
 *
 * PARITY PIN STATUS (see DESIGN.md "Oracle"):
 *   - codec primitives (CRC, Golay, interleaver, de-randomiser, puncture,
 *     convolutional code/Viterbi round trip, callsign, RRC tap design):
 *     pinned by the known-answer values SURVEY.md section 8(c) recorded from the
 *     compiled reference, and by the M17 specification constants.
 *   - streaming DSP stages (a3..a12: int16->float, limiter, discriminator,
 *     timing loop, framer): PARITY UNPINNED.  The reference cannot be built in
 *     this image without writing a stand-in for the absent codec2.h (every
 *     reference TU includes it through m17defines.h:5) and for its HAL/GUI
 *     translation units, which the build rules forbid; only behavioural KATs
 *     (loop-back lock / delivery counts from SURVEY.md 8c) are available.
 */
#ifndef M17_ORACLE_H
#define M17_ORACLE_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define M17O_BLOCK_SAMPLES 1920   /* m17defines.h:17 N_SAMPLES */
#define M17O_DISC_OUT       384   /* N_SAMPLES/5, m17_dsp.cpp:463 */
#define M17O_FRAME_SYMS     192   /* m17defines.h:66 */
#define M17O_NF              40   /* m17_rx_sync.cpp:3 */
#define M17O_FN              31   /* m17_rx_sync.cpp:4 */
#define M17O_SYM_MAX        200   /* >= 193 symbols per block */

/* record flags (identical numbering to include/m17gpu.h by convention) */
#define M17O_F_SYNC_OK    0x0001  /* m17_locked_sync_check passed */
#define M17O_F_PARSED     0x0002  /* m17_rx_parse was called */
#define M17O_F_LICH_OK    0x0004  /* update_lich: CRC(m_lsf[0])==0, good copy refreshed */
#define M17O_F_DELIVERED  0x0008  /* stream payload handed to the sink */
#define M17O_F_EOT        0x0010  /* EOT sync -> unlock */
#define M17O_F_LOST       0x0020  /* > N_FERROR bad syncs -> unlock */
#define M17O_F_LSF_GATE   0x0040  /* decode_link_frame CRC gate (m_packet quirk) open */
#define M17O_F_PKT_VALID  0x0080  /* parse_packet: CRC==0 at EOF */
#define M17O_F_AOS        0x0100  /* lock acquired (not a frame) */

typedef struct {
    uint8_t  type;          /* sync type 0..5 (m17_rx_frame.cpp:5-12 row) */
    uint8_t  votes;         /* sign mismatches vs winning template */
    uint8_t  golay_errs;    /* stream: sum of the four Golay weights */
    uint8_t  frame_errors;  /* m_frame_errors after this frame */
    uint16_t flags;
    uint16_t fn;            /* stream: FN; packet: (eof<<8)|fn */
    float    variance;
    uint32_t block;         /* block index since reset in which it completed */
    uint16_t sym_pos;       /* index of the completing symbol inside the block */
    uint16_t rsv0;
    uint8_t  data[32];      /* LSF: 30 B; stream: lich[6]+pld[18]; packet: 26 B */
    uint8_t  rsv[12];
} m17o_rec;                 /* 64 bytes */

typedef struct {
    /* front end, m17_dsp.cpp:195-196 */
    int32_t disc_count;
    float   z0re, z0im, z1re, z1im;
    /* timing loop, m17_rx_sync.cpp:6-11,43,78 */
    int32_t m_clk, m_thr, m_index;
    float   sum, dif;
    float   m_buff[M17O_FN];
    /* framer, m17_rx_frame.cpp:14-18,104 */
    int32_t m_flock, m_fclk, m_frame_errors;
    float   m_sync[8];
    float   m_f_sym[M17O_FRAME_SYMS];
    /* parser, m17_rx_parse.cpp:5-8 */
    uint8_t m_lsf[2][30];
    uint8_t pad0[4];
    uint8_t m_packet[800];
    int32_t m_packet_idx;
    /* bookkeeping mirrored from m17_dbase.cpp:60-82 */
    uint32_t g_errors, n_frames, in_frame, frame_id_epoch;
    uint32_t block_count;
    /* AFC (optional, off by default like radio.cpp:8): radio.cpp:9-10 m_afc / m_afc_delta, the NCO phase of
     * dsp_nco_mixer (m17_dsp.cpp:391).  Appended so that the offsets above stay what tests/oracle.py knows. */
    int32_t  afc_on;
    float    afc_delta;
    double   afc_acc;
} m17o_chan;

/* ---- init / tables ---- */
void m17o_init(void);                                  /* main.cpp:110-118 order */
void m17o_chan_reset(m17o_chan *st);
const float *m17o_tab_mf(void);                        /* [40][31] m17_rx_sync.cpp:12 */
const float *m17o_tab_md(void);                        /* [40][31] m17_rx_sync.cpp:13 */
const uint16_t *m17o_tab_golay_enc(void);              /* [4096] */
const uint16_t *m17o_tab_golay_err(void);              /* [4096] */
const uint8_t *m17o_tab_derand(void);                  /* [368] */
const uint16_t *m17o_tab_crc(void);                    /* [256] */
/* literal constants by name: "sframe" f32[6][8], "derand_bytes" u8[46], "golay_rows" u16[12],
 * "punc1|2|3" u8[61|12|8], "butterfly" u8[16][5] = BF(v,w,x,y,z), "crc_poly" u16; bytes written or -1 */
int m17o_get_constant(const char *name, void *out, int cap);

/* ---- primitives ---- */
void     m17o_build_rrc_filter(float *f, float rolloff, int ntaps, int sps);
void     m17o_set_filter_gain(float *f, float gain, int stride, int ntaps);
uint16_t m17o_crc(const uint8_t *in, int len);
uint32_t m17o_golay_encode(uint16_t data);
int      m17o_golay_decode(uint32_t word, uint16_t *odata);
int      m17o_conv_encode_8(const uint8_t *in, uint8_t *out, int len);
int      m17o_conv_encode_1(const uint8_t *in, uint8_t *out, int len);
int      m17o_punc(int which, const uint8_t *in, uint8_t *out, int len);
int      m17o_de_punc(int which, const float *in, float *out, int len);
void     m17o_interleave_u8(const uint8_t *in, uint8_t *out, int len);
void     m17o_de_interleave(const float *in, float *out, int len);
void     m17o_de_correlate_u8(const uint8_t *in, uint8_t *out, int len);
void     m17o_de_correlate_f(const float *in, float *out, int len);
int      m17o_viterbi_decode(const float *in, uint8_t *out, int len);
uint32_t m17o_hard_decode_24(const float *in);
int      m17o_pack_1_to_8(const uint8_t *in, uint8_t *out, int len);
void     m17o_demap_frame(const float *in, float *out);
uint64_t m17o_encode_call(const char *call);
void     m17o_decode_call(uint64_t w, char *call);
/* parse_lsf (m17_rx_parse.cpp:52-70) + m17_decode_call + m17_upack_type: 64 bytes, the layout of the product's
 * m17gpu_lsf_fields so that the two can be compared byte for byte */
typedef struct {
    uint64_t dst, src;
    char     dst_call[10], src_call[10];
    uint8_t  p_s, dt, et, est, can, reserved;
    uint8_t  meta[14];
    uint16_t crc;
    uint8_t  crc_ok;
} m17o_lsf_fields;
void     m17o_parse_lsf(const uint8_t *lsf, m17o_lsf_fields *out);
void     m17o_prbs9(uint8_t *out, int len);
void     m17o_sync_check(const float *v, uint8_t *type, uint8_t *votes, float *variance);

/* ---- streaming stages (one channel) ---- */
/* a3+a5+a6: int16 IQ[1920*2] -> d[384] (DC removed); raw (before DC removal)
 * and offset optionally returned */
void m17o_frontend(m17o_chan *st, const int16_t *iq, float *d, float *d_raw, float *offset);
/* a9: returns number of symbols (191..193) */
int  m17o_rx_sync_samples(m17o_chan *st, const float *in, float *out, int len);
/* a12+a13..a25: symbols -> records; returns number of records appended */
int  m17o_rx_symbols(m17o_chan *st, const float *sym, int n, m17o_rec *recs, int cap);
/* frame decode alone: float s[192] + type -> record fields (stateful LICH) */
void m17o_rx_parse(m17o_chan *st, const float *s, uint8_t type, m17o_rec *r);
/* a2: whole block; returns records appended; syms/nsym optional outputs */
int  m17o_dsp_rx(m17o_chan *st, const int16_t *iq, m17o_rec *recs, int cap,
                 float *syms, int *nsym);

/* ---- batch driver for the CPU baseline (OpenMP over channels) ----
 * iq: [C][nblk][1920][2]; recs: [C][cap]; counts: [C]; syms: per channel a
 * contiguous symbol stream [C][nblk*193+8] or NULL; nsyms: per-block symbol
 * counts [C][nblk] or NULL; mode 0 = front end only (a2 without parse), 1 = full chain */
/* the same with the network sink of decode_stream_frame attached (m17_net_new_rx_data, m17_net.cpp:25-74): net
 * [C][cap][56] receives the 54-byte M17-over-IP frame of every DELIVERED record at that record's index */
int  m17o_rx_blocks_net(m17o_chan *st, int C, int nblk, const int16_t *iq, m17o_rec *recs, int cap, int32_t *counts,
                        float *syms, int32_t *nsyms, int mode, int nthreads,
                        uint8_t *net, const uint16_t *stream_ids, uint64_t dst_override);
void m17o_format_net_frame(uint16_t stream_id, const uint8_t *lsf, uint16_t fn, const uint8_t *pld,
                           uint64_t dst_override, uint8_t *out);
int  m17o_rx_blocks(m17o_chan *st, int C, int nblk, const int16_t *iq,
                    m17o_rec *recs, int cap, int32_t *counts,
                    float *syms, int32_t *nsyms, int mode, int nthreads);
int  m17o_sizeof_chan(void);
/* AFC of one channel (radio_set_afc_on/off mmi.cpp:90-101; state radio.cpp:10, m17_dsp.cpp:391) */
void m17o_set_afc(m17o_chan *st, int on);
void m17o_get_afc(const m17o_chan *st, float *delta, double *acc);

/* ---- transmit side: the checker of the product's signal source ----
 * frame builders of m17_tx_routines.cpp:24-255 and the 4-FSK RRC modulator of m17_modulate.cpp:22-92 (10 samples per
 * symbol).  reference_quirks: 0 = the frame as specified; 1 = with the reference's tx_bit[2][388] / txb[2][388]
 * overrun restated (m17_tx_routines.cpp:93,203: the link-setup and packet frames the reference really sends). */
#define M17O_TX_FN 31             /* m17_modulate.cpp:6 TX_FN */
#define M17O_TX_OS 10             /* radio_get_oversample(), radio.cpp:207-215 */
typedef struct {
    float c[M17O_TX_FN * M17O_TX_OS];   /* m_tx_c */
    float s[M17O_TX_FN];                /* m_tx_s */
    float lu[4];                        /* m_tx_lu */
    float acc;                          /* m_acc */
} m17o_mod;
int  m17o_build_lsf(uint64_t dst, uint64_t src, uint16_t type_word, const uint8_t *meta, uint8_t *lsf);
int  m17o_preamble_dibits(uint8_t *dibits);
int  m17o_eot_dibits(uint8_t *dibits);
int  m17o_lsf_frame_dibits(const uint8_t *lsf, uint8_t *dibits, int reference_quirks);
int  m17o_stream_frame_dibits(const uint8_t *lsf, int lich_count, uint16_t fn, const uint8_t *payload, uint8_t *dibits);
int  m17o_packet_frame_dibits(const uint8_t *payload, int len, int eof, int nf, uint8_t *dibits, int reference_quirks);
void m17o_mod_init(m17o_mod *m);
int  m17o_modulate(m17o_mod *m, const uint8_t *dibits, int n, int16_t *iq, float *sums, float *phases);
int  m17o_sizeof_mod(void);

/* ---- wide-band ingest (radio.cpp:18-51,157-177): 31-tap symmetric /8 decimator, Q15 ---- */
void m17o_pluto_build_dec_filter(int16_t *coffs /* [31] */);
void m17o_pluto_decimate(int16_t *hist /* [31][2] state */, const int16_t *in /* [n_in][2] */, int n_in,
                         int16_t *out /* [n_in/8][2] */);

#ifdef __cplusplus
}
#endif
#endif
