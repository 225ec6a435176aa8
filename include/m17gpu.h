/*
 * m17gpu.h -- C-ABI of the MI355X-native M17 receive chain.
 *
 * This is the drop-in boundary for the receive hot path of G4GUO/m17_sdr
 * (reference tree m17gismo/): one process-wide shared library, `extern "C"`,
 * plain pointers and sizes.  The reference runs ONE channel per process through
 * file-static state; this core is re-entrant and batched:
 * an explicit context holds the per-channel state of C independent 48 kHz
 * channels in HBM and every entry point processes all of them in one launch.
 *
 * Pointer convention: arguments named d_* are DEVICE pointers (HBM, e.g. from
 * hipMalloc or torch.Tensor.data_ptr()) on the context's device; h_* are host pointers.
 * `stream` is a hipStream_t of that device passed as void* (NULL = default stream).  Every
 * entry point selects the context's device for its own duration and restores the caller's.  All calls are
 * asynchronous on `stream` unless stated.  Return value: 0 = ok, <0 = error
 * (m17gpu_last_error() gives the text).  There is NO CPU fallback: without a
 * HIP device every compute entry point fails with M17GPU_ERR_NO_DEVICE.
 *
 * Citations (file:line) name the reference interface each entry replaces.
 */
#ifndef M17GPU_H
#define M17GPU_H
#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define M17GPU_BLOCK_SAMPLES 1920   /* m17defines.h:17  N_SAMPLES (40 ms @ 48 kHz) */
#define M17GPU_DISC_OUT       384   /* m17_dsp.cpp:463  N_SAMPLES/5 */
#define M17GPU_FRAME_SYMS     192   /* m17defines.h:66  FRAME_SYM_LENGTH */
#define M17GPU_SOFT_BITS      368   /* 184 payload symbols x 2 */
#define M17GPU_SYM_STRIDE(nblk) ((size_t)(nblk) * 193 + 8)  /* floats per channel in d_syms */

#define M17GPU_OK             0
#define M17GPU_ERR_NO_DEVICE (-1)
#define M17GPU_ERR_HIP       (-2)
#define M17GPU_ERR_ARG       (-3)
#define M17GPU_ERR_NOMEM     (-4)

/* frame-record flags */
#define M17GPU_F_SYNC_OK    0x0001  /* m17_locked_sync_check passed   (m17_rx_frame.cpp:93-103) */
#define M17GPU_F_PARSED     0x0002  /* m17_rx_parse was invoked       (m17_rx_frame.cpp:142,151) */
#define M17GPU_F_LICH_OK    0x0004  /* update_lich found CRC==0       (m17_rx_parse.cpp:78-83) */
#define M17GPU_F_DELIVERED  0x0008  /* payload handed to the sink     (m17_rx_parse.cpp:148-158) */
#define M17GPU_F_EOT        0x0010  /* EOT sync, framer unlocked      (m17_rx_frame.cpp:136-139) */
#define M17GPU_F_LOST       0x0020  /* > N_FERROR bad syncs, unlocked (m17_rx_frame.cpp:145-149) */
#define M17GPU_F_LSF_GATE   0x0040  /* decode_link_frame's CRC gate   (m17_rx_parse.cpp:98, quirk H9) */
#define M17GPU_F_PKT_VALID  0x0080  /* parse_packet CRC==0 at EOF     (m17_rx_parse.cpp:42-45) */
#define M17GPU_F_AOS        0x0100  /* lock acquired, not a frame     (m17_rx_frame.cpp:165-169) */

/* One 64-byte record per framer event of one channel, in event order.
 * It carries what the reference hands to its sinks (m17_net_new_rx_data
 * m17_net.cpp:53, m17_db_golay_errors m17_dbase.cpp:79, m17_aos/m17_los :60-75). */
typedef struct {
    uint8_t  type;          /* sync class 0..5: preamble, LSF, stream, packet, BERT, EOT */
    uint8_t  votes;         /* sign mismatches against the winning template */
    uint8_t  golay_errs;    /* stream frames: sum of the four Golay weights */
    uint8_t  frame_errors;  /* m_frame_errors after this frame */
    uint16_t flags;         /* M17GPU_F_* */
    uint16_t fn;            /* stream: frame number; packet: (eof<<8)|fn */
    float    variance;      /* amplitude spread of the 8 sync symbols */
    uint32_t block;         /* 1920-sample block index since reset */
    uint16_t sym_pos;       /* index of the completing symbol inside that block */
    uint16_t rsv0;
    uint8_t  data[32];      /* LSF: 30 B | stream: LICH chunk[6] + FN/payload[18] | packet: 26 B */
    uint8_t  rsv[12];
} m17gpu_rec;

typedef struct m17gpu_ctx m17gpu_ctx;

/* ---------------- lifecycle ---------------- */
/* Builds the tables the reference builds in main.cpp:110-118 (CRC LUT, conv
 * LUT, de-randomiser bits, Golay tables, 2x40x31 polyphase RRC taps), uploads
 * them, and allocates state + workspace for n_channels channels and up to
 * max_blocks 1920-sample blocks per call.  Synchronous. */
int  m17gpu_create(m17gpu_ctx **ctx, int n_channels, int max_blocks, int device);
void m17gpu_destroy(m17gpu_ctx *ctx);
/* zero-initialised statics + m17_rx_sync_init's m_clk=1,m_thr=0,m_index=10 */
int  m17gpu_reset(m17gpu_ctx *ctx, void *stream);
const char *m17gpu_last_error(void);
int  m17gpu_device_count(void);            /* 0 when no HIP device is visible */
int  m17gpu_channels(const m17gpu_ctx *ctx);

/* ---------------- the hot path ---------------- */
/* Batched m17_dsp_rx (m17_dsp.cpp:461-476) over C channels x nblk blocks.
 *   d_iq     [C][nblk][1920][2] int16  (scmplx, m17defines.h:130-133)
 *   mode     0 = front end only (discriminator, timing recovery, sync
 *                correlator/framer; records carry sync fields only)
 *            1 = full chain (+ demap, de-randomise, de-interleave, de-puncture,
 *                Viterbi, Golay, LICH/LSF bookkeeping)
 *   d_recs   [C][rec_cap] records, d_counts [C] number of events per channel.
 *            mode 0: events beyond rec_cap are counted, not stored.
 *            mode 1: 2*nblk+2 <= rec_cap <= 2*max_blocks+2 is required (M17GPU_ERR_ARG otherwise):
 *            a block yields at most two events on average, so no event is ever dropped and every
 *            frame reaches the LICH / counter / packet bookkeeping, as in the reference.
 *   d_syms   optional [C][M17GPU_SYM_STRIDE(nblk)] recovered symbols
 *            (m17_rx_sync_samples output), d_nsyms optional [C][nblk] counts */
int m17gpu_rx_blocks(m17gpu_ctx *ctx, const int16_t *d_iq, int nblk, int mode,
                     m17gpu_rec *d_recs, int rec_cap, int32_t *d_counts,
                     float *d_syms, int32_t *d_nsyms, void *stream);
/* Ordering: the calls of ONE context must be ordered on the device -- made on one stream, or on streams the caller has
 * ordered with events: a call reads the channel state its predecessor wrote, and the work-list counters of a full-chain
 * call are left zeroed for the next one by its last kernel.  Contexts are independent of each other. */

/* ---------------- stage entry points (batched reference functions) -------- */
/* dsp_short_to_float + dsp_limit + dsp_arctan_disc2 (m17_dsp.cpp:136-141,
 * :412-419, :194-222): d_disc [C][nblk][384] DC-removed discriminator output,
 * d_offset [C][nblk] the per-block DC estimate.  Advances z[2] state. */
int m17gpu_frontend(m17gpu_ctx *ctx, const int16_t *d_iq, int nblk,
                    float *d_disc, float *d_offset, void *stream);
/* m17_rx_sync_samples + m17_rx_symbols without the parse (m17_rx_sync.cpp:77-99,
 * m17_rx_frame.cpp:126-177) from a discriminator stream d_disc [C][nblk][384]. */
int m17gpu_sync_frame(m17gpu_ctx *ctx, const float *d_disc, int nblk,
                      m17gpu_rec *d_recs, int rec_cap, int32_t *d_counts,
                      float *d_syms, int32_t *d_nsyms, void *stream);
/* Wide-band ingest ahead of the hot path: the Pluto receive decimator,
 * rx_decimate_filter / sub_filter (radio.cpp:18-40) as run by radio_receive_samples
 * (:157-177): d_in [C][n_in][2] int16 at 384 kHz -> d_out [C][n_in/8][2] int16 at 48 kHz,
 * 31-tap symmetric Q15 low-pass, >> 15; the 31-sample history of each channel lives in the
 * context (zero after m17gpu_reset).  n_in: multiple of 8, >= 32. */
int m17gpu_pluto_decimate(m17gpu_ctx *ctx, const int16_t *d_in, int n_in, int16_t *d_out, void *stream);
/* m17_rx_sync_samples alone (m17_rx_sync.cpp:77-99): timing recovery with the lock
 * flag of an EXTERNAL framer (what m17_rx_lock() returns, m17_rx_frame.cpp:187),
 * the same flag for every channel and block of the call.  Only the timing state (m_buff, m_clk,
 * m_thr, m_index, sum, dif) advances; the context's framer state, hunt window, block counter and
 * record counts are untouched. */
int m17gpu_sync_samples(m17gpu_ctx *ctx, const float *d_disc, int nblk, int lock,
                        float *d_syms, int32_t *d_nsyms, void *stream);
/* m17_viterbi_decode (m17_conv.cpp:148-168) on n independent soft-bit vectors:
 * d_soft [n][len] floats -> d_bits [n][len/2] one bit per byte.  len <= 488, even. */
int m17gpu_viterbi_decode(m17gpu_ctx *ctx, const float *d_soft, uint8_t *d_bits,
                          int len, int n, void *stream);
/* m17_dsp_demap_frame (m17_dsp.cpp:82-95): d_sym [n][192] -> d_soft [n][368] */
int m17gpu_demap_frame(m17gpu_ctx *ctx, const float *d_sym, float *d_soft, int n, void *stream);
/* stateless part of m17_rx_parse (m17_rx_parse.cpp:86-177): frame symbols
 * d_sym [n][192] + d_type [n] -> data/fn/golay_errs of d_recs [n]; no LICH state. */
int m17gpu_decode_frames(m17gpu_ctx *ctx, const float *d_sym, const uint8_t *d_type,
                         m17gpu_rec *d_recs, int n, void *stream);
/* m_17_golay_decode (m17_golay.cpp:103-116) on n 24-bit words: d_out[i] = data | weight<<12 */
int m17gpu_golay_decode(m17gpu_ctx *ctx, const uint32_t *d_words, uint16_t *d_out, int n, void *stream);

/* Kernel-variant selectors (A/B measurement, and so the parity tests cover every variant).
 * Every accepted value selects a kernel held to bit-exact parity; anything else returns
 * M17GPU_ERR_ARG:
 *   "sync_impl"          0 | 6 = by size (default): timing wave + framer wave per channel, decoupled by one
 *                            block, up to 1,024 channels; beyond, one wave per channel with scalar control and the
 *                            filter taps in SGPRs, taps and window through half the registers at eight waves per
 *                            SIMD; 8 = that kernel at every size
 *   "fe_impl"            the stand-alone front end (fir_impl 1): 0 | 2 = four lanes per channel-block (default);
 *                            3 / 4 = the sixteen-row tiles the fused kernels run (k_sync_frame_duo<1>'s on 64-sample
 *                            chunks, k_rx_chan6's on 32-sample chunks) as kernels of their own, so that each tile is
 *                            held to the oracle by itself through m17gpu_frontend
 *   "fir_impl"           0 = by call (default): 5 up to 1,024 channels (calls of >= 16 blocks: from 512 channels on),
 *                            4 on >= 10,000 channels for calls of >= 12 blocks, else 1;
 *                            1 = front end and timing / framer as two kernels; 4 = a wave per channel that runs the front
 *                            end over sixteen of its own blocks at a time and the timing loop / framer behind it, the
 *                            rows through the workspace, six waves per SIMD (k_rx_chan6; a last group of fewer than
 *                            sixteen blocks shares its tiles among the four channels of a workgroup);
 *                            5 = up to 1,024 channels: front end, timing loop and framer of a channel on three waves of
 *                            one workgroup, the front end sixteen blocks ahead of the timing loop (k_sync_frame_duo<1>;
 *                            calls of up to eight blocks start on four-row tiles; falls back to 1 where the two-wave
 *                            timing kernel does not apply)
 *   "slot_impl"          how the framer hands a stream frame to the decoder: 1 = its 192 symbols (768 B; the decoder stages
 *                            them in LDS), 2 = regrouped into the order the decoder reads (1,600 B), 0 = by path (default):
 *                            1 behind the wave-per-channel FIR stage, 2 behind front end + timing kernel
 *   "book_impl"          the bookkeeping kernel: 0 = by batch (default): a lane per channel (k_book_lanes) from 8,192 channels
 *                            on, a wave per channel (k_book_chan) below and whenever the network sink is attached; 1 / 2 force
 *                            the wave / the lane kernel (2 still yields to the network sink)
 * and one functional switch:
 *   "afc"                0 (default, as the reference ships: radio.cpp:8) | 1 = radio_set_afc_on(): the
 *                            NCO mixer of m17_dsp.cpp:390-408,468 with the loop of radio.cpp:196-208 per channel.
 *                            The correction closes a loop over the whole chain, so an AFC context is processed
 *                            block by block; its double-precision cos/sin make this the one path held to
 *                            tolerance parity (same payloads, correction within 1e-4) instead of bit parity. */
int m17gpu_set_option(m17gpu_ctx *ctx, const char *name, int value);

/* ---------------- measurement hooks ---------------- */
/* When on, m17gpu_rx_blocks brackets each of its kernels with HIP events on the
 * launch stream (up to 512 calls are kept).  m17gpu_get_kernel_ms waits for the
 * recorded events and returns the average milliseconds per launch of
 * {k_frontend, k_sync_frame, k_worklist + k_decode, k_bookkeeping} and the number of calls averaged,
 * then clears the record. */
int m17gpu_set_profiling(m17gpu_ctx *ctx, int on);
int m17gpu_get_kernel_ms(m17gpu_ctx *ctx, float h_ms[4], int *h_calls);
/* The same, plus the average duration of a whole m17gpu_rx_blocks call between two events on the caller's stream
 * (kernels, the gaps between them and the small memset included). */
int m17gpu_get_call_ms(m17gpu_ctx *ctx, float h_ms[4], float *h_call_ms, int *h_calls);

/* Exhaustive on-device equivalence check of the front end's shortened exact
 * arithmetic (int16 scaling of m17_dsp.cpp:136-141, sqrt and reciprocal of
 * dsp_limit :412-419) against the literal fp64 / IEEE expressions.
 * h_bad[4] = mismatches of {scale, sqrt, reciprocal, composed limiter}; all 0
 * is required for the bit-exactness claim.  Synchronous, ~0.1 s. */
int m17gpu_selftest(m17gpu_ctx *ctx, unsigned *h_bad);

/* ---------------- state access (host, synchronous) ---------------- */
/* the reassembled LSF pair m_lsf[2][30] of each channel (m17_rx_parse.cpp:5) */
int m17gpu_get_lsf(m17gpu_ctx *ctx, uint8_t *h_lsf /* [C][2][30] */);
/* g_errors, n_frames, in_frame, frame_id_epoch per channel (m17_dbase.cpp:60-82) */
int m17gpu_get_counters(m17gpu_ctx *ctx, uint32_t *h_cnt /* [C][4] */);
/* m_afc_delta of every channel (radio.cpp:10), radians per sample; 0 while AFC is off or outside a frame */
int m17gpu_get_afc(m17gpu_ctx *ctx, float *h_delta /* [C] */);
/* m17_rx_lock() of every channel (m17_rx_frame.cpp:187-189) */
int m17gpu_get_lock(m17gpu_ctx *ctx, uint8_t *h_lock /* [C] */);
/* The timing loop's and the framer's control state of every channel, for inspection / tests:
 *   h_int [C][6]  = m_clk, m_thr, m_index (m17_rx_sync.cpp:6-9), m_flock, m_fclk, m_frame_errors (m17_rx_frame.cpp:16-18)
 *   h_flt [C][36] = sum, dif (m17_rx_sync.cpp:78), z[0].re, z[0].im, z[1].re, z[1].im (m17_dsp.cpp:196), m_buff[1 .. 30]
 *                   (m17_rx_sync.cpp:11; m_buff[0] leaves the window with the next input and is not kept) */
int m17gpu_get_timing_state(m17gpu_ctx *ctx, int32_t *h_int, float *h_flt);
/* What the last m17gpu_rx_blocks call ran (the library picks its kernels by call): h_path[0] = FIR stage ("fir_impl"
 * value: 1, 4, 5), [1] = stream frame slots plain (1) or regrouped (0), [2] = bookkeeping kernel (1 wave / 2 lane per
 * channel, 0 = none: mode 0), [3] = 0 (reserved). */
int m17gpu_get_last_path(const m17gpu_ctx *ctx, int h_path[4]);
/* host copies of the uploaded tables, for inspection / tests */
int m17gpu_get_taps(float *h_mf /* [40][31] */, float *h_md /* [40][31] */);
int m17gpu_get_golay_tables(uint16_t *h_enc /* [4096] */, uint16_t *h_err /* [4096] */);
/* The literal constants the library is built from, by name (host only, no device needed); returns
 * the number of bytes written or M17GPU_ERR_ARG.  "sframe" float[6][8] (m17_rx_frame.cpp:5-12),
 * "derand_bits" u8[368] (m17_correlate.cpp:3-7,35-42), "golay_rows" u16[12] (m17_golay.cpp:11),
 * "punc1" u8[61] / "punc2" u8[12] / "punc3" u8[8] (m17_puncture.cpp:4-10), "butterfly" u8[16][5] =
 * the BF(v,w,x,y,z) rows (m17_conv.cpp:93-108), "crc_poly" u16 (m17_crc.cpp:4), "tx_lut" float[4]
 * (m17_modulate.cpp:9), "sync_words" u16[4] link/stream/packet/BERT (m17_tx_routines.cpp:6-9), "rx_literals" double[24] =
 * the literals and control constants of the streaming arithmetic, each under the one name the kernels use it by (int16 scale
 * 0.00003 m17_dsp.cpp:138, demapper 0.6666 :41 and 8.0 :88 over 8 sync symbols :85, discriminator 0.5 :199 and % 5 :207, limiter
 * 1.0 / m :415, thresholds 10 / 80 m17_rx_sync.cpp:93,95, % 2 :82, initial m_clk 1 / m_thr 0 / m_index 10 :123-126, framer gates
 * votes > 0 / < 0.3 and votes > 1 / < 0.5 m17_rx_frame.cpp:83-98, N_FERROR 5 :122, m_fclk 8 :166, Viterbi m_acm[0] 1.0 and
 * state & 0x08 m17_conv.cpp:153,165, Golay table fill 0xFFF / 0x400 / bits < 5 m17_golay.cpp:53-61): tests/test_ref_constants.py
 * holds them against the values extracted from the reference's source text. */
int m17gpu_get_constant(const char *name, void *h_out, int cap_bytes);

/* ---------------- output wire format (host) ----------------
 * The 54-byte M17-over-IP stream frame of the reference's reflector client
 * (net_add_* m17_net.cpp:25-49, m17_net_new_rx_data :53-74) built from a delivered
 * record: stream id, the CRC-good LSF (m17gpu_get_lsf row [c][1]), r.fn, &r.data[8].
 * dst_override: 0 = keep the LSF's destination, else the 48-bit callsign to put there.
 * Returns 54. */
int m17gpu_format_net_frame(uint16_t stream_id, const uint8_t lsf[30], uint16_t fn, const uint8_t payload[16],
                            uint64_t dst_override, uint8_t out[54]);

/* LSF field extraction (parse_lsf m17_rx_parse.cpp:52-70 with m17_decode_call
 * m17_bit_utils.cpp:209-226 and m17_upack_type :245-254): what the reference hands to
 * gui_save_dest_address / gui_save_src_address / valid_lsf_received.  The base-40
 * callsign has its first character least significant; 0xFFFFFFFFFFFF reads "BROADCAST". */
typedef struct {
    uint64_t dst, src;            /* 48-bit encoded addresses */
    char     dst_call[10], src_call[10];
    uint8_t  p_s, dt, et, est, can, reserved;     /* type word, packet/stream bit upward */
    uint8_t  meta[14];
    uint16_t crc;                 /* lsf[28..29]; crc_ok = CRC-16 over all 30 bytes is 0 */
    uint8_t  crc_ok;
} m17gpu_lsf_fields;
int m17gpu_parse_lsf(const uint8_t lsf[30], m17gpu_lsf_fields *out);

/* ---------------- multi-GPU fan-out for a C / C++ host (RCCL over xGMI) ----------------
 * The path shards by channel (the reference has one channel per process, m17_tx_rx.cpp:28-40; here rank r of
 * `world` owns the contiguous range m17gpu_shard_range gives and a context of exactly that many channels).  No
 * collective sits on the data path; per step the ingest rank fans the IQ out and the records come back:
 *   m17gpu_shard_scatter_iq      src_rank holds d_iq_all [n_channels_total][nblk][1920][2]; every rank receives its
 *                                range into d_iq_mine [C][nblk][1920][2].  ONE ncclGroupStart/End: the sends to
 *                                the world-1 peers are in flight together, one xGMI link each.
 *   m17gpu_shard_gather_records  every rank's d_recs_mine [C][rec_cap] + d_counts_mine [C] land on dst_rank in
 *                                d_recs_all [n_channels_total][rec_cap], d_counts_all [n_channels_total].
 * comm is the caller's ncclComm_t (passed as void* so that this header needs no RCCL header); both calls are
 * enqueued on `stream` and return at once.  RCCL is looked up in the process at first use (the copy already
 * loaded, else librccl.so.1); without it these two entries return M17GPU_ERR_HIP and nothing else is affected. */
void m17gpu_shard_range(int rank, int world, int n_channels_total, int *lo, int *hi);
int m17gpu_shard_scatter_iq(m17gpu_ctx *ctx, void *comm, int rank, int world, int src_rank,
                            const int16_t *d_iq_all, int n_channels_total, int nblk, int16_t *d_iq_mine, void *stream);
int m17gpu_shard_gather_records(m17gpu_ctx *ctx, void *comm, int rank, int world, int dst_rank,
                                const m17gpu_rec *d_recs_mine, const int32_t *d_counts_mine, int rec_cap,
                                int n_channels_total, m17gpu_rec *d_recs_all, int32_t *d_counts_all, void *stream);

/* Packed gather (what SURVEY 8e budgets: about 1 MB per GPU per step).  m17gpu_rx_blocks leaves recs [C][rec_cap] with
 * counts [C] valid rows; only the valid rows need to cross xGMI:
 *   m17gpu_pack_records         d_offsets [C+1] = exclusive scan of the counts (d_offsets[C] = records in this step),
 *                               d_packed [packed_cap] = the valid records, channel-major, event order inside a channel.
 *                               Enqueued on `stream`, nothing read back.
 *                               packed_cap must be >= the sum of the counts: rows beyond it are NOT written, while
 *                               d_offsets still describes all of them -- m17gpu_shard_gather_packed checks
 *                               d_offsets[C] against the capacity it is given and refuses the step on every rank.
 *   m17gpu_shard_gather_packed  every rank's packed rows (d_packed_mine [packed_cap_mine], the capacity given to
 *                               m17gpu_pack_records) land on dst_rank in d_packed_all, rank after rank, with
 *                               d_offsets_all [n_channels_total + 1] the GLOBAL offsets (channel c of the node: rows
 *                               d_offsets_all[c] .. d_offsets_all[c+1]); h_totals [world] (host, may be NULL) the
 *                               records per rank.  Three grouped exchanges: each rank's verdict on its own buffer with
 *                               its offset table to dst_rank; dst_rank's verdict over all of them (and over
 *                               packed_cap_all) back to every rank; then, only on "go", sum(counts) x 64 B per rank.
 *                               A buffer too small -- or missing, or a context of another channel count -- ANYWHERE makes
 *                               EVERY rank return M17GPU_ERR_ARG with nothing moved and no transfer left unmatched: what is
 *                               wrong on one rank travels as that rank's "no".  EVERY rank of the communicator must make the
 *                               call, also one whose channel range is empty (world > channels: any context, NULL buffers,
 *                               capacity 0); only a NULL ctx / comm or an impossible rank / world returns on that rank alone.  The sizes come from device memory, so the entry
 *                               synchronises `stream` (three times); it runs behind the step, beside nothing.
 *   m17gpu_shard_set_library    bind the fan-out entries to the library at `path` (ncclGroupStart / ncclGroupEnd /
 *                               ncclSend / ncclRecv with RCCL's signatures) instead of the process's RCCL; NULL = the
 *                               default.  Only before the first fan-out call of the process.
 *   m17gpu_unpack_records       packed rows + offsets back into recs [n_channels][rec_cap] + counts (rows beyond a
 *                               channel's count zeroed): the layout m17gpu_rx_blocks writes, for callers that want it.
 * (The reference has nothing to gather: one channel per process, m17_tx_rx.cpp:28-40.) */
int m17gpu_pack_records(m17gpu_ctx *ctx, const m17gpu_rec *d_recs, int rec_cap, const int32_t *d_counts,
                        m17gpu_rec *d_packed, int packed_cap, int32_t *d_offsets, void *stream);
int m17gpu_unpack_records(m17gpu_ctx *ctx, const m17gpu_rec *d_packed, const int32_t *d_offsets, int n_channels,
                          m17gpu_rec *d_recs, int rec_cap, int32_t *d_counts, void *stream);
int m17gpu_shard_gather_packed(m17gpu_ctx *ctx, void *comm, int rank, int world, int dst_rank,
                               const m17gpu_rec *d_packed_mine, int packed_cap_mine, const int32_t *d_offsets_mine, int n_channels_total,
                               m17gpu_rec *d_packed_all, int packed_cap_all, int32_t *d_offsets_all, int32_t *h_totals, void *stream);
int m17gpu_shard_set_library(const char *path);

/* ---------------- output wire format on the device ----------------
 * m17gpu_set_net_output attaches the sink of decode_stream_frame (m17_rx_parse.cpp:151-154 ->
 * m17_net_new_rx_data m17_net.cpp:53-74) to the context: while d_net != NULL every m17gpu_rx_blocks(mode 1)
 * call writes, for each record flagged M17GPU_F_DELIVERED, the 54-byte frame of m17gpu_format_net_frame into
 *   d_net [C][net_rec_cap][56]   at [channel][index of that record]   (rows of 56 bytes, the frame is the first 54)
 * built by the bookkeeping kernel from (stream id, m_lsf[1] AS IT STOOD AT THAT FRAME, fn, payload).  net_rec_cap is
 * the sink's capacity in records per channel and must equal the rec_cap of every mode-1 m17gpu_rx_blocks call made
 * while the sink is attached: a call with another rec_cap returns M17GPU_ERR_ARG and launches nothing (the sink is
 * indexed with the call's rec_cap; another capacity would be written out of bounds or at the wrong rows).  stream id = d_stream_ids[channel] (0 when NULL) + the channel's frame-id
 * epoch (the event counter that stands in for the reference's rand(), m17_rx_parse.cpp:10-12), mod 2^16.
 * dst_override as in m17gpu_format_net_frame.  Rows of other records are left untouched.  d_net = NULL detaches. */
int m17gpu_set_net_output(m17gpu_ctx *ctx, uint8_t *d_net, int net_rec_cap, const uint16_t *d_stream_ids, uint64_t dst_override);
/* m17gpu_parse_lsf for n LSFs resident on the device: d_lsf [n][30] -> d_out [n] (64-byte structs) */
int m17gpu_parse_lsf_batch(m17gpu_ctx *ctx, const uint8_t *d_lsf, m17gpu_lsf_fields *d_out, int n, void *stream);

/* ---------------- synthetic signal source (host) ----------------
 * A restatement of the reference transmitter (framer m17_tx_routines.cpp:24-255,
 * 4-FSK modulator m17_modulate.cpp:22-86, 10 samples/symbol) used to produce
 * the benchmark / test IQ.  Not on the hot path. */
typedef struct {
    uint64_t seed;            /* per-channel PRNG seed (splitmix64) */
    int32_t  n_stream_frames; /* stream frames per transmission */
    int32_t  delay_samples;   /* un-modulated carrier samples prepended (0..1919) */
    float    ebn0_db;         /* AWGN level; >= 100 means noiseless */
    int32_t  packet_mode;     /* 0 = stream transmissions, 1 = packet-mode bursts */
    float    noise_cutoff_hz; /* 0 = white noise over 48 kHz; >0 = one-sided cutoff of the channel filter on the noise */
} m17gen_params;

/* h_iq [nblk*1920*2]; h_lsf [30] the LSF sent (incl. CRC); h_payload
 * [max_payload_frames][16] the stream payloads in transmit order; returns the
 * number of stream frames fully generated (or <0). */
int m17gen_channel(const m17gen_params *p, int nblk, int16_t *h_iq,
                   uint8_t *h_lsf, uint8_t *h_payload, int max_payload_frames);
/* C channels, channel c uses seed base_seed + c and delay (hash of c) % 1920;
 * h_iq [C][nblk][1920][2]; h_lsf [C][30]; h_payload [C][max_payload_frames][16];
 * h_nframes [C]; nthreads host threads. */
int m17gen_batch(int C, uint64_t base_seed, int first_channel, int nblk, int n_stream_frames,
                 float ebn0_db, float noise_cutoff_hz, int packet_mode, int16_t *h_iq, uint8_t *h_lsf,
                 uint8_t *h_payload, int max_payload_frames, int32_t *h_nframes, int nthreads);
/* transmit-side codec pieces, exposed for round-trip tests */
int m17gen_stream_frame_dibits(const uint8_t lsf[30], int lich_count, uint16_t fn,
                               const uint8_t payload[16], uint8_t dibits[192]);
int m17gen_lsf_frame_dibits(const uint8_t lsf[30], uint8_t dibits[192]);
int m17gen_packet_frame_dibits(const uint8_t *payload, int len, int eof, int nf, uint8_t dibits[192]);
int m17gen_build_lsf(uint64_t dst, uint64_t src, uint16_t type_word, const uint8_t meta[14], uint8_t lsf[30]);
uint64_t m17gen_encode_call(const char *call9);
/* modulate n dibits (value 0..3, or 255 = zero deviation) appending 10 samples each */
int m17gen_modulate(const uint8_t *dibits, int n, int16_t *h_iq, int reset);

/* GPU-side signal source: the signal of m17gen_batch (stream mode) for the
 * context's C channels, made on the device: d_iq [C][nblk][1920][2] int16; optional d_lsf
 * [C][30], d_payload [C][max_payload_frames][16], d_nframes [C].  Restates m17_tx_routines.cpp:24-255
 * and m17_modulate.cpp:22-86 like the host generator; dibits, filter sums and phases are
 * bit-identical to it, cos/sin/log come from the device math library (an IQ sample may differ by
 * one LSB with probability ~1e-12).  Synchronises the stream before returning. */
int m17gpu_gen_batch(m17gpu_ctx *ctx, uint64_t base_seed, int first_channel, int nblk, int n_stream_frames,
                     float ebn0_db, float noise_cutoff_hz, int16_t *d_iq, uint8_t *d_lsf, uint8_t *d_payload,
                     int max_payload_frames, int32_t *d_nframes, void *stream);
/* The same with the generator's stages copied out for verification against a reference transmitter (both optional):
 * d_dibits [C][nblk + 1][192] the frames as dibits (0..3; 255 = the unmodulated carrier of m17_mod_carrier,
 * m17_modulate.cpp:88-92), d_phase [C][nblk * 1920] the modulator's phase accumulator m_acc after every sample
 * (m17_modulate.cpp:24; 0 during the channel's start delay). */
int m17gpu_gen_batch_stages(m17gpu_ctx *ctx, uint64_t base_seed, int first_channel, int nblk, int n_stream_frames,
                            float ebn0_db, float noise_cutoff_hz, int16_t *d_iq, uint8_t *d_lsf, uint8_t *d_payload,
                            int max_payload_frames, int32_t *d_nframes, uint8_t *d_dibits, float *d_phase, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* M17GPU_H */
