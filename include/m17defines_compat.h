/*
 * m17defines_compat.h -- C++-linkage declarations of the functions that the
 * reference's m17_rx_frame.cpp / m17_rx_parse.cpp call DOWN into, and of the
 * block entry point that calls them, with exactly the reference signatures
 * (m17gismo/m17defines.h, tag v1; line numbers below).  m17_compat.cpp
 * implements them on top of the C-ABI core (include/m17gpu.h) for a hidden
 * "channel 0", so those two reference translation units link unchanged.
 *
 * The reference has no extern "C" anywhere; overloads and a C++ reference
 * parameter are part of its interface, hence a C++ header.
 */
#ifndef M17DEFINES_COMPAT_H
#define M17DEFINES_COMPAT_H
#include <stdint.h>

typedef uint16_t uint12_t;     /* m17defines.h:20 */
typedef uint32_t uint24_t;     /* m17defines.h:21 */
typedef uint64_t uint48_t;     /* m17defines.h:22 */

typedef struct { uint8_t p_s, dt, et, est, can, reserved; } M17Type;   /* m17defines.h:34-41 */
typedef struct { int16_t re, im; } scmplx;                             /* m17defines.h:130-133 */

/* init entry points, call order main.cpp:110-118 */
void m17_crc_init(void);                                  /* m17defines.h:321 */
void m17_init_conv(void);                                 /* :299 */
void m17_init_de_correlate(void);                         /* :330 */
void m17_dsp_init(void);                                  /* :224 */
void m17_golay_init(void);                                /* :315 */
void m17_rx_sync_init(void);                              /* :374 */

/* called by m17_rx_parse.cpp */
void     m17_dsp_demap_frame(float *in, float *out);      /* :244  GPU */
void     m17_de_correlate_1(float *in, float *out, int len);   /* :329 */
void     m17_de_interleave(float *in, float *out, int len);    /* :308 */
int      m17_de_punc_p1(float *in, float *out, int len);  /* :262 */
int      m17_de_punc_p2(float *in, float *out, int len);  /* :263 */
int      m17_de_punc_p3(float *in, float *out, int len);  /* :264 */
int      m17_viterbi_decode(float *in, uint8_t *out, int len); /* :302  GPU */
uint24_t hard_decode_24_bits(float *in);                  /* :289 */
int      m_17_golay_decode(uint24_t word, uint12_t &odata);    /* :314  GPU */
int      pack_1_to_8(uint8_t *in, uint8_t *out, int len); /* :275 */
int      pack_12_to_8_x4x6(uint12_t *in, uint8_t *out);   /* :282 */
uint48_t pack_8_to_48(uint8_t *in);                       /* :283 */
uint16_t pack_8_to_16(uint8_t *in);                       /* :285 */
uint16_t m17_crc_array_encode(uint8_t *in, int len);      /* :320 */
M17Type  m17_upack_type(uint16_t word);                   /* :294 */

/* called by m17_tx_rx.cpp; calls m17_rx_symbols() of the unchanged m17_rx_frame.cpp
 * and reads its m17_rx_lock() */
void m17_dsp_rx(scmplx *in, int len);                     /* :228  GPU front end */
int  m17_rx_sync_samples(float *in, float *out, int len); /* :373  GPU timing recovery */

/* provided by the unchanged reference m17_rx_frame.cpp */
void m17_rx_symbols(float *sym, int len);                 /* :362 */
bool m17_rx_lock(void);                                   /* :365 */

#endif
