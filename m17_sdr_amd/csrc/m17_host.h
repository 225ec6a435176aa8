// m17_host.h -- host-side tables shared by the HIP core, the signal source and
// the C++ compatibility shim.  Product code (no oracle dependency).
#pragma once
#include <cstdint>
#include <cstddef>

#define M17_PI 3.14159265358979323846   /* M_PI of <math.h> */

namespace m17 {

constexpr int kBlockSamples = 1920;   // m17defines.h:17
constexpr int kDiscOut      = 384;    // m17_dsp.cpp:463
constexpr int kFrameSyms    = 192;    // m17defines.h:66
constexpr int kSoftBits     = 368;
constexpr int kPhases       = 40;     // m17_rx_sync.cpp:3  NF
constexpr int kTaps         = 31;     // m17_rx_sync.cpp:4  FN

// Transmit-side literals of the signal sources (m17_tx_routines.cpp:6-9, m17_modulate.cpp:9)
constexpr uint16_t kSyncLinkSetup = 0x55F7, kSyncStream = 0xFF5D, kSyncPacket = 0x75FF, kSyncBert = 0xDF55;
constexpr uint16_t kCrcPoly = 0x5935;  // m17_crc.cpp:4
inline void tx_deviation_lut(float lut[4])     // phase step per sample of dibits 0..3
{
    lut[0] = (float)(M17_PI / 30.0); lut[1] = (float)(M17_PI / 10.0);
    lut[2] = (float)(-M17_PI / 30);  lut[3] = (float)(-M17_PI / 10.0);
}

// Everything the reference builds once in main.cpp:110-118.
struct Tables {
    uint16_t crc[256];                 // m17_crc.cpp:8-22
    uint8_t  derand[kSoftBits];        // m17_correlate.cpp:35-42
    uint16_t interleave[kSoftBits];    // m17_interleave.cpp:8-12  dst index of src i
    uint16_t golay_enc[4096];          // m17_golay.cpp:31-40
    uint16_t golay_err[4096];          // m17_golay.cpp:49-72
    float    mf[kPhases][kTaps];       // m17_rx_sync.cpp:114-122 (gain-normalised)
    float    md[kPhases][kTaps];       // m17_rx_sync.cpp:109-119 (central difference)
    // Per frame type (1 LSF/P1, 2 stream/P2, 3 packet/P3): for each de-punctured
    // position k the index into the 368 demapped soft bits it is fed from, or
    // -1 for an erasure, with the de-randomiser sign folded in.  This is the
    // composition of m17_de_correlate_1 . m17_de_interleave . m17_de_punc_pN
    // (m17_rx_parse.cpp:90-94, :115-135, :165-169).
    int16_t  gather[4][488];           // [type][k] -> source soft-bit index or -1
    int8_t   gsign[4][488];            // +1 / -1
    int16_t  glen[4];                  // 0, 488, 296, 420
    // Golay part of a stream frame: de-interleaved position j (0..95) <- source
    int16_t  lich_src[96];
    int8_t   lich_sign[96];
    // Viterbi branch tables (m17_conv.cpp:93-108): for new state v the metric
    // index used from the even / odd predecessor.
    uint8_t  bm_even[16], bm_odd[16];
    uint8_t  clut[32][2];              // m17_conv.cpp:24-29
};

const Tables &tables();                // built on first use, thread-safe

void build_rrc(float *f, float rolloff, int ntaps, int sps);        // m17_dsp.cpp:295-315
void set_filter_gain(float *f, float gain, int stride, int ntaps);  // m17_dsp.cpp:420-429
uint16_t crc16(const uint8_t *p, int n);                           // m17_crc.cpp:26-35
void build_pluto_dec_filter(int16_t *coffs /* [31] */);          // radio.cpp:45-51
int puncture_keep(int type, int k);    // P1 / P2 / P3 keep flag of coded bit k (m17_puncture.cpp:4-10)

} // namespace m17
