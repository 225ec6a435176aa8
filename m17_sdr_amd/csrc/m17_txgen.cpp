// m17_txgen.cpp -- synthetic M17 signal source (host).
//
// Restates the reference transmitter so that benchmarks and tests have a
// bit-faithful M17 waveform to receive: frame builders of
// m17_tx_routines.cpp:24-255 and the 4-FSK RRC modulator of
// m17_modulate.cpp:22-86 at 10 samples/symbol (radio.cpp:212-214, Lime).
// Adds what the reference has no need for: per-channel seeding, start delay and
// AWGN.  Compiled with -ffp-contract=off.
#include "m17_host.h"
#include "../../include/m17gpu.h"
#include <cmath>
#include <cstring>
#include <thread>
#include <vector>
#include <algorithm>

namespace {

using m17::Tables;
using m17::tables;

constexpr int kOs = 10;              // samples per symbol
constexpr int kTxTaps = 31;          // m17_modulate.cpp:6 TX_FN

struct SplitMix64 {
    uint64_t s;
    explicit SplitMix64(uint64_t seed) : s(seed) {}
    uint64_t next() {
        uint64_t z = (s += 0x9E3779B97F4A7C15ull);
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        return z ^ (z >> 31);
    }
    double uniform() { return ((next() >> 11) + 0.5) * (1.0 / 9007199254740992.0); }
    void gauss2(double &a, double &b) {            // Box-Muller
        const double u = uniform(), v = uniform();
        const double r = std::sqrt(-2.0 * std::log(u));
        a = r * std::cos(2.0 * M_PI * v);
        b = r * std::sin(2.0 * M_PI * v);
    }
};

// ---- bit plumbing of the transmit framer --------------------------------
// m17_conv.cpp:53-71: shift register with the new bit at 0x10, 4 flush steps
int conv_encode_bytes(const uint8_t *in, int nbytes, uint8_t *out)
{
    const Tables &T = tables();
    int n = 0; unsigned sr = 0;
    for (int i = 0; i < nbytes; ++i)
        for (int m = 0x80; m; m >>= 1) {
            if (in[i] & m) sr |= 0x10;
            out[n++] = T.clut[sr][0];
            out[n++] = T.clut[sr][1];
            sr >>= 1;
        }
    for (int i = 0; i < 4; ++i) {
        out[n++] = T.clut[sr][0];
        out[n++] = T.clut[sr][1];
        sr >>= 1;
    }
    return n;
}

int puncture(int type, const uint8_t *in, int len, uint8_t *out)
{
    int n = 0;
    for (int k = 0; k < len; ++k) {
        bool keep = (type == 1) ? ((k % 61) % 4 != 2) : (type == 2) ? (k % 12 != 11) : (k % 8 != 7);
        if (keep) out[n++] = in[k];
    }
    return n;
}

// interleave (m17_interleave.cpp:3-7) then randomise (m17_correlate.cpp:16-20)
// then pack to dibits behind the 16-bit sync word (m17_bit_utils.cpp:74-85,19-25)
void finish_frame(uint16_t sync, const uint8_t bits[368], uint8_t dibits[192])
{
    const Tables &T = tables();
    uint8_t il[368];
    for (int i = 0; i < 368; ++i) il[T.interleave[i]] = bits[i];
    for (int i = 0; i < 368; ++i) il[i] = (uint8_t)((il[i] ^ T.derand[i]) & 1);
    for (int i = 0; i < 8; ++i) dibits[i] = (uint8_t)((sync >> (14 - 2 * i)) & 3);
    for (int i = 0; i < 184; ++i) dibits[8 + i] = (uint8_t)((il[2 * i] << 1) | il[2 * i + 1]);
}

// ---- modulator state (m17_modulate.cpp:6-16) -----------------------------
struct Modulator {
    float taps[kTxTaps * kOs];
    float hist[kTxTaps];
    float acc;
    float lut[4];
    Modulator() { reset(); }
    void reset() {
        m17::build_rrc(taps, 0.5f, kTxTaps * kOs, kOs);          // m17_modulate.cpp:73
        m17::set_filter_gain(taps, 10, 1, kTxTaps * kOs);        // :74
        std::memset(hist, 0, sizeof hist);
        acc = 0;
        m17::tx_deviation_lut(lut);                              // :9
    }
    // one symbol in, 10 IQ samples out (mod_filter :49-61 + mod_fsk :22-38)
    void symbol(float dev, int16_t *out) {
        for (int i = 0; i < kTxTaps - 1; ++i) hist[i] = hist[i + 1];
        hist[kTxTaps - 1] = dev;
        for (int i = 0, n = kOs - 1; i < kOs; ++i, --n) {
            const float *c = &taps[n];
            float sum = hist[0] * c[0];
            for (int j = 1; j < kTxTaps; ++j) sum += hist[j] * c[j * kOs];
            acc += sum;
            // cos(m_acc) with a float argument under <math.h> in C++ resolves to the float overload,
            // and float * int stays float (m17_modulate.cpp:25-26): single precision throughout
            out[2 * i]     = (int16_t)(cosf(acc) * (float)0x3FFF);
            out[2 * i + 1] = (int16_t)(sinf(acc) * (float)0x3FFF);
        }
        acc = (float)(acc / (2.0 * M_PI));       // phase wrap, :33-37
        double ip;
        acc = (float)std::modf((double)acc, &ip);
        acc = (float)(acc * 2.0 * M_PI);
    }
};

struct Transmitter {
    Modulator mod;
    std::vector<int16_t> &iq;      // interleaved I,Q
    size_t limit;                  // samples wanted
    explicit Transmitter(std::vector<int16_t> &dst, size_t lim) : iq(dst), limit(lim) {}
    bool full() const { return iq.size() / 2 >= limit; }
    void send_dev(float dev) {
        int16_t s[2 * kOs];
        mod.symbol(dev, s);
        iq.insert(iq.end(), s, s + 2 * kOs);
    }
    void send_dibits(const uint8_t *d, int n) { for (int i = 0; i < n; ++i) send_dev(mod.lut[d[i] & 3]); }
    void carrier(int nsym) { for (int i = 0; i < nsym; ++i) send_dev(0.0f); }   // m17_mod_carrier :88-92
    void preamble() {                                                           // m17_tx_routines.cpp:24-31
        uint8_t d[192];
        for (int i = 0; i < 96; ++i) { d[2 * i] = 1; d[2 * i + 1] = 3; }
        send_dibits(d, 192);
    }
    void eot() {                                                                // m17_tx_routines.cpp:242-255
        static const uint8_t pat[8] = {1, 1, 1, 1, 1, 1, 3, 1};
        uint8_t d[192];
        for (int i = 0; i < 192; ++i) d[i] = pat[i & 7];
        send_dibits(d, 192);
    }
};

void random_call(SplitMix64 &rng, char call[10])
{
    static const char alpha[] = "ABCDEFGHIJKLMNOPQRSTUVWXYZ0123456789";
    const int n = 4 + (int)(rng.next() % 3);
    for (int i = 0; i < 9; ++i) call[i] = (i < n) ? alpha[rng.next() % 36] : ' ';
    call[9] = 0;
}

// AWGN of complex variance 2*sigma^2 per 48 kHz sample (PSD N0), optionally band-limited
// by a unity-gain windowed-sinc low-pass of one-sided cutoff `cutoff_hz` -- the channel
// filter a radio front end has ahead of the ADC (the reference itself has none in software,
// m17_dsp.cpp:461-476).  0 = white over the whole 48 kHz.
void add_awgn(std::vector<int16_t> &iq, double sigma, SplitMix64 &rng, double cutoff_hz)
{
    const size_t n = iq.size() / 2;
    std::vector<double> nr(n), ni(n);
    for (size_t i = 0; i < n; ++i) rng.gauss2(nr[i], ni[i]);
    if (cutoff_hz > 0.0) {
        constexpr int L = 63;
        double h[L], hs = 0.0;
        const double fc = cutoff_hz / 48000.0;
        for (int k = 0; k < L; ++k) {
            const int m = k - L / 2;
            const double sinc = (m == 0) ? 2.0 * fc : std::sin(2.0 * M_PI * fc * m) / (M_PI * m);
            h[k] = sinc * (0.54 - 0.46 * std::cos(2.0 * M_PI * k / (L - 1)));
            hs += h[k];
        }
        for (int k = 0; k < L; ++k) h[k] /= hs;
        std::vector<double> fr(n), fi(n);
        for (size_t i = 0; i < n; ++i) {
            double ar = 0.0, ai = 0.0;
            for (int k = 0; k < L; ++k) {
                const long j = (long)i + k - L / 2;
                if (j >= 0 && j < (long)n) { ar += h[k] * nr[(size_t)j]; ai += h[k] * ni[(size_t)j]; }
            }
            fr[i] = ar; fi[i] = ai;
        }
        nr.swap(fr); ni.swap(fi);
    }
    for (size_t i = 0; i + 1 < iq.size(); i += 2) {
        const double a = nr[i / 2], b = ni[i / 2];
        long re = std::lrint((double)iq[i] + sigma * a);
        long im = std::lrint((double)iq[i + 1] + sigma * b);
        re = std::min(32767l, std::max(-32767l, re));
        im = std::min(32767l, std::max(-32767l, im));
        if (re == 0 && im == 0) re = 1;          // the limiter divides by |z|
        iq[i] = (int16_t)re; iq[i + 1] = (int16_t)im;
    }
}

} // namespace

extern "C" {

uint64_t m17gen_encode_call(const char *call)
{
    // base-40, first character least significant (m17_bit_utils.cpp:191-208)
    uint64_t w = 0;
    for (int i = 8; i >= 0; --i) {
        const char ch = call[i];
        w *= 40;
        if (ch >= 'A' && ch <= 'Z') w += (uint64_t)(ch - 'A' + 1);
        else if (ch >= '0' && ch <= '9') w += (uint64_t)(ch - '0' + 27);
        else if (ch == '-') w += 37;
        else if (ch == '/') w += 38;
        else if (ch == '.') w += 39;
    }
    return w;
}

// build_lich, m17_tx_routines.cpp:38-54
int m17gen_build_lsf(uint64_t dst, uint64_t src, uint16_t type_word, const uint8_t meta[14], uint8_t lsf[30])
{
    for (int i = 0; i < 6; ++i) lsf[i] = (uint8_t)(dst >> (40 - 8 * i));
    for (int i = 0; i < 6; ++i) lsf[6 + i] = (uint8_t)(src >> (40 - 8 * i));
    lsf[12] = (uint8_t)(type_word >> 8); lsf[13] = (uint8_t)type_word;
    std::memcpy(&lsf[14], meta, 14);
    const uint16_t crc = m17::crc16(lsf, 28);
    lsf[28] = (uint8_t)(crc >> 8); lsf[29] = (uint8_t)crc;
    return 30;
}

// m17_fmt_add_link_setup_frame, m17_tx_routines.cpp:92-117
int m17gen_lsf_frame_dibits(const uint8_t lsf[30], uint8_t dibits[192])
{
    uint8_t coded[488], bits[368];
    const int n = conv_encode_bytes(lsf, 30, coded);
    if (puncture(1, coded, n, bits) != 368) return -1;
    finish_frame(m17::kSyncLinkSetup, bits, dibits);
    return 192;
}

// m17_fmt_add_stream_frame, m17_tx_routines.cpp:143-187
int m17gen_stream_frame_dibits(const uint8_t lsf[30], int lich_count, uint16_t fn,
                               const uint8_t payload[16], uint8_t dibits[192])
{
    const Tables &T = tables();
    uint8_t chunk[6], bits[368], coded[296], body[18];
    std::memcpy(chunk, &lsf[(lich_count % 6) * 5], 5);
    chunk[5] = (uint8_t)((lich_count & 7) << 5);
    const uint16_t w[4] = {                                      // pack_8_to_12_x4
        (uint16_t)((chunk[0] << 4) | (chunk[1] >> 4)), (uint16_t)(((chunk[1] & 0xF) << 8) | chunk[2]),
        (uint16_t)((chunk[3] << 4) | (chunk[4] >> 4)), (uint16_t)(((chunk[4] & 0xF) << 8) | chunk[5]) };
    int n = 0;
    for (int k = 0; k < 4; ++k) {
        const uint32_t cw = ((uint32_t)w[k] << 12) | T.golay_enc[w[k]];   // m17_golay_encode
        for (int b = 23; b >= 0; --b) bits[n++] = (uint8_t)((cw >> b) & 1);
    }
    body[0] = (uint8_t)(fn >> 8); body[1] = (uint8_t)fn;
    std::memcpy(&body[2], payload, 16);
    const int nc = conv_encode_bytes(body, 18, coded);
    if (puncture(2, coded, nc, &bits[n]) != 272) return -1;
    finish_frame(m17::kSyncStream, bits, dibits);
    return 192;
}

// m17_fmt_add_packet, m17_tx_routines.cpp:201-222 (only the first 420 coded bits are used)
int m17gen_packet_frame_dibits(const uint8_t *payload, int len, int eof, int nf, uint8_t dibits[192])
{
    if (len > 25 || len < 0) return -1;
    uint8_t tmp[26], coded[424], bits[368];
    std::memset(tmp, 0, sizeof tmp);
    std::memcpy(tmp, payload, (size_t)len);
    tmp[25] = (uint8_t)((eof ? 0x80 : 0x00) | ((nf & 0x1F) << 2));
    conv_encode_bytes(tmp, 26, coded);
    if (puncture(3, coded, 420, bits) != 368) return -1;
    finish_frame(m17::kSyncPacket, bits, dibits);
    return 192;
}

// M17-over-IP stream frame as the reference's reflector client emits it (m17_net.cpp:25-49,
// 53-74): "M17 " | stream id | LSF bytes 0..27 (no CRC) | FN | 16 payload bytes | CRC-16 of
// the first 52 bytes.  dst_override != 0 replaces the 6 destination bytes (the reference
// writes the reflector's own callsign there, m17_net.cpp:60-62).  Unlike the reference
// (memcpy of 54 bytes from a 30-byte array, :58) nothing is read past lsf[27].
int m17gpu_format_net_frame(uint16_t stream_id, const uint8_t lsf[30], uint16_t fn, const uint8_t payload[16],
                            uint64_t dst_override, uint8_t out[54])
{
    if (!lsf || !payload || !out) return M17GPU_ERR_ARG;
    out[0] = 0x4D; out[1] = 0x31; out[2] = 0x37; out[3] = 0x20;
    out[4] = (uint8_t)(stream_id >> 8); out[5] = (uint8_t)stream_id;
    std::memcpy(&out[6], lsf, 28);
    if (dst_override)
        for (int i = 0; i < 6; ++i) out[6 + i] = (uint8_t)(dst_override >> (40 - 8 * i));
    out[34] = (uint8_t)(fn >> 8); out[35] = (uint8_t)fn;
    std::memcpy(&out[36], payload, 16);
    const uint16_t crc = m17::crc16(out, 52);
    out[52] = (uint8_t)(crc >> 8); out[53] = (uint8_t)crc;
    return 54;
}

static void decode_call(uint64_t word, char call[10])          // m17_bit_utils.cpp:209-226
{
    if (word == 0xFFFFFFFFFFFFull) { std::memcpy(call, "BROADCAST", 10); return; }
    for (int i = 0; i < 9; ++i) {
        const int ch = (int)(word % 40);
        char o = ' ';
        if (ch >= 1 && ch <= 26) o = (char)(ch + 'A' - 1);
        else if (ch >= 27 && ch <= 36) o = (char)(ch + '0' - 27);
        else if (ch == 37) o = '-';
        else if (ch == 38) o = '/';
        else if (ch == 39) o = '.';
        call[i] = o;
        word /= 40;
    }
    call[9] = 0;
}

int m17gpu_parse_lsf(const uint8_t lsf[30], m17gpu_lsf_fields *out)
{
    if (!lsf || !out) return M17GPU_ERR_ARG;
    std::memset(out, 0, sizeof *out);
    for (int i = 0; i < 6; ++i) { out->dst = (out->dst << 8) | lsf[i]; out->src = (out->src << 8) | lsf[6 + i]; }   // pack_8_to_48
    decode_call(out->dst, out->dst_call);
    decode_call(out->src, out->src_call);
    const unsigned tw = ((unsigned)lsf[12] << 8) | lsf[13];                 // pack_8_to_16, m17_upack_type
    out->reserved = (uint8_t)((tw >> 11) & 0x1F);
    out->can = (uint8_t)((tw >> 7) & 0xF);
    out->est = (uint8_t)((tw >> 5) & 0x3);
    out->et = (uint8_t)((tw >> 3) & 0x3);
    out->dt = (uint8_t)((tw >> 1) & 0x3);
    out->p_s = (uint8_t)(tw & 1);
    std::memcpy(out->meta, &lsf[14], 14);
    out->crc = (uint16_t)(((unsigned)lsf[28] << 8) | lsf[29]);
    out->crc_ok = m17::crc16(lsf, 30) == 0;
    return M17GPU_OK;
}

static thread_local Modulator g_mod;

int m17gen_modulate(const uint8_t *dibits, int n, int16_t *h_iq, int reset)
{
    if (reset) g_mod.reset();
    for (int i = 0; i < n; ++i) {
        const float dev = (dibits[i] == 255) ? 0.0f : g_mod.lut[dibits[i] & 3];
        g_mod.symbol(dev, h_iq + (size_t)i * 2 * kOs);
    }
    return n * kOs;
}

int m17gen_channel(const m17gen_params *p, int nblk, int16_t *h_iq,
                   uint8_t *h_lsf, uint8_t *h_payload, int max_payload_frames)
{
    if (!p || !h_iq || nblk <= 0) return M17GPU_ERR_ARG;
    const size_t want = (size_t)nblk * m17::kBlockSamples;
    SplitMix64 rng(p->seed);
    std::vector<int16_t> iq;
    iq.reserve(2 * (want + 4096));
    Transmitter tx(iq, want);

    // station identity: broadcast destination, random source, voice stream type
    char call[10];
    random_call(rng, call);
    uint8_t meta[14] = {0}, lsf[30];
    const uint16_t type_word = p->packet_mode ? 0x0002 /* packet, data */ : 0x0005 /* stream, voice */;
    m17gen_build_lsf(0xFFFFFFFFFFFFull, m17gen_encode_call(call), type_word, meta, lsf);
    if (h_lsf) std::memcpy(h_lsf, lsf, 30);

    int delay = p->delay_samples;
    if (delay < 0) delay = 0;
    // un-modulated carrier ahead of the first transmission: cos/sin of a zero
    // phase accumulator, like m17_mod_carrier before any symbol was sent
    for (int i = 0; i < delay; ++i) { iq.push_back(0x3FFF); iq.push_back(0); }

    int sent = 0;
    uint8_t d[192];
    while (!tx.full()) {
        // m17_tx_rx.cpp:95-98 : carrier, two preambles, link setup
        tx.carrier(192);
        tx.preamble();
        tx.preamble();
        m17gen_lsf_frame_dibits(lsf, d);
        tx.send_dibits(d, 192);
        if (!p->packet_mode) {
            for (int f = 0; f < p->n_stream_frames && !tx.full(); ++f) {
                uint8_t pld[16];
                for (int i = 0; i < 16; i += 8) {
                    uint64_t r = rng.next();
                    std::memcpy(&pld[i], &r, 8);
                }
                m17gen_stream_frame_dibits(lsf, f % 6, (uint16_t)f, pld, d);
                tx.send_dibits(d, 192);
                if (h_payload && sent < max_payload_frames)
                    std::memcpy(h_payload + (size_t)sent * 16, pld, 16);
                ++sent;
            }
        } else {
            // m17_send_packet_frames, m17_tx_routines.cpp:324-350
            uint8_t pkt[128];
            int len = 20 + (int)(rng.next() % 70);
            for (int i = 0; i < len; ++i) pkt[i] = (uint8_t)rng.next();
            const uint16_t crc = m17::crc16(pkt, len);
            pkt[len] = (uint8_t)(crc >> 8); pkt[len + 1] = (uint8_t)crc; len += 2;
            const int frames = len / 25, leftover = len % 25;
            if (leftover == 0) {
                for (int i = 0; i < frames - 1; ++i) { m17gen_packet_frame_dibits(&pkt[i * 25], 25, 0, i, d); tx.send_dibits(d, 192); }
                m17gen_packet_frame_dibits(&pkt[(frames - 1) * 25], 25, 1, 25, d); tx.send_dibits(d, 192);
            } else {
                for (int i = 0; i < frames; ++i) { m17gen_packet_frame_dibits(&pkt[i * 25], 25, 0, i, d); tx.send_dibits(d, 192); }
                m17gen_packet_frame_dibits(&pkt[frames * 25], leftover, 1, leftover, d); tx.send_dibits(d, 192);
            }
            ++sent;
        }
        tx.eot();
    }
    iq.resize(2 * want);
    if (p->ebn0_db < 100.0f) {
        // Es = A^2 * sps (A = 0x3FFF), Es/N0 = 2 Eb/N0 (two channel bits per
        // symbol), complex noise variance per sample = N0
        const double esn0 = 2.0 * std::pow(10.0, p->ebn0_db / 10.0);
        const double a = 16383.0;
        const double sigma = std::sqrt(a * a * kOs / (2.0 * esn0));
        SplitMix64 nrng(p->seed ^ 0xA36E0000A36E0000ull);
        add_awgn(iq, sigma, nrng, (double)p->noise_cutoff_hz);
    }
    std::memcpy(h_iq, iq.data(), sizeof(int16_t) * 2 * want);
    return sent;
}

int m17gen_batch(int C, uint64_t base_seed, int first_channel, int nblk, int n_stream_frames,
                 float ebn0_db, float noise_cutoff_hz, int packet_mode, int16_t *h_iq, uint8_t *h_lsf,
                 uint8_t *h_payload, int max_payload_frames, int32_t *h_nframes, int nthreads)
{
    if (C <= 0 || nblk <= 0 || !h_iq) return M17GPU_ERR_ARG;
    if (nthreads < 1) nthreads = 1;
    nthreads = std::min(nthreads, C);
    std::vector<std::thread> pool;
    std::vector<int> status((size_t)nthreads, 0);
    const size_t chan_stride = (size_t)nblk * m17::kBlockSamples * 2;
    for (int t = 0; t < nthreads; ++t) {
        pool.emplace_back([=, &status] {
            for (int c = t; c < C; c += nthreads) {
                const uint64_t ch = (uint64_t)(first_channel + c);
                m17gen_params p;
                p.seed = base_seed + ch;
                p.n_stream_frames = n_stream_frames;
                SplitMix64 h(0xD1B54A32D192ED03ull ^ (ch * 0x9E3779B97F4A7C15ull));
                p.delay_samples = (int)(h.next() % 1920);
                p.ebn0_db = ebn0_db;
                p.packet_mode = packet_mode;
                p.noise_cutoff_hz = noise_cutoff_hz;
                int r = m17gen_channel(&p, nblk, h_iq + (size_t)c * chan_stride,
                                       h_lsf ? h_lsf + (size_t)c * 30 : nullptr,
                                       h_payload ? h_payload + (size_t)c * max_payload_frames * 16 : nullptr,
                                       max_payload_frames);
                if (r < 0) status[(size_t)t] = r;
                else if (h_nframes) h_nframes[c] = r;
            }
        });
    }
    for (auto &th : pool) th.join();
    for (int s : status) if (s < 0) return s;
    return 0;
}

} // extern "C"
