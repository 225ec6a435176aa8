// AddressSanitizer / UBSan driver for the CPU-side code (tests/test_sanitizers.py builds and runs it):
// the host signal source (m17_txgen.cpp), the table builders (m17_tables.cpp) and the oracle's whole
// receive chain on what the generator makes, including noise, packet mode and hostile input.
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <vector>
#include <cstdint>
extern "C" {
#include "../../oracle/m17_oracle.h"
}
#include "../../include/m17gpu.h"

int main()
{
    m17o_init();
    const int C = 6, nblk = 14, cap = 2 * nblk + 2;
    std::vector<int16_t> iq((size_t)C * nblk * 1920 * 2);
    std::vector<uint8_t> lsf(C * 30), pl((size_t)C * (nblk + 2) * 16);
    std::vector<int32_t> nf(C);
    long total = 0;
    for (int pass = 0; pass < 4; ++pass) {
        const float eb = (pass == 1) ? 9.0f : (pass == 2 ? 6.0f : 200.0f);
        if (m17gen_batch(C, 0x4D313700ull + pass, 3, nblk, 9, eb, pass == 2 ? 6250.0f : 0.0f, pass == 3, iq.data(), lsf.data(),
                         pl.data(), nblk + 2, nf.data(), 2) != 0) { std::puts("gen failed"); return 2; }
        if (pass == 0) {                                   // hostile: squelch gap, saturation, isolated zeros
            std::memset(&iq[(size_t)1 * nblk * 3840 + 3 * 3840], 0, 3840 * 2 * sizeof(int16_t));
            for (size_t i = (size_t)2 * nblk * 3840; i < (size_t)3 * nblk * 3840; ++i) iq[i] = (i & 1) ? -32768 : 32767;
        }
        std::vector<uint8_t> st((size_t)C * m17o_sizeof_chan());
        for (int c = 0; c < C; ++c) m17o_chan_reset(reinterpret_cast<m17o_chan *>(&st[(size_t)c * m17o_sizeof_chan()]));
        std::vector<m17o_rec> recs((size_t)C * cap);
        std::vector<int32_t> counts(C), nsyms((size_t)C * nblk);
        std::vector<float> syms((size_t)C * (nblk * 193 + 8));
        m17o_rx_blocks(reinterpret_cast<m17o_chan *>(st.data()), C, nblk, iq.data(), recs.data(), cap, counts.data(),
                       syms.data(), nsyms.data(), 1, 2);
        for (int c = 0; c < C; ++c) total += counts[c];
    }
    int16_t hist[62] = {0}, wide[2 * 256], out[2 * 32];
    for (int i = 0; i < 512; ++i) wide[i] = (int16_t)(i * 37 - 9000);
    m17o_pluto_decimate(hist, wide, 256, out);
    uint8_t frame[54], l30[30] = {0}, p16[16] = {0};
    m17gpu_format_net_frame(0x1234, l30, 7, p16, 0, frame);
    std::printf("sanitizer driver ok: %ld records\n", total);
    return total > 0 ? 0 : 3;
}
