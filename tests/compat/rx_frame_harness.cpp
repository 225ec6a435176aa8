// rx_frame_harness.cpp -- TEST INFRASTRUCTURE: stands where the reference's m17_rx_frame.cpp stands
// when it is linked against libm17compat.so.  It defines the two functions that translation unit
// provides -- m17_rx_symbols(float*,int) (m17_rx_frame.cpp:173-177), here capturing every symbol it
// is handed, and m17_rx_lock() (m17_rx_frame.cpp:187-189), here returning a scripted flag -- and
// drives the shim's block entry exactly as m17_tx_rx.cpp:164-165 does: m17_dsp_rx(samples, 1920)
// once per 40 ms block, after the init calls of main.cpp:110-118.
// Built as a shared object (tests load it into the test process; see tests/test_gpu_compat.py).
#include "../../include/m17defines_compat.h"
#include <vector>
#include <cstring>

namespace {
std::vector<float> g_syms;
std::vector<int>   g_counts;
int g_block = 0, g_lock_from = 1 << 30, g_lock_until = 1 << 30;
}

void m17_rx_symbols(float *sym, int len)
{
    g_counts.push_back(len);
    g_syms.insert(g_syms.end(), sym, sym + len);
}

bool m17_rx_lock(void) { return g_block >= g_lock_from && g_block < g_lock_until; }

// iq: nblk x 1920 scmplx; lock flag reads true for blocks [lock_from, lock_until).
// out_syms: capacity floats; out_counts: nblk ints.  Returns the number of symbols captured, -1 on overflow.
extern "C" int harness_run(const int16_t *iq, int nblk, int lock_from, int lock_until,
                           float *out_syms, int capacity, int *out_counts, int do_init)
{
    if (do_init) {
        m17_crc_init(); m17_init_conv(); m17_init_de_correlate(); m17_dsp_init();
        m17_golay_init(); m17_rx_sync_init();              // main.cpp:110-118 order
    }
    g_syms.clear(); g_counts.clear();
    g_lock_from = lock_from; g_lock_until = lock_until;
    for (int b = 0; b < nblk; ++b) {
        g_block = b;
        scmplx *blk = reinterpret_cast<scmplx *>(const_cast<int16_t *>(iq)) + (size_t)b * 1920;
        m17_dsp_rx(blk, 1920);                              // m17_tx_rx.cpp:164-165
    }
    g_lock_from = g_lock_until = 1 << 30;
    if ((int)g_counts.size() != nblk || (int)g_syms.size() > capacity) return -1;
    std::memcpy(out_syms, g_syms.data(), g_syms.size() * sizeof(float));
    for (int b = 0; b < nblk; ++b) out_counts[b] = g_counts[b];
    return (int)g_syms.size();
}
