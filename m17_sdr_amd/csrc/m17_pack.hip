// m17_pack.hip -- record compaction in front of the multi-GPU gather ("gather of output records, <= 64 B
// x 16,384 ~ 1 MB per GPU").  m17gpu_rx_blocks writes recs[C][rec_cap] with counts[C] valid rows per channel; what has
// to cross xGMI is the valid rows only: counts -> exclusive scan -> packed[sum(counts)][64 B], channel-major, each
// channel's records in event order.  The reference has one channel per process and nothing to gather
// (m17_tx_rx.cpp:28-40); the 64-byte record stands for its sink calls (m17_net_new_rx_data m17_net.cpp:53-74,
// m17_db_golay_errors / m17_aos / m17_los m17_dbase.cpp:60-82).
#pragma clang fp contract(off)

namespace m17dev {

// offs[0] = 0, offs[c + 1] = counts[0] + .. + counts[c] (counts clamped to [0, cap]); ONE workgroup of 1024 lanes walks
// the channels in tiles of 1024: 131,072 channels are 128 tiles of a few hundred cycles each
__global__ __launch_bounds__(1024)
void k_pack_scan(const int32_t *__restrict__ counts, int C, int cap, int32_t *__restrict__ offs)
{
    __shared__ int32_t wsum[16];
    __shared__ int32_t base_s;
    const int tid = (int)threadIdx.x, lane = tid & 63, wv = tid >> 6;
    if (tid == 0) { base_s = 0; offs[0] = 0; }
    __syncthreads();
    for (int c0 = 0; c0 < C; c0 += 1024) {
        const int c = c0 + tid;
        int v = 0;
        if (c < C) { v = counts[c]; v = v < 0 ? 0 : (v > cap ? cap : v); }
        int s = v;                                            // inclusive scan inside the wave
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int o = __shfl_up(s, d, 64);
            if (lane >= d) s += o;
        }
        if (lane == 63) wsum[wv] = s;
        __syncthreads();
        int before = 0;
        for (int k = 0; k < wv; ++k) before += wsum[k];
        const int base = base_s;
        if (c < C) offs[c + 1] = base + before + s;
        __syncthreads();
        if (tid == 1023) base_s = base + before + s;
        __syncthreads();
    }
}

// one 16-byte piece per lane: record i of channel c -> packed[offs[c] + i]
__global__ void k_pack_copy(const uint4 *__restrict__ recs, int C, int cap, const int32_t *__restrict__ offs,
                            uint4 *__restrict__ packed, int packed_cap)
{
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long slot = t >> 2;
    const int q = (int)(t & 3);
    if (slot >= (long long)C * cap) return;
    const int c = (int)(slot / cap), i = (int)(slot - (long long)c * cap);
    const int a = offs[c], n = offs[c + 1] - a;
    if (i >= n || a + i >= packed_cap) return;
    packed[(size_t)(a + i) * 4 + q] = recs[(size_t)slot * 4 + q];
}

// the inverse, on the gathering rank: packed rows back into recs_all[C][cap] (rows beyond a channel's count zeroed)
__global__ void k_unpack_copy(const uint4 *__restrict__ packed, int C, int cap, const int32_t *__restrict__ offs,
                              uint4 *__restrict__ recs, int32_t *__restrict__ counts)
{
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long slot = t >> 2;
    const int q = (int)(t & 3);
    if (slot >= (long long)C * cap) return;
    const int c = (int)(slot / cap), i = (int)(slot - (long long)c * cap);
    const int a = offs[c], n = offs[c + 1] - a;
    recs[(size_t)slot * 4 + q] = i < n ? packed[(size_t)(a + i) * 4 + q] : make_uint4(0, 0, 0, 0);
    if (i == 0 && q == 0) counts[c] = n;
}

// offs_all[lo + k] += shift for k = 1 .. n (the rank's local offsets moved behind the ranks before it)
__global__ void k_offs_shift(int32_t *__restrict__ offs, int n, int shift)
{
    const int i = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (i < n) offs[i + 1] += shift;
}

} // namespace m17dev
