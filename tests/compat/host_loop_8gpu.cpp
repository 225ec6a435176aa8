// host_loop_8gpu.cpp -- the C++ host loop of INTEGRATION.md section B.2, kept compilable: one process (or thread) per GPU,
// the application's own ncclComm_t, the IQ fanned out from rank 0 and the packed records gathered back, with the scatter of
// the next step double-buffered on a second stream.  Compiled (not run) by tests/test_capi_and_shard.py against
// include/m17gpu.h alone -- the three RCCL / HIP names it needs are declared here the way <rccl/rccl.h> and
// <hip/hip_runtime_api.h> declare them, so the file also documents exactly what a host has to bring.
#include <cstdint>
#include <cstddef>
#include "m17gpu.h"

typedef struct ncclComm *ncclComm_t;                     // <rccl/rccl.h>
typedef struct ihipStream_t *hipStream_t;                // <hip/hip_runtime_api.h>
typedef struct ihipEvent_t *hipEvent_t;
extern "C" {
int hipStreamSynchronize(hipStream_t);
int hipEventRecord(hipEvent_t, hipStream_t);
int hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned);
}

struct RankBuffers {                                     // device memory, allocated by the host once
    int16_t *d_iq_all;                                   // rank 0 only: [TOTAL][NBLK][1920][2] of one step (the channeliser's output)
    int16_t *d_iq_mine[2];                               // [hi-lo][NBLK][1920][2], double-buffered
    m17gpu_rec *d_recs_mine;   int32_t *d_cnt_mine;      // [hi-lo][cap], [hi-lo]
    m17gpu_rec *d_packed_mine; int32_t *d_offs_mine;     // [(hi-lo)*cap], [hi-lo+1]
    m17gpu_rec *d_packed_all;  int32_t *d_offs_all;      // rank 0 only: [TOTAL*cap] (capacity), [TOTAL+1]
};

// steps: how many NBLK x 40 ms slabs to process; next_slab(): the host's source of d_iq_all contents on rank 0
int run_rank(m17gpu_ctx *rx, ncclComm_t comm, int rank, int world, int TOTAL, int NBLK, RankBuffers &B,
             hipStream_t s, hipStream_t s_in, hipEvent_t ev_in[2], hipEvent_t ev_free[2], int steps)
{
    int lo, hi;
    m17gpu_shard_range(rank, world, TOTAL, &lo, &hi);
    const int cap = 2 * NBLK + 2;
    int32_t totals[64];
    int rc;
    // prologue: the first slab's scatter
    if ((rc = m17gpu_shard_scatter_iq(rx, comm, rank, world, 0, B.d_iq_all, TOTAL, NBLK, B.d_iq_mine[0], s_in)) != M17GPU_OK) return rc;
    hipEventRecord(ev_in[0], s_in);
    for (int k = 0; k < steps; ++k) {
        const int cur = k & 1, nxt = cur ^ 1;
        if (k + 1 < steps) {                             // the scatter of step k+1 beside the compute of step k
            hipStreamWaitEvent(s_in, ev_free[nxt], 0);   // buffer nxt was last read by step k-1
            if ((rc = m17gpu_shard_scatter_iq(rx, comm, rank, world, 0, B.d_iq_all, TOTAL, NBLK, B.d_iq_mine[nxt], s_in)) != M17GPU_OK) return rc;
            hipEventRecord(ev_in[nxt], s_in);
        }
        hipStreamWaitEvent(s, ev_in[cur], 0);
        if ((rc = m17gpu_rx_blocks(rx, B.d_iq_mine[cur], NBLK, /*full chain*/ 1, B.d_recs_mine, cap, B.d_cnt_mine, nullptr, nullptr, s)) != M17GPU_OK) return rc;
        hipEventRecord(ev_free[cur], s);
        if ((rc = m17gpu_pack_records(rx, B.d_recs_mine, cap, B.d_cnt_mine, B.d_packed_mine, (hi - lo) * cap, B.d_offs_mine, s)) != M17GPU_OK) return rc;
        if ((rc = m17gpu_shard_gather_packed(rx, comm, rank, world, 0, B.d_packed_mine, (hi - lo) * cap, B.d_offs_mine, TOTAL,
                                             B.d_packed_all, TOTAL * cap, B.d_offs_all, totals, s)) != M17GPU_OK) return rc;
        hipStreamSynchronize(s);
        // rank 0: channel c's records of this step are rows d_offs_all[c] .. d_offs_all[c+1] of d_packed_all
        // (r.flags & M17GPU_F_DELIVERED -> m17_net_new_rx_data(frame id, LSF, r.fn, &r.data[8]))
    }
    return M17GPU_OK;
}
