// m17_book.hip -- k_book_chan: per-channel, in-order bookkeeping of what m17_rx_parse / m17_aos /
// m17_los do to file-static state, one wave per channel (included after m17_decode_quad.hip).
//
// The channel's records are replayed in event order: LICH reassembly with its CRC, the delivery
// gate, decode_link_frame's CRC quirk, packet reassembly, the m17_dbase counters
// (m17_rx_parse.cpp:34-101,144-158; m17_dbase.cpp:60-82).  One wave runs it with uniform control
// flow; the 30-byte CRC is evaluated across 30 lanes: CRC-16 is linear over GF(2), so
// crc(msg) = crc(30 zero bytes) xor XOR_i XOR_{bit k of msg[i]} E[i][k] with 240 precomputed
// basis words.
#pragma clang fp contract(off)

namespace m17dev {

// XOR over lanes 0..31, result wave-uniform.  DPP row operations (quad swaps, half-row and
// row mirror) instead of ds_bpermute shuffles: this sits on the per-record critical path of
// the sequential bookkeeping.
__device__ __forceinline__ uint32_t xor_reduce32(uint32_t v)
{
    v ^= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xF, 0xF, true);     // quad_perm [1,0,3,2]
    v ^= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xF, 0xF, true);     // quad_perm [2,3,0,1]
    v ^= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x141, 0xF, 0xF, true);    // row_half_mirror
    v ^= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x140, 0xF, 0xF, true);    // row_mirror
    return (uint32_t)__builtin_amdgcn_readlane((int)v, 0) ^ (uint32_t)__builtin_amdgcn_readlane((int)v, 16);
}

// CRC-16/M17 (m17_crc.cpp:26-35) of 30 bytes held in LDS, computed by lanes 0..29
__device__ __forceinline__ uint32_t crc30_wave(const uint8_t *p, const uint16_t *basis, int lane)
{
    uint32_t v = 0;
    if (lane < 30) {
        const uint32_t b = p[lane];
        const uint4 bv = reinterpret_cast<const uint4 *>(basis)[lane];         // the lane's 8 basis words, one LDS read
        const uint32_t w[4] = {bv.x, bv.y, bv.z, bv.w};
#pragma unroll
        for (int k = 0; k < 8; ++k) v ^= (b >> k & 1u) ? ((w[k >> 1] >> (16 * (k & 1))) & 0xFFFFu) : 0u;
    }
    v = xor_reduce32(v);
    return (uint32_t)__builtin_amdgcn_readfirstlane((int)(v ^ 0x1B73u));   // 0x1B73 = crc of 30 zero bytes
}

// same CRC with the 30 bytes held one per lane and the lane's 8 basis words in registers
__device__ __forceinline__ uint32_t crc30_reg(uint32_t byte, uint4 bv, int lane)
{
    uint32_t v = 0;
    if (lane < 30) {
        const uint32_t w[4] = {bv.x, bv.y, bv.z, bv.w};
#pragma unroll
        for (int k = 0; k < 8; ++k) v ^= (byte >> k & 1u) ? ((w[k >> 1] >> (16 * (k & 1))) & 0xFFFFu) : 0u;
    }
    v = xor_reduce32(v);
    return (uint32_t)__builtin_amdgcn_readfirstlane((int)(v ^ 0x1B73u));
}

// CRC of the first L bytes of p (LDS), any L <= 800, across the wave: tab = d_crc_basis
__device__ __forceinline__ uint32_t crc_var_wave(const uint8_t *p, int L, const uint16_t *tab, int lane)
{
    const uint16_t *Z = tab + 240;
    uint32_t v = 0;
    for (int i = lane; i < L; i += 64) {
        const uint32_t b = p[i];
        const uint16_t *e = tab + 240 + 801 + (size_t)(L - 1 - i) * 8;
#pragma unroll
        for (int k = 0; k < 8; ++k) v ^= (b >> k & 1u) ? (uint32_t)e[k] : 0u;
    }
    v ^= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xF, 0xF, true);
    v ^= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xF, 0xF, true);
    v ^= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x141, 0xF, 0xF, true);
    v ^= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x140, 0xF, 0xF, true);
    const uint32_t r = (uint32_t)__builtin_amdgcn_readlane((int)v, 0) ^ (uint32_t)__builtin_amdgcn_readlane((int)v, 16) ^
                       (uint32_t)__builtin_amdgcn_readlane((int)v, 32) ^ (uint32_t)__builtin_amdgcn_readlane((int)v, 48);
    return r ^ (uint32_t)Z[L];
}

struct alignas(16) LsfShared {
    uint16_t basis[240];
    uint8_t  lsf[2][32];
    uint8_t  packet[800];
};

// channel tables and LICH / packet buffers into LDS, by `nthreads` threads
__device__ __forceinline__ void lsf_shared_init(LsfShared &ls, const ChanState &cs, const uint16_t *crc_basis, int t, int nthreads)
{
    for (int q = t; q < 240; q += nthreads) ls.basis[q] = crc_basis[q];
    for (int q = t; q < 16; q += nthreads) reinterpret_cast<uint32_t *>(ls.lsf)[q] = reinterpret_cast<const uint32_t *>(cs.lsf)[q];
    // m_packet: only its first 30 bytes are needed up front (decode_link_frame's CRC quirk, m17_rx_parse.cpp:98);
    // the other 768 come in when the first packet frame of the call does (bookkeeping_wave), and the buffer goes
    // back to the state only if a packet frame wrote to it -- stream traffic never touches it
    for (int q = t; q < 8; q += nthreads) reinterpret_cast<uint32_t *>(ls.packet)[q] = reinterpret_cast<const uint32_t *>(cs.packet)[q];
}

// What m17_rx_parse does to file-static state, replayed over the channel's records in event
// order by one wave with uniform control flow (see the file header).  Records are read
// from rsrc (HBM or an LDS copy); updated flags go to crecs.
__device__ void bookkeeping_wave(ChanState &cs, m17gpu_rec_dev *crecs, const m17gpu_rec_dev *rsrc, int n, LsfShared &ls, int lane,
                                 const uint16_t *crc_tab)
{
    uint32_t g_errors = (uint32_t)uni((int)cs.g_errors), n_frames = (uint32_t)uni((int)cs.n_frames);
    uint32_t in_frame = (uint32_t)uni((int)cs.in_frame), epoch = (uint32_t)uni((int)cs.frame_id_epoch);
    int packet_idx = uni(cs.packet_idx);
    bool packet_in = false;                                               // m_packet[32..800) loaded / buffer written to
    // m_lsf[0] / m_lsf[1] live one byte per lane in registers for the replay (lanes 0..29), and so do
    // the lane's eight CRC basis words: a LICH update and its CRC need no LDS round trip
    uint32_t b0 = (lane < 30) ? ls.lsf[0][lane] : 0u, b1 = (lane < 30) ? ls.lsf[1][lane] : 0u;
    const uint4 bv = (lane < 30) ? reinterpret_cast<const uint4 *>(ls.basis)[lane] : make_uint4(0u, 0u, 0u, 0u);
    bool lsf1_ok = crc30_reg(b1, bv, lane) == 0;                         // m_lsf[1] only ever changes to CRC-good content
    bool gate_ok = crc30_wave(ls.packet, ls.basis, lane) == 0;           // decode_link_frame's quirk (m17_rx_parse.cpp:98)
    // 64 records at a time, one per lane: the five words the replay needs sit in registers and reach
    // the (wave-uniform) control code through v_readlane; changed flag words go back with one store
    // per lane.  No memory access on the per-record path except packet payload bytes (rare).
    for (int base = 0; base < n; base += 64) {
        const int m = min(64, n - base);
        uint32_t rw0 = 0, rw1 = 0, rd0 = 0, rd1 = 0, rd11 = 0;
        if (lane < m) {
            const uint32_t *r = reinterpret_cast<const uint32_t *>(&rsrc[base + lane]);
            rw0 = r[0]; rw1 = r[1]; rd0 = r[5]; rd1 = r[6]; rd11 = r[11];
        }
        uint32_t myflags = rw1 & 0xFFFF;
        for (int i = 0; i < m; ++i) {
            const uint32_t w0 = (uint32_t)__builtin_amdgcn_readlane((int)rw0, i), w1 = (uint32_t)__builtin_amdgcn_readlane((int)rw1, i);
            uint32_t flags = w1 & 0xFFFF;
            const int type = (int)(w0 & 0xFF);
            if (flags & M17_F_AOS) { g_errors = 0; n_frames = 0; in_frame = 1; epoch++; continue; }
            if (flags & (M17_F_EOT | M17_F_LOST)) { in_frame = 0; epoch++; continue; }
            if (!(flags & M17_F_PARSED)) continue;
            const uint32_t old_flags = flags;
            if (type == 0 || type == 5) {
                epoch++;
            } else if (type == 1) {
                if (gate_ok) flags |= M17_F_LSF_GATE;
            } else if (type == 2) {
                g_errors += (w0 >> 16) & 0xFF; n_frames++;
                // update_lich (m17_rx_parse.cpp:71-85): data[0..5] = words 5 and low half of 6
                const uint32_t d0 = (uint32_t)__builtin_amdgcn_readlane((int)rd0, i), d1 = (uint32_t)__builtin_amdgcn_readlane((int)rd1, i);
                const int seq = (int)((d1 >> 8 & 0xFF) >> 5);
                if (seq < 6) {
                    const int k = lane - seq * 5;                           // byte k of the chunk lands in lane seq*5 + k
                    if (k >= 0 && k < 5) b0 = ((k < 4) ? (d0 >> (8 * k)) : d1) & 0xFFu;
                    if (crc30_reg(b0, bv, lane) == 0) {
                        b1 = b0;
                        lsf1_ok = true;
                        flags |= M17_F_LICH_OK;
                    }
                }
                if (lsf1_ok) flags |= M17_F_DELIVERED;                      // :148
            } else if (type == 3) {
                // parse_packet (m17_rx_parse.cpp:34-51); data bytes live in words 5..11
                const uint32_t d25 = ((uint32_t)__builtin_amdgcn_readlane((int)rd11, i) >> 8) & 0xFF;     // data[25]
                const int eof = (int)(d25 >> 7), fnv = (int)((d25 >> 2) & 0x1F);
                const uint8_t *dbytes = reinterpret_cast<const uint8_t *>(reinterpret_cast<const uint32_t *>(&rsrc[base + i]) + 5);
                if (!packet_in) {                                           // wave-uniform: type comes from a readlane
                    for (int q = 8 + lane; q < 200; q += 64)
                        reinterpret_cast<uint32_t *>(ls.packet)[q] = reinterpret_cast<const uint32_t *>(cs.packet)[q];
                    packet_in = true;
                    group_sync();
                }
                if (eof) {
                    int cnt = fnv;
                    if (packet_idx + cnt > 800) cnt = 800 - packet_idx;
                    if (lane < cnt) ls.packet[packet_idx + lane] = dbytes[lane];
                    packet_idx += cnt;
                    group_sync();
                    // variable length, up to 800 bytes: by linearity across the wave (a serial table walk
                    // took 24 us for a full buffer and set the kernel's tail on spurious packet frames)
                    if (crc_var_wave(ls.packet, packet_idx, crc_tab, lane) == 0) flags |= M17_F_PKT_VALID;
                    packet_idx = 0;
                } else {
                    if (lane < 25) ls.packet[fnv * 25 + lane] = dbytes[lane];
                    packet_idx = fnv * 25;
                    group_sync();
                }
                gate_ok = crc30_wave(ls.packet, ls.basis, lane) == 0;
            }
            if (flags != old_flags && lane == i) myflags = flags;
        }
        if (lane < m && myflags != (rw1 & 0xFFFF))
            reinterpret_cast<uint32_t *>(&crecs[base + lane])[1] = (rw1 & 0xFFFF0000u) | myflags;
    }
    // ---- state back
    if (lane < 30) { ls.lsf[0][lane] = (uint8_t)b0; ls.lsf[1][lane] = (uint8_t)b1; }
    if (lane == 0) {
        cs.g_errors = g_errors; cs.n_frames = n_frames; cs.in_frame = in_frame; cs.frame_id_epoch = epoch;
        cs.packet_idx = packet_idx;
    }
    group_sync();
    if (lane < 16) reinterpret_cast<uint32_t *>(cs.lsf)[lane] = reinterpret_cast<const uint32_t *>(ls.lsf)[lane];
    if (packet_in)
        for (int q = lane; q < 200; q += 64) reinterpret_cast<uint32_t *>(cs.packet)[q] = reinterpret_cast<const uint32_t *>(ls.packet)[q];
}


// one wave per channel
__global__ __launch_bounds__(64)
void k_book_chan(ChanState *__restrict__ st, m17gpu_rec_dev *__restrict__ recs, int rec_cap,
                 const int32_t *__restrict__ counts, const uint16_t *__restrict__ crc_basis)
{
    __shared__ LsfShared ls;
    const int lane = lane_id(), chan = (int)blockIdx.x;
    ChanState &cs = st[chan];
    m17gpu_rec_dev *crecs = recs + (size_t)chan * rec_cap;
    lsf_shared_init(ls, cs, crc_basis, lane, 64);
    group_sync();
    bookkeeping_wave(cs, crecs, crecs, min(counts[chan], rec_cap), ls, lane, crc_basis);
}

} // namespace m17dev
