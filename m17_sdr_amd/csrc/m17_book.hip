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

// XOR over all 64 lanes, result wave-uniform
__device__ __forceinline__ uint32_t xor_reduce64(uint32_t v)
{
    v ^= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xF, 0xF, true);
    v ^= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xF, 0xF, true);
    v ^= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x141, 0xF, 0xF, true);
    v ^= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x140, 0xF, 0xF, true);
    return (uint32_t)__builtin_amdgcn_readlane((int)v, 0) ^ (uint32_t)__builtin_amdgcn_readlane((int)v, 16) ^
           (uint32_t)__builtin_amdgcn_readlane((int)v, 32) ^ (uint32_t)__builtin_amdgcn_readlane((int)v, 48);
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
    return xor_reduce64(v) ^ (uint32_t)Z[L];
}

// ChanState.pad[0]: facts about the channel's LICH / packet buffers that only the bookkeeping kernels change
constexpr uint32_t BK_VALID = 1u, BK_LSF1_OK = 2u, BK_GATE_OK = 4u, BK_LSF0_GOOD = 8u, BK_EQ01 = 16u;
//   BK_LSF1_OK  CRC(m_lsf[1]) == 0 (m_lsf[1] only ever changes to CRC-good content)
//   BK_GATE_OK  CRC(m_packet[0..30)) == 0: decode_link_frame's quirk (m17_rx_parse.cpp:98)
//   BK_LSF0_GOOD CRC(m_lsf[0]) == 0;  BK_EQ01 m_lsf[1] == m_lsf[0]

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
// cnet (optional): the channel's rows of the network sink, [rec_cap][56] bytes: every DELIVERED stream frame is
// written there as the 54-byte M17-over-IP frame m17_net_new_rx_data builds (net_add_* m17_net.cpp:25-49, :53-74) from
// (m_frame_id, m_lsf[1], fn, payload) AT THAT POINT of the replay -- the LSF of a later transmission in the same call
// must not leak into earlier frames, which is why the formatter lives here and not in a pass over the finished
// records.  One byte per lane; the CRC-16 over the first 52 bytes by GF(2) linearity like the LICH CRC.  Not mirrored:
// the reference's 54-byte memcpy out of the 30-byte m_lsf[1] (:58).  m_frame_id = sid_base + frame_id_epoch.
__device__ void bookkeeping_wave(ChanState &cs, m17gpu_rec_dev *crecs, const m17gpu_rec_dev *rsrc, int n, LsfShared &ls, int lane,
                                 const uint16_t *crc_tab, uint8_t *cnet = nullptr, uint32_t sid_base = 0,
                                 unsigned long long dst_override = 0ull)
{
    // the lane's eight basis words of a 52-byte message (byte `lane`), and the CRC of 52 zero bytes
    uint32_t e52[4] = {0u, 0u, 0u, 0u};
    uint32_t z52 = 0;
    if (cnet) {
        z52 = crc_tab[240 + 52];
        if (lane < 52) {
            const uint16_t *e = crc_tab + 240 + 801 + (size_t)(51 - lane) * 8;
#pragma unroll
            for (int k = 0; k < 4; ++k) e52[k] = (uint32_t)e[2 * k] | ((uint32_t)e[2 * k + 1] << 16);
        }
    }
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
        uint32_t rw0 = 0, rw1 = 0, rd0 = 0, rd1 = 0, rd11 = 0, rp0 = 0, rp1 = 0, rp2 = 0, rp3 = 0;
        if (lane < m) {
            const uint32_t *r = reinterpret_cast<const uint32_t *>(&rsrc[base + lane]);
            rw0 = r[0]; rw1 = r[1]; rd0 = r[5]; rd1 = r[6]; rd11 = r[11];
            if (cnet) { rp0 = r[7]; rp1 = r[8]; rp2 = r[9]; rp3 = r[10]; }            // data[8..24): the 16 payload bytes
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
                flags &= ~(uint32_t)(M17_F_LICH_OK | M17_F_DELIVERED);     // the decoder's guess (m17_decode_quad.hip): decided here
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
                if (cnet && lsf1_ok) {
                    // m17_net_new_rx_data(m_frame_id, m_lsf[1], fn, data): byte L of the frame in lane L
                    const uint32_t pw[4] = {(uint32_t)__builtin_amdgcn_readlane((int)rp0, i), (uint32_t)__builtin_amdgcn_readlane((int)rp1, i),
                                            (uint32_t)__builtin_amdgcn_readlane((int)rp2, i), (uint32_t)__builtin_amdgcn_readlane((int)rp3, i)};
                    const uint32_t fnv = w1 >> 16, sid = (sid_base + epoch) & 0xFFFFu;
                    const uint32_t lb = (uint32_t)__shfl((int)b1, (lane + 58) & 63, 64);     // m_lsf[1][L - 6]
                    uint32_t by = 0;
                    if (lane < 4) by = (0x2037314Du >> (8 * lane)) & 0xFFu;                 // "M17 "
                    else if (lane == 4) by = sid >> 8;
                    else if (lane == 5) by = sid & 0xFFu;
                    else if (lane < 12) by = dst_override ? (uint32_t)(dst_override >> (40 - 8 * (lane - 6))) & 0xFFu : lb;
                    else if (lane < 34) by = lb;
                    else if (lane == 34) by = fnv >> 8;
                    else if (lane == 35) by = fnv & 0xFFu;
                    else if (lane < 52) {
                        const int q = lane - 36;
                        const uint32_t w = (q < 4) ? pw[0] : (q < 8) ? pw[1] : (q < 12) ? pw[2] : pw[3];
                        by = (w >> (8 * (q & 3))) & 0xFFu;
                    }
                    uint32_t v = 0;
#pragma unroll
                    for (int k = 0; k < 8; ++k) v ^= (by >> k & 1u) ? ((e52[k >> 1] >> (16 * (k & 1))) & 0xFFFFu) : 0u;
                    const uint32_t crc = xor_reduce64(v) ^ z52;                                // net_add_crc, :45-49
                    if (lane == 52) by = crc >> 8;
                    if (lane == 53) by = crc & 0xFFu;
                    // four lanes to a dword, rows of 56 bytes
                    const uint32_t w4 = by | ((uint32_t)__builtin_amdgcn_update_dpp(0, (int)by, 0x55, 0xF, 0xF, true) << 8) |
                                        ((uint32_t)__builtin_amdgcn_update_dpp(0, (int)by, 0xAA, 0xF, 0xF, true) << 16) |
                                        ((uint32_t)__builtin_amdgcn_update_dpp(0, (int)by, 0xFF, 0xF, 0xF, true) << 24);
                    if ((lane & 3) == 0 && lane < 56)
                        *reinterpret_cast<uint32_t *>(cnet + (size_t)(base + i) * 56 + lane) = w4;
                }
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
    {
        // what k_book_lanes would otherwise have to evaluate per channel and call (BK_* bits): both kernels leave it current
        const bool good0 = crc30_reg(b0, bv, lane) == 0;
        const bool eq01 = __builtin_amdgcn_ballot_w64(lane < 30 && b0 != b1) == 0ull;
        if (lane == 0) cs.pad[0] = (int32_t)(BK_VALID | (lsf1_ok ? BK_LSF1_OK : 0u) | (gate_ok ? BK_GATE_OK : 0u) | (good0 ? BK_LSF0_GOOD : 0u) |
                                             (eq01 ? BK_EQ01 : 0u));
    }
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
                 const int32_t *__restrict__ counts, const uint16_t *__restrict__ crc_basis,
                 uint8_t *__restrict__ net, const uint16_t *__restrict__ stream_ids, unsigned long long dst_override, int chan0,
                 int32_t *__restrict__ nwork)
{
    __shared__ LsfShared ls;
    const int lane = lane_id(), chan = (int)blockIdx.x;
    // the work-list counters of this call have been consumed (the decoder ran in front of this kernel): zero them for the
    // next call here instead of a fill kernel in front of every call (the context starts with them zeroed)
    if (nwork && chan == 0 && lane < 4) nwork[lane] = 0;
    ChanState &cs = st[chan];
    m17gpu_rec_dev *crecs = recs + (size_t)chan * rec_cap;
    lsf_shared_init(ls, cs, crc_basis, lane, 64);
    group_sync();
    bookkeeping_wave(cs, crecs, crecs, min(counts[chan], rec_cap), ls, lane, crc_basis,
                     net ? net + (size_t)(chan0 + chan) * rec_cap * 56 : nullptr,
                     stream_ids ? (uint32_t)stream_ids[chan0 + chan] : 0u, dst_override);
}

// ---------------------------------------------------------------------------------------------------------------
// k_book_lanes (round 5): the same replay with ONE LANE PER CHANNEL (BL_CH = 8 channel lanes per wave; all 64 lanes serve).
//
// k_book_chan replays a channel's records with uniform control flow -- 790 scalar instructions per channel -- and a CU issues
// one scalar instruction per cycle for all of its waves: 16,384 channels take 0.041 ms whatever else is done to the kernel
// (profiles/r05_bookkeeping_kernel_bound.txt).  What the replay does per record is a handful of integer operations,
// the same for every channel; only the LICH CRC needs lanes side by side, and it needs evaluating only when a chunk
// CHANGES m_lsf[0] -- once per transmission at most, never in the steady state of a stream.  So: a lane walks its channel's
// records (their words staged in LDS by the whole wave), keeps the channel's counters and the six LICH chunks of m_lsf[0] in
// registers and the two LICH buffers in LDS; the lanes
// whose chunk changed their buffer are served one after the other by the whole wave with the thirty-lane CRC of
// crc30_reg; what both kernels would otherwise evaluate per call (is m_lsf[1] good, is the packet gate open, ...) sits in
// the channel state (BK_* bits, kept current by both).  A packet frame (reassembly into the channel's 800-byte buffer,
// CRCs of variable length; in a stream they only occur as false syncs, a few per thousand channels and call) is served
// the same way: by the whole wave, for that one record.
// The network sink is the wave-per-channel kernel's: contexts with one attached launch k_book_chan.
// ---------------------------------------------------------------------------------------------------------------
constexpr int BL_ROW = 80;                      // bytes per lane in LDS: m_lsf[0] at 0, m_lsf[1] at 32 (ChanState layout), 16 spare
#ifndef BL_CH
#define BL_CH 8                                 // channels per wave: lanes 0 .. BL_CH - 1 own one each, all 64 serve the CRCs
#endif
__global__ __launch_bounds__(64)
void k_book_lanes(ChanState *__restrict__ st, m17gpu_rec_dev *__restrict__ recs, int rec_cap,
                  const int32_t *__restrict__ counts, const uint16_t *__restrict__ crc_basis,
                  int32_t *__restrict__ nwork, int cn)
{
    __shared__ __attribute__((aligned(16))) uint8_t rows[BL_CH * BL_ROW];
    __shared__ LsfShared ls;                                            // its packet buffer: the variable-length CRC of a packet frame
    const int lane = lane_id();
    const int chan = (int)blockIdx.x * BL_CH + lane;
    const bool have = lane < BL_CH && chan < cn;
    if (nwork && blockIdx.x == 0 && lane < 4) nwork[lane] = 0;          // see k_book_chan
    ChanState &cs = st[have ? chan : cn - 1];
    m17gpu_rec_dev *crecs = recs + (size_t)(have ? chan : cn - 1) * rec_cap;
    // the four words the replay needs of every record of the wave's channels (0, 1: type / votes / golay errors / frame
    // errors, flags / fn; 5, 6: data[0..8), the LICH bytes), staged in LDS by all 64 lanes
    extern __shared__ __attribute__((aligned(16))) uint32_t recw[];                 // [BL_CH][rec_cap][4]
    // every load of the prologue is requested before the first wait: the count, the state's scalars, the two LICH buffers,
    // the CRC basis, and the first eight records' words per lane (512 of the wave's BL_CH x rec_cap records; the rest, for
    // calls of more than 31 blocks, in further rounds of eight)
    const int total = BL_CH * rec_cap;
    constexpr int BL_G = 8;
    uint2 sa[BL_G]; uint32_t se0[BL_G], se1[BL_G];
    auto stage_request = [&](int base) {
#pragma unroll
        for (int u = 0; u < BL_G; ++u) {
            const int idx = min(base + u * 64 + lane, total - 1);
            const int ch = idx / rec_cap;
            const int c2 = min((int)blockIdx.x * BL_CH + ch, cn - 1);
            const uint32_t *r = reinterpret_cast<const uint32_t *>(recs + (size_t)c2 * rec_cap + (idx - ch * rec_cap));
            sa[u] = *reinterpret_cast<const uint2 *>(r);
            se0[u] = r[5]; se1[u] = r[6];
        }
    };
    auto stage_commit = [&](int base) {
#pragma unroll
        for (int u = 0; u < BL_G; ++u) {
            const int idx = base + u * 64 + lane;
            if (idx < total) *reinterpret_cast<uint4 *>(&recw[4 * idx]) = make_uint4(sa[u].x, sa[u].y, se0[u], se1[u]);
        }
    };
    const int cnt_raw = have ? counts[chan] : 0;
    uint32_t g_errors = cs.g_errors, n_frames = cs.n_frames, in_frame = cs.in_frame, epoch = cs.frame_id_epoch;
    int packet_idx = cs.packet_idx;
    uint32_t bk = (uint32_t)cs.pad[0];
    const uint4 *lsrc = reinterpret_cast<const uint4 *>(cs.lsf);
    const uint4 la = lsrc[0], lb = lsrc[1], lc = lsrc[2], ld = lsrc[3];
    const uint4 bv = (lane < 30) ? reinterpret_cast<const uint4 *>(crc_basis)[lane] : make_uint4(0u, 0u, 0u, 0u);
    stage_request(0);
    const int n = have ? min(cnt_raw, rec_cap) : 0;
    if (lane < BL_CH) {
        uint4 *dst = reinterpret_cast<uint4 *>(rows + lane * BL_ROW);
        dst[0] = la; dst[1] = lb; dst[2] = lc; dst[3] = ld;
    }
    stage_commit(0);
    for (int base = BL_G * 64; base < total; base += BL_G * 64) { stage_request(base); stage_commit(base); }
    group_sync();
    // CRC of thirty bytes at rows[L * BL_ROW + off ..), L wave-uniform: the wave's lanes 0..29 take a byte each
    auto crc_of = [&](int L, int off) -> bool {
        const uint32_t b = (lane < 30) ? rows[L * BL_ROW + off + lane] : 0u;
        return crc30_reg(b, bv, lane) == 0;
    };
    // a state that was never left by a bookkeeping kernel (after a reset): evaluate what the BK_* bits say
    {
        unsigned long long todo = __builtin_amdgcn_ballot_w64(have && !(bk & BK_VALID));
        while (todo) {
            const int L = (int)__builtin_ctzll(todo);
            todo &= todo - 1;
            const bool ok1 = crc_of(L, 32), good0 = crc_of(L, 0);
            const uint32_t x = (lane < 30) ? rows[L * BL_ROW + lane] : 0u, y = (lane < 30) ? rows[L * BL_ROW + 32 + lane] : 0u;
            const bool eq = __builtin_amdgcn_ballot_w64(x != y) == 0ull;
            // the gate: CRC of the first thirty bytes of the channel's packet buffer
            const ChanState &cl = st[(int)blockIdx.x * BL_CH + L];
            const uint32_t pb = (lane < 30) ? cl.packet[lane] : 0u;
            const bool gate = crc30_reg(pb, bv, lane) == 0;
            if (lane == L) bk = BK_VALID | (ok1 ? BK_LSF1_OK : 0u) | (gate ? BK_GATE_OK : 0u) | (good0 ? BK_LSF0_GOOD : 0u) | (eq ? BK_EQ01 : 0u);
        }
    }
    bool lsf_dirty = false, pkt_dirty = false;
    const int maxn = [&]() {                                            // wave maximum of n
        int m = n;
        m = max(m, __builtin_amdgcn_update_dpp(0, m, 0xB1, 0xF, 0xF, true));
        m = max(m, __builtin_amdgcn_update_dpp(0, m, 0x4E, 0xF, 0xF, true));
        m = max(m, __builtin_amdgcn_update_dpp(0, m, 0x141, 0xF, 0xF, true));
        m = max(m, __builtin_amdgcn_update_dpp(0, m, 0x140, 0xF, 0xF, true));
        return max(max(__builtin_amdgcn_readlane(m, 0), __builtin_amdgcn_readlane(m, 16)),
                   max(__builtin_amdgcn_readlane(m, 32), __builtin_amdgcn_readlane(m, 48)));
    }();
    // m_lsf[0] as its six LICH chunks in registers (bytes 5 s .. 5 s + 3 and byte 5 s + 4): what a frame's chunk is compared
    // with -- in the steady state of a stream it finds itself there, and the iteration touches LDS for its record only
    uint32_t c32[6], c8[6];
    {
        const uint8_t *p = rows + (lane < BL_CH ? lane : 0) * BL_ROW;
#pragma unroll
        for (int q = 0; q < 6; ++q) {
            c32[q] = (uint32_t)p[5 * q] | ((uint32_t)p[5 * q + 1] << 8) | ((uint32_t)p[5 * q + 2] << 16) | ((uint32_t)p[5 * q + 3] << 24);
            c8[q] = p[5 * q + 4];
        }
    }
    const uint32_t *myrec = &recw[4 * ((lane < BL_CH ? lane : 0) * rec_cap)];
    uint4 rw_next = *reinterpret_cast<const uint4 *>(myrec);
    for (int i = 0; i < maxn; ++i) {
        {
            const uint4 rw = rw_next;
            rw_next = *reinterpret_cast<const uint4 *>(myrec + 4 * min(i + 1, rec_cap - 1));
            const uint32_t cw0q = rw.x, cw1q = rw.y, cd0q = rw.z, cd1q = rw.w;
            const bool act = i < n;
            const uint32_t w0 = cw0q, w1 = cw1q;
            uint32_t flags = w1 & 0xFFFFu;
            const uint32_t type = w0 & 0xFFu;
            const bool aos = act && (flags & M17_F_AOS);
            const bool end = act && !aos && (flags & (M17_F_EOT | M17_F_LOST));
            const bool parsed = act && !aos && !end && (flags & M17_F_PARSED);
            if (aos) { g_errors = 0; n_frames = 0; in_frame = 1; epoch++; }
            if (end) { in_frame = 0; epoch++; }
            const uint32_t old_flags = flags;
            if (parsed && (type == 0 || type == 5)) epoch++;
            if (parsed && type == 1 && (bk & BK_GATE_OK)) flags |= M17_F_LSF_GATE;
            const bool strm = parsed && type == 2;
            bool need_crc = false;
            bool chunk = false;
            if (strm) {
                flags &= ~(uint32_t)(M17_F_LICH_OK | M17_F_DELIVERED);     // the decoder's guess: decided here
                g_errors += (w0 >> 16) & 0xFFu; n_frames++;
                // update_lich (m17_rx_parse.cpp:71-85): data[0..5], counter in data[5] >> 5
                const uint32_t d0 = cd0q, d1 = cd1q;
                const uint32_t seq = ((d1 >> 8) & 0xFFu) >> 5;
                if (seq < 6) {
                    chunk = true;
                    uint32_t o32 = c32[0], o8 = c8[0];
#pragma unroll
                    for (int q = 1; q < 6; ++q) { o32 = (seq == (uint32_t)q) ? c32[q] : o32; o8 = (seq == (uint32_t)q) ? c8[q] : o8; }
                    need_crc = o32 != d0 || o8 != (d1 & 0xFFu);
                }
            }
            // the lanes whose chunk changed m_lsf[0]: registers and LDS row updated, then its CRC, by the whole wave, one lane after the other
            {
                unsigned long long todo = __builtin_amdgcn_ballot_w64(need_crc);
                if (todo) {
                    if (need_crc) {
                        const uint32_t d0 = cd0q, d1 = cd1q;
                        const uint32_t seq = ((d1 >> 8) & 0xFFu) >> 5;
#pragma unroll
                        for (int q = 0; q < 6; ++q) { if (seq == (uint32_t)q) { c32[q] = d0; c8[q] = d1 & 0xFFu; } }
                        uint8_t *p = rows + (lane < BL_CH ? lane : 0) * BL_ROW + seq * 5;
                        p[0] = (uint8_t)d0; p[1] = (uint8_t)(d0 >> 8); p[2] = (uint8_t)(d0 >> 16); p[3] = (uint8_t)(d0 >> 24); p[4] = (uint8_t)d1;
                        lsf_dirty = true;
                        bk &= ~BK_EQ01;
                    }
                    wave_fence();
                    while (todo) {
                        const int L = (int)__builtin_ctzll(todo);
                        todo &= todo - 1;
                        const bool good0 = crc_of(L, 0);
                        if (lane == L) bk = (bk & ~BK_LSF0_GOOD) | (good0 ? BK_LSF0_GOOD : 0u);
                    }
                }
            }
            // packet frames (parse_packet, m17_rx_parse.cpp:34-51): the wave serves them one after the other.
            // Coherence of the global packet buffer inside this wave: some lanes store bytes of c2s.packet, group_sync() (a
            // workgroup-scope release + acquire: the stores have left the wave, vmcnt = 0) and other lanes load them back
            // through the SAME CU's vector L1 -- which is write-through and shared by everything that runs on the CU, so a
            // later load of this wave (or of a sibling in the workgroup) cannot be served a copy older than the store: what
            // the AMDGPU memory model gives at workgroup scope outside tgsplit mode.  (The agent-scope loads of the DC
            // offsets in sync_wave_channel<.., OFFS_AGENT> are stronger than that model requires; they are kept because
            // they cost nothing measurable.  Data another CU wrote IN THIS KERNEL is never read here.)
            {
                unsigned long long todo = __builtin_amdgcn_ballot_w64(parsed && type == 3);
                while (todo) {
                    const int L = (int)__builtin_ctzll(todo);
                    todo &= todo - 1;
                    const int c2 = (int)blockIdx.x * BL_CH + L;
                    ChanState &c2s = st[c2];
                    const m17gpu_rec_dev *r2 = recs + (size_t)c2 * rec_cap + i;
                    int pidx = __builtin_amdgcn_readlane(packet_idx, L);
                    const uint8_t *dbytes = reinterpret_cast<const uint8_t *>(reinterpret_cast<const uint32_t *>(r2) + 5);
                    const uint32_t d25 = (uint32_t)uni((int)dbytes[25]);                          // data[25]
                    const int eof = (int)(d25 >> 7), fnv = (int)((d25 >> 2) & 0x1F);
                    bool valid_pkt = false;
                    if (eof) {
                        int cnt = fnv;
                        if (pidx + cnt > 800) cnt = 800 - pidx;
                        if (lane < cnt) c2s.packet[pidx + lane] = dbytes[lane];
                        pidx += cnt;
                        group_sync();
                        for (int q = lane; q < 200; q += 64) reinterpret_cast<uint32_t *>(ls.packet)[q] = reinterpret_cast<const uint32_t *>(c2s.packet)[q];
                        group_sync();
                        valid_pkt = crc_var_wave(ls.packet, pidx, crc_basis, lane) == 0;
                        pidx = 0;
                    } else {
                        if (lane < 25) c2s.packet[fnv * 25 + lane] = dbytes[lane];
                        pidx = fnv * 25;
                        group_sync();
                    }
                    const uint32_t pb = (lane < 30) ? c2s.packet[lane] : 0u;
                    const bool gate = crc30_reg(pb, bv, lane) == 0;
                    if (lane == L) {
                        packet_idx = pidx;
                        if (valid_pkt) flags |= M17_F_PKT_VALID;
                        bk = (bk & ~BK_GATE_OK) | (gate ? BK_GATE_OK : 0u);
                        pkt_dirty = true;
                    }
                }
            }
            if (chunk && (bk & BK_LSF0_GOOD)) {
                if (!(bk & BK_EQ01)) {                                      // copy_lich: m_lsf[1] = m_lsf[0]
                    const uint4 *s4 = reinterpret_cast<const uint4 *>(rows + (lane < BL_CH ? lane : 0) * BL_ROW);
                    uint4 *d4 = reinterpret_cast<uint4 *>(rows + (lane < BL_CH ? lane : 0) * BL_ROW + 32);
                    const uint4 a = s4[0], b = s4[1];
                    d4[0] = a; d4[1] = b;
                    bk |= BK_EQ01; lsf_dirty = true;
                }
                bk |= BK_LSF1_OK;
                flags |= M17_F_LICH_OK;
            }
            if (strm && (bk & BK_LSF1_OK)) flags |= M17_F_DELIVERED;        // :148
            if (parsed && flags != old_flags)
                reinterpret_cast<uint32_t *>(&crecs[i])[1] = (w1 & 0xFFFF0000u) | flags;
        }
    }
    // ---- state back
    wave_fence();
    if (have) {
        cs.g_errors = g_errors; cs.n_frames = n_frames; cs.in_frame = in_frame; cs.frame_id_epoch = epoch;
        cs.pad[0] = (int32_t)bk;
        if (pkt_dirty) cs.packet_idx = packet_idx;
        if (lsf_dirty) {
            const uint4 *src = reinterpret_cast<const uint4 *>(rows + (lane < BL_CH ? lane : 0) * BL_ROW);
            uint4 *dst = reinterpret_cast<uint4 *>(cs.lsf);
            dst[0] = src[0]; dst[1] = src[1]; dst[2] = src[2]; dst[3] = src[3];
        }
    }
}

// LSF field extraction for n LSFs (parse_lsf m17_rx_parse.cpp:52-70 with m17_decode_call m17_bit_utils.cpp:209-226 and
// m17_upack_type :245-254), one thread each: the batch form of the host's m17gpu_parse_lsf, same output struct.
struct LsfFieldsDev {                   // = m17gpu_lsf_fields (include/m17gpu.h), checked in m17gpu_capi.hip
    uint64_t dst, src;
    char     dst_call[10], src_call[10];
    uint8_t  p_s, dt, et, est, can, reserved;
    uint8_t  meta[14];
    uint16_t crc;
    uint8_t  crc_ok;
};
__device__ __forceinline__ void decode_call_dev(uint64_t word, char *call)
{
    if (word == 0xFFFFFFFFFFFFull) {
        const char b[10] = {'B', 'R', 'O', 'A', 'D', 'C', 'A', 'S', 'T', 0};
        for (int i = 0; i < 10; ++i) call[i] = b[i];
        return;
    }
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
__global__ void k_parse_lsf(const uint8_t *__restrict__ lsf, LsfFieldsDev *__restrict__ out, int n)
{
    const int i = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (i >= n) return;
    const uint8_t *in = lsf + (size_t)i * 30;
    uint8_t b[30];
    for (int k = 0; k < 30; ++k) b[k] = in[k];
    LsfFieldsDev f;
    f.dst = 0; f.src = 0;
    for (int k = 0; k < 6; ++k) { f.dst = (f.dst << 8) | b[k]; f.src = (f.src << 8) | b[6 + k]; }      // pack_8_to_48
    decode_call_dev(f.dst, f.dst_call);
    decode_call_dev(f.src, f.src_call);
    const unsigned tw = ((unsigned)b[12] << 8) | b[13];                                                 // pack_8_to_16
    f.reserved = (uint8_t)((tw >> 11) & 0x1F); f.can = (uint8_t)((tw >> 7) & 0xF); f.est = (uint8_t)((tw >> 5) & 0x3);
    f.et = (uint8_t)((tw >> 3) & 0x3); f.dt = (uint8_t)((tw >> 1) & 0x3); f.p_s = (uint8_t)(tw & 1);
    for (int k = 0; k < 14; ++k) f.meta[k] = b[14 + k];
    f.crc = (uint16_t)(((unsigned)b[28] << 8) | b[29]);
    uint16_t crc = 0xFFFF;                                                                              // m17_crc.cpp:26-35
    for (int k = 0; k < 30; ++k) crc = (uint16_t)((crc << 8) ^ c_tab.crc[((crc >> 8) ^ b[k]) & 0xFF]);
    f.crc_ok = crc == 0;
    // whole 64-byte struct, padding zeroed
    uint32_t w[16];
    for (int k = 0; k < 16; ++k) w[k] = 0;
    char *pw = reinterpret_cast<char *>(w);
    const char *pf = reinterpret_cast<const char *>(&f);
    for (int k = 0; k < 59; ++k) pw[k] = pf[k];
    uint32_t *o = reinterpret_cast<uint32_t *>(out + i);
    for (int k = 0; k < 16; ++k) o[k] = w[k];
}

} // namespace m17dev
