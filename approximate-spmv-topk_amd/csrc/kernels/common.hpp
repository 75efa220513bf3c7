// kernels/common.hpp -- Device helpers shared by every kernel: order keys, kernel parameter blocks, packet loads, fixed-point conversions.
// Part of engine.hip (one translation unit: included there in this order; device code only).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>

namespace tkspmv {

// ------------------------------------------------------------------------------------------------------------
// Device helpers
// ------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t order_key(float f) {  // monotone float -> u32; 0 is "nothing"
    uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float key_to_float(uint32_t k) {
    uint32_t u = (k & 0x80000000u) ? (k & 0x7FFFFFFFu) : ~k;
    return __uint_as_float(u);
}

struct StreamParams {
    const uint8_t *packets;
    const uint32_t *pkt_row;
    const uint32_t *part_first;
    const uint32_t *part_count;
    const float *x;
    uint32_t n_parts, cols, packet_bytes;
    // Uniform partition table (the packer's usual outcome: every partition but the last holds uni_ppp packets, back to back):
    // a wave derives its range from its partition number instead of two dependent loads at the head of the launch (a cold
    // round trip through memory before the first packet can be requested). 0: read part_first / part_count.
    uint32_t uni_ppp, uni_last;
    uint32_t n_sets;        // 0 => threshold exchange disabled (fewer publishing groups than k), else 1
    uint32_t k;
    uint32_t n_groups_pub;  // groups [0, n_groups_pub) publish maxima (<= 1024)
    uint32_t gpw;           // groups per workgroup
    float min_score;
    uint32_t fixed_width, fixed_mask;  // TKSPMV_FIXED: bits per value and the mask of the top fixed_width bits
    uint32_t *gmax;  // [MAX_GM*64] order keys of the group maxima (zero beyond n_groups_pub)
    uint32_t *tau_g; // one word: order key of the broadcast threshold (monotone, atomic max)
    uint32_t tau_possible;  // 1: at least k publishing groups own rows, so a threshold can form (else nobody waits for one)
    uint32_t n_reducers;  // workgroups [0, n_reducers) reduce gmax -> tau_g; the others only read tau_g
    unsigned long long *wg_cand;  // [grid][WG_SLOTS] packed {score bits | row << 32}; unused slots: row SLOT_INVALID
    unsigned long long *ovf_cand;
    uint32_t *ovf_count;
    uint32_t ovf_cap;
    // Flow control of an overflow list that several queries of one launch share (batch kernel; 0 elsewhere): nothing is
    // appended before the LDS word at byte address ovf_gate_lds shows ovf_need -- the selections of the list's earlier users have
    // finished. The word is a copy the workgroup's server wave keeps of the list's epoch in global memory: a streaming wave that
    // went to memory for it would wait for its packet loads as well (the counter retires in order) -- measured, 4 % of a query at
    // 3M rows. (mutable: a wave that has seen the gate open clears its copy -- the word never goes back.)
    mutable uint32_t ovf_gate_lds;
    uint32_t ovf_need;
    uint32_t fused;  // 1: the last workgroup to finish runs the selection (no second launch)
    // 1: deferred selection. Workgroup 0 of this launch selects the PREVIOUS query's top-k (its survivors sit in
    // the other exchange-state set, complete and visible since that launch ended) and exits; workgroups 1..grid-1
    // stream the current query and end with the flush. No ticket, no second launch, nothing on the critical path.
    uint32_t deferred;
    float *unit_inv_out;  // 1 / (score units per 1.0) of this query, for a selection that runs in a later launch
    float *scores;  // SCORES variant only
    unsigned long long *trace;   // optional (TKSPMV_TRACE=1): per-wave s_memrealtime stamps, [grid+1][9 waves][8]
    unsigned long long *stamps;  // optional (TKSPMV_STAMPS=1): s_memtime stamps of the selection tail, last workgroup
    unsigned long long *dbg;  // optional counters (TKSPMV_STATS=1): [0] slow-path executions, [1] appended rows
};

// First packet and packet count of wave partition q (q < n_parts). (Scalars, not the parameter block by reference: a block whose
// address is taken ends up in scratch memory.)
__device__ __forceinline__ void partition_range(const uint32_t uni_ppp, const uint32_t uni_last, const uint32_t n_parts, const uint32_t *part_first,
                                                const uint32_t *part_count, uint32_t q, uint32_t &p0, uint32_t &np) {
    if (uni_ppp != 0u) {
        p0 = q * uni_ppp;
        np = q + 1u < n_parts ? uni_ppp : uni_last;
    } else {
        p0 = part_first[q];
        np = part_count[q];
    }
}
#define TKSPMV_PARTITION_RANGE(P, q, p0, np) partition_range((P).uni_ppp, (P).uni_last, (P).n_parts, (P).part_first, (P).part_count, q, p0, np)

constexpr int MISC_CAND_CNT = 0, MISC_TAU = 1, MISC_TAUKEY = 4 /* local thresholds: order key of the largest threshold formed in this workgroup */, MISC_DONE = 5, MISC_XMAX = 6, MISC_SLOW_CNT = 7,
              MISC_GRPMAX = 8 /* [8] */, MISC_PUBLISHED = 16 /* [8] */, MISC_BOUND = 24 /* local thresholds: what was dropped lies below */, MISC_WORDS = 32;  // <= 8 groups per workgroup
// Private candidate list of a streaming wave (entries in LDS): 256 while x is small, 128 when x itself takes 64 KiB
// (two workgroups must still fit the CU's 160 KiB).
template <int XCOLS>
struct ListGeom {
    static constexpr uint32_t WAVE_CAP = XCOLS <= 1024 ? 256u : 128u;
    static constexpr uint32_t CAND_CAP = 8u * WAVE_CAP;  // per workgroup: up to 8 streaming waves
};
constexpr uint32_t WG_SLOTS = 8;              // fixed result slots every workgroup writes (no count round trip)
constexpr uint32_t SLOT_INVALID = 0xFFFFFFFFu;  // row id of an unused slot

// One lane's share of a packet. VT = value type of the stream: 0 = C fp32 values; 1 = C Q1.7 values packed four to a
// dword; 2 = C fp16 values packed two to a dword.
// QM (kernel template parameter): 0 = fp32, 1 = Q1.7 strict (8-bit wrapping sums, the FPGA's real_type), 2 = Q1.7 values
// with x block-scaled by a power of two per query and exact wide accumulation, 3 = fp16 values, fp32 x, fp32 arithmetic
// (the CUDA comparator's half mode, -a: host_spmv_topk_csr_gpu.cu:132-136,152-160), 4 = fixed point of W bits (the
// FPGA's real_type for any FIXED_WIDTH): values and x as left-aligned Q1.31 words, integer products and sums.
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x2_a4 __attribute__((ext_vector_type(2), aligned(4)));  // a dwordx2 at a 4-byte boundary
// QM 5 = Q1.7 values (rounded to nearest) dequantised to fp32 (v_cvt_f32_ubyteN: one VALU per entry), fp32 x held in LDS
// pre-scaled by 2^-7 (exact), fp32 products and sums: the byte stream of the Q1.7 modes with the arithmetic of the fp32
// path (TKSPMV_Q1_7_F32, BASELINE configs[4]).
// QM 6 = fixed point of at most 20 bits, bit-packed (wbscsr.hpp FIXED20): one dword per entry carrying value, column and
// flags; the arithmetic is QM 4's with both factors as 20-bit integers.
// QM 7 = fp32 exactly like QM 0, the column words travelling as 12 bits each in a split plane (wbscsr.hpp F32C12): value type 4.
// QM 8 = fixed point of 21..26 bits in five bytes per entry (wbscsr.hpp FIXED26): value type 6; the arithmetic is QM 4's.
constexpr int value_type_of(int QM) { return QM == 8 ? 6 : (QM == 7 ? 4 : (QM == 6 ? 3 : (QM == 3 ? 2 : ((QM == 1 || QM == 2 || QM == 5) ? 1 : 0)))); }  // QM 4: one u32 per value, loaded like fp32
// Byte b (0..3) of a dword as a float: v_cvt_f32_ubyte0..3.
template <int B>
__device__ __forceinline__ float ubyte_to_float(uint32_t w) {
    float r;
    if (B == 0) asm("v_cvt_f32_ubyte0 %0, %1" : "=v"(r) : "v"(w));
    else if (B == 1) asm("v_cvt_f32_ubyte1 %0, %1" : "=v"(r) : "v"(w));
    else if (B == 2) asm("v_cvt_f32_ubyte2 %0, %1" : "=v"(r) : "v"(w));
    else asm("v_cvt_f32_ubyte3 %0, %1" : "=v"(r) : "v"(w));
    return r;
}
__device__ __forceinline__ float ubyte_to_float(uint32_t w, int b) {
    return b == 0 ? ubyte_to_float<0>(w) : (b == 1 ? ubyte_to_float<1>(w) : (b == 2 ? ubyte_to_float<2>(w) : ubyte_to_float<3>(w)));
}
constexpr float Q17_UNIT = 0.0078125f;  // 2^-7

// A wave-uniform word of read-only memory through the SCALAR cache (s_load_dword): the row base of a packet is the same for
// all 64 lanes and only the candidate path looks at it. As a vector load it was one of four vector-memory instructions per
// packet -- and the streaming kernels turned out to be bound by exactly that: a CU gets through ~70 vector-memory
// instructions per microsecond whatever their width (the row-per-lane byte kernel ran at 67 and 71 per us and CU with two
// and with three loads per chunk, 19.6 and 27.8 us per query; the wave-BSCSR batch kernel at 68 with four per packet).
typedef __attribute__((address_space(4))) const uint32_t scalar_cu32;
__device__ __forceinline__ uint32_t scalar_load(const uint32_t *uniform_ptr) {
    return *(scalar_cu32 *)(uintptr_t)uniform_ptr;
}

// The packet stream is read once per query: nontemporal loads (a plain read kernel over the same bytes gains 12 %
// from them when the stream comes from HBM, tools/stream_probe.hip).
template <int C, int VT>
struct Pkt {
    float v[(VT == 0 || VT == 3 || VT == 4 || VT == 6) ? C : 1];  // VT 3, 6: the packed dwords (value | column bits | flags); VT 6: cw[0] = E (column bits 9..4)
    uint32_t vq[(VT == 1 || VT == 5) ? C / 4 : (VT == 2 ? C / 2 : 1)];  // VT 5: byte values with 12-bit column words (row-per-lane chunks)
    uint32_t cw[C / 2];  // VT 4: the two dwords of the pair's 12-byte block that hold the lane's A and the pair's B (split_ab below)
};

template <int C, int VT>
__device__ __forceinline__ void load_packet(const uint8_t *__restrict__ pk, uint32_t lane, Pkt<C, VT> &o) {
#pragma unroll
    for (int q = 0; q < C / 4; ++q) {
        if (VT == 3) {
            const f32x4 f = __builtin_nontemporal_load(reinterpret_cast<const f32x4 *>(pk + q * 1024 + lane * 16));
            o.v[VT == 3 ? 4 * q + 0 : 0] = f.x;
            o.v[VT == 3 ? 4 * q + 1 : 0] = f.y;
            o.v[VT == 3 ? 4 * q + 2 : 0] = f.z;
            o.v[VT == 3 ? 4 * q + 3 : 0] = f.w;
        } else if (VT == 6) {  // FIXED26: the lane's four dwords D_j and its dword E
            const f32x4 f = __builtin_nontemporal_load(reinterpret_cast<const f32x4 *>(pk + lane * 16));
            o.v[VT == 6 ? 0 : 0] = f.x;
            o.v[VT == 6 ? 1 : 0] = f.y;
            o.v[VT == 6 ? 2 : 0] = f.z;
            o.v[VT == 6 ? 3 : 0] = f.w;
            o.cw[0] = __builtin_nontemporal_load(reinterpret_cast<const uint32_t *>(pk + 1024 + lane * 4));
        } else if (VT == 4) {
            const f32x4 f = __builtin_nontemporal_load(reinterpret_cast<const f32x4 *>(pk + q * 1024 + lane * 16));
            o.v[VT == 4 ? 4 * q + 0 : 0] = f.x;
            o.v[VT == 4 ? 4 * q + 1 : 0] = f.y;
            o.v[VT == 4 ? 4 * q + 2 : 0] = f.z;
            o.v[VT == 4 ? 4 * q + 3 : 0] = f.w;
            // split 12-bit plane (wbscsr.hpp colw12s_*), 12 bytes per pair of lanes [A_even][B_even | B_odd << 16][A_odd]: ONE
            // dwordx2 at a 4-byte boundary per lane -- dwords 0-1 on the even lane, 1-2 on the odd one. cw[0] / cw[1] hold the
            // two dwords as loaded: reduce_packet picks A and B out of them by the lane's parity.
            const u32x2_a4 c = __builtin_nontemporal_load(reinterpret_cast<const u32x2_a4 *>(pk + C * 256 + q * 384 + (lane >> 1) * 12 + (lane & 1u) * 4));
            o.cw[2 * q + 0] = c.x;
            o.cw[2 * q + 1] = c.y;
        } else if (VT == 5) {
            o.vq[VT == 5 ? q : 0] = __builtin_nontemporal_load(reinterpret_cast<const uint32_t *>(pk + q * 256 + lane * 4));
            // Back-to-back 12-bit words, two lanes sharing three dwords: one dwordx2 at a 4-byte boundary per lane. Two other
            // arrangements were measured on BASELINE configs[4] (19.6 us per query as it stands): the split plane of the
            // wave-BSCSR packets (one dword + one halfword per lane: three loads per chunk, 9 of 29 vector instructions fewer)
            // took 27.8 us, and a pair-interleaved chunk read with ONE dwordx4 at a 4-byte boundary per lane 37.0 us -- the
            // kernel is bound by what the vector-memory pipeline does per load (an unaligned wide load counts several
            // times), not by its vector instructions.
            const u32x2_a4 c = __builtin_nontemporal_load(reinterpret_cast<const u32x2_a4 *>(pk + C * 64 + q * 384 + (lane >> 1) * 12 + (lane & 1u) * 4));
            o.cw[2 * q + 0] = c.x;
            o.cw[2 * q + 1] = c.y;
        } else if (VT == 1) {
            o.vq[VT == 1 ? q : 0] = __builtin_nontemporal_load(reinterpret_cast<const uint32_t *>(pk + q * 256 + lane * 4));
            const u32x2 c = __builtin_nontemporal_load(reinterpret_cast<const u32x2 *>(pk + C * 64 + q * 512 + lane * 8));
            o.cw[2 * q + 0] = c.x;
            o.cw[2 * q + 1] = c.y;
        } else if (VT == 2) {
            const u32x2 hv = __builtin_nontemporal_load(reinterpret_cast<const u32x2 *>(pk + q * 512 + lane * 8));
            o.vq[VT == 2 ? 2 * q + 0 : 0] = hv.x;
            o.vq[VT == 2 ? 2 * q + 1 : 0] = hv.y;
            const u32x2 c = __builtin_nontemporal_load(reinterpret_cast<const u32x2 *>(pk + C * 128 + q * 512 + lane * 8));
            o.cw[2 * q + 0] = c.x;
            o.cw[2 * q + 1] = c.y;
        } else {
            const f32x4 f = __builtin_nontemporal_load(reinterpret_cast<const f32x4 *>(pk + q * 1024 + lane * 16));
            o.v[VT == 0 ? 4 * q + 0 : 0] = f.x;
            o.v[VT == 0 ? 4 * q + 1 : 0] = f.y;
            o.v[VT == 0 ? 4 * q + 2 : 0] = f.z;
            o.v[VT == 0 ? 4 * q + 3 : 0] = f.w;
            const u32x2 c = __builtin_nontemporal_load(reinterpret_cast<const u32x2 *>(pk + C * 256 + q * 512 + lane * 8));
            o.cw[2 * q + 0] = c.x;
            o.cw[2 * q + 1] = c.y;
        }
    }
}

// The same packet through BUFFER loads (round 5; fp32 streams of 4 entries per lane: value types 0 and 4). A buffer load takes its
// address as resource (4 SGPRs: the partition's first byte in the stream copy of the query, its length) + SGPR byte offset (the
// packet) + VGPR byte offset (the lane's share, the same for every packet): nothing is added per packet on the vector unit -- the
// flat form spent three 64-bit vector adds per packet on the two addresses --, and the packet pointer is ONE scalar add. Loads
// beyond the partition's length return 0 (never requested: the kernels clamp to the last packet).
struct LaneOffsets {
    uint32_t v, c;  // byte offsets of the lane's 16 bytes of values / of its column words inside a packet
};
template <int C, int VT>
__device__ __forceinline__ LaneOffsets lane_offsets(uint32_t lane) {
    static_assert(C == 4 && (VT == 0 || VT == 4), "buffer-load packets: fp32 values, 4 entries per lane");
    LaneOffsets o;
    o.v = lane * 16u;
    o.c = VT == 4 ? (uint32_t)C * 256u + (lane >> 1) * 12u + (lane & 1u) * 4u : (uint32_t)C * 256u + lane * 8u;
    return o;
}
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ __amdgpu_buffer_rsrc_t stream_resource(const uint8_t *first_byte, uint32_t bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t *>(first_byte), 0, (int)bytes, 0x00020000);  // (raw buffer, 32-bit data format)
}
template <int C, int VT>
__device__ __forceinline__ void load_packet_buf(__amdgpu_buffer_rsrc_t rsrc, uint32_t packet_off, const LaneOffsets &lo, Pkt<C, VT> &o) {
    const u32x4 f = __builtin_amdgcn_raw_buffer_load_b128(rsrc, lo.v, packet_off, 2);  // (aux 2: non-temporal, like the flat form)
    o.v[0] = __uint_as_float(f.x);
    o.v[VT == 0 || VT == 4 ? 1 : 0] = __uint_as_float(f.y);
    o.v[VT == 0 || VT == 4 ? 2 : 0] = __uint_as_float(f.z);
    o.v[VT == 0 || VT == 4 ? 3 : 0] = __uint_as_float(f.w);
    const u32x2 c = __builtin_amdgcn_raw_buffer_load_b64(rsrc, lo.c, packet_off, 2);
    o.cw[0] = c.x;
    o.cw[1] = c.y;
}

// Q1.7 helpers (restating ap_ufixed<8,1,AP_TRN_ZERO>, fpga_types.hpp:20: 1 integer + 7 fraction bits, truncation).
// Conversion from float saturates at the top of the range (the HLS type would wrap there; inputs are expected in
// [0, 2)). Products are truncated to Q1.7 and wrap to 8 bits; sums wrap to 8 bits (mod 2.0).
__device__ __forceinline__ uint32_t to_q1_7_dev(float v) {
    const float s = fminf(fmaxf(v * 128.0f, 0.0f), 255.0f);  // NaN -> 0
    return (uint32_t)s;                                       // truncation
}
// Generic width (wbscsr.hpp to_fixed): W bits, 1 integer bit, left-aligned in a u32; truncation, saturation at the top.
__device__ __forceinline__ uint32_t to_fixed_dev(float v, uint32_t W) {
    if (!(v > 0.0f)) return 0u;
    const float s = v * (float)(1u << (W - 1u));
    const float top = W == 32u ? 4294967296.0f : (float)(1u << W);
    const uint32_t q = s >= top ? (W == 32u ? 0xFFFFFFFFu : (1u << W) - 1u) : (uint32_t)s;
    return q << (32u - W);
}
__device__ __forceinline__ float q17_wrap(float units) {  // units = exact integer sum held in fp32
    return (float)(((uint32_t)units) & 255u);
}

}  // namespace tkspmv
