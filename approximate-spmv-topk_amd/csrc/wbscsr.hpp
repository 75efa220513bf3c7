// wbscsr.hpp -- "wave block-streaming CSR": the packet layout the fused kernel streams from HBM.
//
// Counterpart of the reference's BSCSR 512-bit packets (src/common/types.hpp:57-79,
// src/fpga/src/ip/fpga_utils.hpp:264-365, packer src/fpga/src/host_spmv_bscsr.cpp:133-248), re-derived for
// 64-lane wavefronts and 128-byte HBM3E transactions instead of 32 pseudo-channels x 512-bit words:
//
//   * a PACKET holds PE = 64 * C consecutive entries of the row-sorted matrix (C = entries per lane, 4 or 8);
//     lane l of the wave owns entries [l*C, l*C+C) so every lane loads 16 contiguous bytes of values
//     (global_load_dwordx4 => 1 KiB per wave instruction) and 8 (C=4) or 16 (C=8) bytes of column words;
//   * packet bytes: [PE values (fp32, fp16, u8 for Q1.7, or u32 for generic fixed point)] [PE x u16 column words],
//     packets are 128-B aligned;
//   * a column word is  bit0 = ROW_END (this entry is the last of its row; the reference stores 4-bit
//     cumulative row-end offsets + the xf bit instead), bit1 = SKIP (placeholder entry of an empty row:
//     keeps row counting implicit, never becomes a candidate), bits 15..2 = column (reference: 10 bits),
//     i.e. (word & 0xFFFC) is directly the byte offset of x[col] in the LDS copy of the fp32 query;
//   * for C = 8 the lane's entries are stored as two planes of 4 (plane q at +q*1024 B values, +q*512 B
//     column words) so that every wave load instruction still touches one dense 1 KiB / 512 B span;
//   * no row ids are stored per entry: rows are counted from pkt_row[p] (row id of the first row that
//     ends in packet p; 4 B per packet) -- the reference counts rows in its "summary" stage;
//   * rows are grouped into WAVE PARTITIONS (contiguous row ranges, one per wave, balanced by entries and
//     padded to whole packets so a partition starts and ends on a row boundary). The reference's
//     32 partitions -> HBM channels becomes n_CU * waves_per_CU partitions -> waves.
#pragma once
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "options.hpp"

// The conversions and the slot arithmetic below are shared, source for source, by the host packer (wbscsr.cpp) and the
// device packer (device_pack.hip): one definition, so the two cannot drift apart.
#if defined(__HIP__)
#define TKSPMV_HD __host__ __device__
#else
#define TKSPMV_HD
#endif

namespace tkspmv {

constexpr uint16_t COLW_ROW_END = 0x0001u;
constexpr uint16_t COLW_SKIP = 0x0002u;
constexpr uint16_t COLW_COL_SHIFT = 2;
constexpr uint32_t MAX_COLS = 1u << 14;
constexpr uint32_t WAVE = 64;

// Position of stream slot s (0..PE-1, row-major order of the matrix) inside a packet: lane = s / C owns
// C consecutive slots; they are stored in planes of 4.
TKSPMV_HD inline uint32_t slot_to_index(uint32_t s, uint32_t C) {
    uint32_t lane = s / C, j = s % C;
    return (j >> 2) * (WAVE * 4) + lane * 4 + (j & 3);
}

// value type in the packet stream; FIXED = unsigned fixed point of `fixed_width` bits (1 integer bit), one u32 per value
// Q1_7 = truncated (ap_ufixed<8,1,AP_TRN_ZERO>), Q1_7_RND = rounded to nearest (ap_ufixed<8,1,AP_RND>): same bytes per value.
// FIXED20 = fixed point of at most 20 bits with at most 1024 columns, ONE DWORD PER ENTRY (the reference's point at 20 / 25
// bits is more entries per 512-bit transaction: B = 15 at 20 bits against 11 at 32, types.hpp:57-79): bits 31..12 = the
// value (the top 20 bits of the left-aligned Q1.31 word of wbscsr's fixed-point type), bits 11..2 = the column -- so
// (word & 0xFFC) is the LDS byte offset of x[col] -- and bits 1..0 = the ROW_END / SKIP flags of a column word. A packet
// is [64 * C dwords] = 4 B per entry; there is no separate column-word region.
// F32C12 = fp32 values with 12-BIT column words (round 2): with at most 1024 columns a column word -- 10 bits of
// column, 2 flags -- needs 12 bits, not 16; a packet is [64 * C floats][64 * C x 12 bits] = 5.5 B per entry instead of 6,
// 8.3 % fewer bytes from HBM for the same arithmetic on the same fp32 values (the reference packs its BSCSR packets to
// the bit for the same reason: types.hpp:57-79). The 12 bits of a lane's 4 entries are split over one dword and one
// halfword per lane (colw12s_* below; round 2 had the words back to back, which the row-per-lane layout of wsell.hpp
// still uses: colw12_*). The default for TKSPMV_F32 with cols <= 1024 and 4 entries per lane (TKSPMV_F32_C12=0 keeps
// 16-bit column words); results are bit-identical either way.
// FIXED26 = fixed point of 21..26 bits with at most 1024 columns in 5 BYTES PER ENTRY (round 3; the reference's 21-, 25- and
// 26-bit builds, test_spmv_topk.py:42-47: B = 13 entries per 512 bits at 25 bits against 11 at 32, types.hpp:57-79): 26 bits
// of value + 10 of column + 2 flags = 38 bits. A lane's 4 entries are 5 dwords -- 4 in a plane of 16 bytes per lane, 1 in a
// plane of 4 bytes per lane (a packet is 1024 + 256 = 1280 bytes):
//   D_j = value (bits 31..6: the top 26 bits of its left-aligned Q1.31 word) | column bits 3..0 << 2 | SKIP << 1 | ROW_END
//   E   = column bits 9..4 of entry j at bits 6 j .. 6 j + 5
// so the value is one mask, the flags sit where FIXED20 has them and the LDS offset of x[col] is (D_j & 0x3C) | field(E) << 6.
enum class Precision : int32_t { F32 = 0, Q1_7 = 1, F16 = 3, FIXED = 4, Q1_7_RND = 5, FIXED20 = 6, F32C12 = 7, FIXED26 = 8 };
constexpr uint32_t FIXED20_MAX_WIDTH = 20, FIXED20_MAX_COLS = 1024, F32C12_MAX_COLS = 1024, FIXED26_MAX_WIDTH = 26, FIXED26_MAX_COLS = 1024;

// bytes of a packet per entry, minus the 2 of a column word (FIXED20: 20-bit value + 10-bit column + 2 flags in 4 bytes)
TKSPMV_HD inline uint32_t value_bytes(Precision p) {
    return (p == Precision::F32 || p == Precision::FIXED || p == Precision::F32C12) ? 4u
                                                                                      : (p == Precision::FIXED26 ? 3u : ((p == Precision::F16 || p == Precision::FIXED20) ? 2u : 1u));
}
// bytes of a packet of PE entries
TKSPMV_HD inline uint32_t packet_bytes_for(Precision p, uint32_t PE) {
    return p == Precision::F32C12 ? PE * 4u + PE * 3u / 2u : PE * (value_bytes(p) + 2u);
}
// 12-bit column words (F32C12): `planes` = start of the packet's column region, `slot` = slot_to_index() of the entry
// (plane = slot / 256, entry t = slot % 256 of the plane at bit 12 t of the plane's 384 bytes).
TKSPMV_HD inline void colw12_store(uint8_t *planes, uint32_t slot, uint16_t cw) {
    const uint32_t t = slot & 255u;
    uint8_t *b = planes + (size_t)(slot >> 8) * 384u + (t * 3u) / 2u;
    if (t & 1u) {
        b[0] = (uint8_t)((b[0] & 0x0Fu) | ((cw & 0xFu) << 4));
        b[1] = (uint8_t)(cw >> 4);
    } else {
        b[0] = (uint8_t)(cw & 0xFFu);
        b[1] = (uint8_t)((b[1] & 0xF0u) | ((cw >> 8) & 0xFu));
    }
}
TKSPMV_HD inline uint16_t colw12_load(const uint8_t *planes, uint32_t slot) {
    const uint32_t t = slot & 255u;
    const uint8_t *b = planes + (size_t)(slot >> 8) * 384u + (t * 3u) / 2u;
    return (t & 1u) ? (uint16_t)((b[0] >> 4) | ((uint16_t)b[1] << 4)) : (uint16_t)(b[0] | ((uint16_t)(b[1] & 0x0Fu) << 8));
}
// F32C12's plane is SPLIT (round 3): the 12 bits of a lane's 4 entries (entries 4l .. 4l+3) are one dword A and one halfword B,
//   A = col0 << 2 | col1 << 12 | col2 << 22 | SKIP0 | SKIP1 << 1
//   B = col3 << 2 | SKIP2 | SKIP3 << 1 | ROW_END0..3 << 12
// and the 384 bytes of a plane of 256 entries are 32 blocks of 12 bytes, one per PAIR of lanes: [A_even][B_even | B_odd << 16]
// [A_odd]. A lane loads ONE dwordx2 at a 4-byte boundary -- the even lane the block's dwords 0-1, the odd lane dwords 1-2 -- and
// has its A and the pair's B word in it. The same 12 bits per entry as the back-to-back words of round 2, arranged for the
// kernel: no funnel shifts, a column's LDS offset is one or two instructions, the four row-end masks are single-bit
// extractions of B and "this lane holds a row end" is B > 0xFFF -- 22 vector instructions fewer per packet (DESIGN.md
// section 3) with the same two vector-memory instructions per packet (a dword plane + a halfword plane, three loads, was
// measured slower: the kernels are bound by what the vector-memory pipeline does per load).
TKSPMV_HD inline uint32_t colw12s_a_offset(uint32_t lane) { return (lane >> 1) * 12u + (lane & 1u) * 8u; }        // byte offset of A
TKSPMV_HD inline uint32_t colw12s_b_offset(uint32_t lane) { return (lane >> 1) * 12u + 4u + (lane & 1u) * 2u; }   // ... of B
TKSPMV_HD inline void colw12s_bits(uint32_t j, uint16_t cw, uint32_t &a, uint32_t &b) {  // what entry j ORs into A and B
    const uint32_t col = (uint32_t)(cw >> COLW_COL_SHIFT), end = cw & COLW_ROW_END, skip = (cw & COLW_SKIP) ? 1u : 0u;
    a = j == 0u ? ((col << 2) | skip) : (j == 1u ? ((col << 12) | (skip << 1)) : (j == 2u ? (col << 22) : 0u));
    b = (j == 2u ? skip : (j == 3u ? ((col << 2) | (skip << 1)) : 0u)) | (end << (12u + j));
}
inline void colw12s_store(uint8_t *planes, uint32_t slot, uint16_t cw) {  // into a zeroed plane
    const uint32_t t = slot & 255u, lane = t >> 2;
    uint8_t *pl = planes + (size_t)(slot >> 8) * 384u;
    uint32_t a, b, A;
    uint16_t Bv;
    colw12s_bits(t & 3u, cw, a, b);
    std::memcpy(&A, pl + colw12s_a_offset(lane), 4);
    std::memcpy(&Bv, pl + colw12s_b_offset(lane), 2);
    A |= a;
    Bv = (uint16_t)(Bv | b);
    std::memcpy(pl + colw12s_a_offset(lane), &A, 4);
    std::memcpy(pl + colw12s_b_offset(lane), &Bv, 2);
}
inline uint16_t colw12s_load(const uint8_t *planes, uint32_t slot) {
    const uint32_t t = slot & 255u, lane = t >> 2, j = t & 3u;
    const uint8_t *pl = planes + (size_t)(slot >> 8) * 384u;
    uint32_t A;
    uint16_t Bv;
    std::memcpy(&A, pl + colw12s_a_offset(lane), 4);
    std::memcpy(&Bv, pl + colw12s_b_offset(lane), 2);
    const uint32_t col = j == 0u ? (A >> 2) & 1023u : (j == 1u ? (A >> 12) & 1023u : (j == 2u ? (A >> 22) & 1023u : ((uint32_t)Bv >> 2) & 1023u));
    const uint32_t skip = j == 0u ? (A & 1u) : (j == 1u ? ((A >> 1) & 1u) : (j == 2u ? (Bv & 1u) : ((Bv >> 1) & 1u)));
    const uint32_t end = ((uint32_t)Bv >> (12u + j)) & 1u;
    return (uint16_t)((col << COLW_COL_SHIFT) | (skip ? COLW_SKIP : 0u) | (end ? COLW_ROW_END : 0u));
}
TKSPMV_HD inline uint32_t fixed20_word(uint32_t q_left_aligned, uint32_t col, uint32_t flags) {
    return (q_left_aligned & 0xFFFFF000u) | (col << 2) | flags;
}
TKSPMV_HD inline uint32_t fixed26_d(uint32_t q_left_aligned, uint32_t col, uint32_t flags) {  // the entry's dword of the 16-byte plane
    return (q_left_aligned & 0xFFFFFFC0u) | ((col & 15u) << 2) | flags;
}
TKSPMV_HD inline uint32_t fixed26_e(uint32_t j, uint32_t col) { return (col >> 4) << (6u * j); }  // what entry j ORs into the lane's E
// Value type of the stream for a tkspmv_precision (TKSPMV_Q1_7 and TKSPMV_Q1_7_WIDE share the truncated Q1.7 stream;
// TKSPMV_Q1_7_F32 streams Q1.7 values rounded to nearest).
inline Precision stream_precision(int32_t api_precision, uint32_t fixed_width = 0, uint32_t cols = 0, uint32_t C = 0) {
    // fp32 values over few columns travel with 12-bit column words (C = entries per lane; 0: not known, keep 16 bits)
    // (TKSPMV_F32_C12=0 keeps 16-bit column words: 8.3 % more bytes, 2.7 % slower back-to-back queries)
    if (api_precision == 0 && C == 4 && cols >= 1 && cols <= F32C12_MAX_COLS) {
        const char *f = opt("F32_C12");
        if (!f || atoi(f) != 0) return Precision::F32C12;
    }
    // narrow fixed point with few columns travels bit-packed (TKSPMV_FIXED_UNPACKED=1 keeps one u32 per value + a column word)
    if (api_precision == 4 && fixed_width >= 8 && fixed_width <= FIXED20_MAX_WIDTH && cols >= 1 && cols <= FIXED20_MAX_COLS &&
        opt("FIXED_UNPACKED") == nullptr)
        return Precision::FIXED20;
    // 21..26 bits: five bytes per entry (4 entries per lane; C = 0: not known, the one-u32-per-value stream)
    if (api_precision == 4 && C == 4 && fixed_width > FIXED20_MAX_WIDTH && fixed_width <= FIXED26_MAX_WIDTH && cols >= 1 && cols <= FIXED26_MAX_COLS &&
        opt("FIXED_UNPACKED") == nullptr)
        return Precision::FIXED26;
    switch (api_precision) {
        case 0: return Precision::F32;
        case 3: return Precision::F16;
        case 4: return Precision::FIXED;
        case 5: return Precision::Q1_7_RND;
        default: return Precision::Q1_7;
    }
}

// IEEE binary16 <-> binary32, round to nearest even, overflow to infinity (the CUDA comparator's half mode converts
// its values the same way: host_spmv_topk_csr_gpu.cu:132-136,152-160 with __float2half).
TKSPMV_HD inline uint16_t to_half(float f) {
    uint32_t x;
    __builtin_memcpy(&x, &f, 4);
    const uint16_t sign = (uint16_t)((x >> 16) & 0x8000u);
    x &= 0x7FFFFFFFu;
    if (x >= 0x7F800000u) return (uint16_t)(sign | (x > 0x7F800000u ? 0x7E00u : 0x7C00u));  // NaN / infinity
    if (x >= 0x477FF000u) return (uint16_t)(sign | 0x7C00u);                                  // >= 65520 rounds to infinity
    uint32_t q, rem, half;
    if (x < 0x38800000u) {  // below 2^-14: subnormal half (or zero)
        if (x < 0x33000000u) return sign;  // below 2^-25 (2^-25 itself ties to even = 0)
        const uint32_t e = x >> 23, M = (x & 0x7FFFFFu) | 0x800000u, shift = 126u - e;  // 14..24
        q = M >> shift;
        rem = M & ((1u << shift) - 1u);
        half = 1u << (shift - 1u);
    } else {
        q = (((x >> 23) - 112u) << 10) | ((x & 0x7FFFFFu) >> 13);
        rem = x & 0x1FFFu;
        half = 0x1000u;
    }
    if (rem > half || (rem == half && (q & 1u))) ++q;  // a carry out of the mantissa bumps the exponent, as it should
    return (uint16_t)(sign | q);
}
inline float from_half(uint16_t hbits) {
    const uint32_t sign = (uint32_t)(hbits & 0x8000u) << 16, e = (hbits >> 10) & 31u, m = hbits & 0x3FFu;
    uint32_t x;
    if (e == 31u) {
        x = sign | 0x7F800000u | (m << 13);
    } else if (e != 0u) {
        x = sign | ((e + 112u) << 23) | (m << 13);
    } else if (m == 0u) {
        x = sign;
    } else {  // subnormal: normalise
        uint32_t mm = m, ee = 113u;
        while (!(mm & 0x400u)) {
            mm <<= 1;
            --ee;
        }
        x = sign | (ee << 23) | ((mm & 0x3FFu) << 13);
    }
    float f;
    std::memcpy(&f, &x, 4);
    return f;
}

// Unsigned fixed point with 1 integer and 7 fraction bits, truncation toward zero and saturation at the
// top of the range when converting from float (restating ap_ufixed<8,1,AP_TRN_ZERO>, fpga_types.hpp:20).
TKSPMV_HD inline uint8_t to_q1_7(float v) {
    if (!(v > 0.0f)) return 0;
    float s = v * 128.0f;
    if (s >= 255.0f) return 255;
    return (uint8_t)s;  // truncation
}
inline float from_q1_7(uint32_t q) { return (float)q * (1.0f / 128.0f); }
// The same format rounded to nearest, ties up, saturating (ap_ufixed<8,1,AP_RND,AP_SAT>): the value stream of
// TKSPMV_Q1_7_F32, whose arithmetic is fp32 -- there the quantisation of the values is the only error, and rounding
// halves it (precision@100 against the fp32 gold on BASELINE configs[4]: 0.97 rounded, 0.94 truncated).
TKSPMV_HD inline uint8_t to_q1_7_rnd(float v) {
    if (!(v > 0.0f)) return 0;
    const float s = v * 128.0f + 0.5f;  // exact for |v| < 2^16 (one binade of slack below 2^24)
    if (s >= 255.0f) return 255;
    return (uint8_t)s;
}

// Unsigned fixed point of W bits with 1 integer bit (restating ap_ufixed<W,1,AP_TRN_ZERO>, fpga_types.hpp:20, for the
// reference's FIXED_WIDTH builds: 20/21/25/26/32 bits, types.hpp:20, test_spmv_topk.py:42-47), kept LEFT-ALIGNED in a
// u32: bit 31 is the integer bit, the W-1 fraction bits follow, the low 32-W bits are zero (so every width shares one
// Q1.31 arithmetic: products are masked back to W bits, sums wrap at 2.0 by the u32 wrap). Truncation toward zero;
// conversion from float saturates at the top of the range (the HLS type would wrap; inputs are expected in [0, 2)).
TKSPMV_HD inline uint32_t fixed_mask(uint32_t W) { return ~((1u << (32u - W)) - 1u); }  // the top W bits
TKSPMV_HD inline uint32_t to_fixed(float v, uint32_t W) {
    if (!(v > 0.0f)) return 0u;                       // negative, zero, NaN
    const float s = v * (float)(1u << (W - 1u));      // exact (power of two)
    const float top = W == 32u ? 4294967296.0f : (float)(1u << W);
    const uint32_t q = s >= top ? (W == 32u ? 0xFFFFFFFFu : (1u << W) - 1u) : (uint32_t)s;  // truncation
    return q << (32u - W);
}
inline float from_fixed(uint32_t q) { return (float)q * (1.0f / 2147483648.0f); }

struct PackedMatrix {
    uint32_t rows = 0, cols = 0;
    uint64_t nnz = 0;
    Precision precision = Precision::F32;
    uint32_t fixed_width = 0;     // Precision::FIXED: bits per value (8..32); 0 otherwise
    uint32_t C = 4;               // entries per lane
    uint32_t packet_entries = 0;  // 64*C
    uint32_t packet_bytes = 0;    // packet_bytes_for(precision, packet_entries)
    uint32_t n_packets = 0;
    uint32_t packets_per_partition = 0;  // m: fill capacity (partitions holding one giant row may exceed it)
    uint64_t packed_entries = 0;         // n_packets * packet_entries
    uint64_t placeholders = 0;           // empty rows inside [0, last_row]
    std::vector<uint8_t> packets;        // n_packets * packet_bytes
    std::vector<uint32_t> pkt_row;       // [n_packets]
    std::vector<uint32_t> part_first;    // [n_parts] first packet
    std::vector<uint32_t> part_count;    // [n_parts] packets
    std::vector<uint32_t> part_row0;     // [n_parts] first row of the partition
    std::vector<uint32_t> part_rows;     // [n_parts] rows (incl. placeholders) in the partition

    uint64_t stream_bytes() const { return (uint64_t)n_packets * packet_bytes; }
    uint64_t side_bytes() const { return (uint64_t)pkt_row.size() * 4 + (uint64_t)part_first.size() * 8; }
};

// Shortest partition the packers cut, a property of the matrix alone so that every packer (host, device, tkspmv_pack, the engine)
// cuts the same partitions: 4 packets; 2 on small matrices (up to SMALL_MATRIX_PACKETS packets: ~700k rows of 20 non-zeros -- where the batch kernel's small-matrix
// settings pay, engine.hip),
// where there are fewer packets than 4 per streaming wave of a 256-CU launch and half of the waves would have nothing to stream;
// 1 up to a tenth of that.
// (TKSPMV_MIN_PACKETS / TKSPMV_SMALL_PACKETS: tuning runs.)
constexpr uint64_t SMALL_MATRIX_PACKETS = 55000;
// (up to this many packets the batch kernel runs with workgroup-local thresholds: engine.hip. Rounds 3-5a: 100000, ~1.3M rows of 20
//  non-zeros -- beyond, the device-wide exchange was as fast. With the timetable of round 5 the local kernel wins at every size
//  measured: 24.4 against 25.9 us per query at 1.5M rows, 32.9 / 34.3 at 2M, 48.6 / 52.4 at 3M, 83.7 / 86.2 at 5M, 159.4 / 172.1 at 10M.)
constexpr uint64_t LOCAL_MATRIX_PACKETS = 4000000;
uint64_t small_matrix_packets();
uint32_t min_packets_per_partition_for(uint64_t nnz, uint32_t C, uint32_t cols);

// Packs a row-sorted COO. Returns empty string on success, else an error message.
// kind: 0 ok, 1 invalid, 2 not sorted.
std::string pack_wbscsr(uint32_t rows, uint32_t cols, uint64_t nnz, const uint32_t *row, const uint32_t *col,
                        const float *val, Precision precision, uint32_t C, uint32_t n_partitions_hint,
                        uint32_t min_packets_per_partition, PackedMatrix &out, int &kind, uint32_t fixed_width = 0);

// Inverse of pack (placeholders and padding dropped). Values come back as float (fp16 / fixed point decoded).
void decode_wbscsr(const PackedMatrix &pm, std::vector<uint32_t> &row, std::vector<uint32_t> &col,
                   std::vector<float> &val);

// Binary cache of a packed matrix (".tkspmv": 128-byte header, the packet stream, the side tables, FNV-1a checksum). The
// reference parses the MatrixMarket text and re-packs on every run (utils.hpp:380-388, host_spmv_bscsr.cpp:133-248);
// a packed file is read back at disk speed. Returns an empty string on success, else an error message.
std::string save_packed(const PackedMatrix &pm, const char *path);
std::string load_packed(const char *path, PackedMatrix &pm);

}  // namespace tkspmv
