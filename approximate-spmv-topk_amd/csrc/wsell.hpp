// wsell.hpp -- "wave-sliced ELL": the second packet layout, built for passes that serve SEVERAL queries at once
// (tkspmv_enqueue_multi).
//
// In the wave-BSCSR layout (wbscsr.hpp) a row's entries lie across lanes and every packet needs a segmented scan per
// query (~12 VALU instructions per non-zero); that is what bounds a pass once its loads are shared by several queries.
// Here ONE LANE OWNS ONE ROW, so a non-zero costs one LDS read, one multiply and one add per query and nothing else:
//
//   * the non-empty rows are sorted by length (descending, ties by row id) and cut into SLICES of 64 lanes; a slice is
//     stored as ceil(longest lane / 4) CHUNKS; a chunk is the 1536-byte packet of wbscsr.hpp --
//     [64 lanes x 4 fp32 values][64 lanes x 4 u16 column words] -- holding entries 4c .. 4c+3 of lane l's row, so the
//     streaming loop, its loads and its prefetch pipeline are the ones of the other kernels;
//   * a row shorter than its slice is padded with (value +0.0, column PAD_NEUTRAL): the kernel keeps -0.0f in that slot
//     of its LDS copy of x, the product is -0.0 and s + (-0.0) == s for every s, bit for bit. A lane without a row
//     (only the very last slice can have some) starts with (value -inf, column PAD_ONE): that slot holds 1.0f, the
//     lane's sum is -inf, it never passes a threshold and never raises a published maximum;
//   * a lane holds at most SELL_SEG = 64 entries (16 chunks), so no slice is longer than a wave's share of the matrix and
//     the partitions can be balanced: a LONGER row (0.35 % of the rows of the BASELINE matrix, 10 % at 40 entries per row)
//     is cut into equal segments of sell_segment_length(len) <= 64 entries on ADJACENT lanes of one slice (80 entries =
//     40 + 40: cut at 64 the second lane would carry 16 entries and 48 of padding -- 20 % of configs[4]'s stream); when
//     the slice ends, the segment sums are added left to right (((s0 + s1) + s2) ...) and the score lands on the last lane;
//   * column word: bits 15..2 = column (so word & 0xFFFC is the LDS byte offset of x[col]). The two low bits are flags in
//     the LAST chunk of a slice only: bit 0 of every lane's first word = "last chunk of its slice" (wave-uniform); the
//     low two bits of words 1, 2, 3 = the lane's segment index (0 for an ordinary row), 2 bits each;
//   * row ids are NOT in the stream: slice_rows[slice][lane] (4 B per lane; the row id on the lane that ends up with the
//     row's score, SELL_NO_ROW elsewhere) is read only when a lane's score clears the threshold, which is rare;
//   * slices go to the wave partitions longest-processing-time-first (each slice, longest first, to the partition with the
//     fewest chunks so far): partitions within one slice of each other, long and short rows in every partition (equally
//     strong published group maxima for the threshold exchange). A partition's slices are contiguous in the stream.
//
// A row's sum is accumulated entry by entry in the row's own order, which is the order of the reference's gold
// (spmv_coo_gold_top_k, gold_algorithms.hpp:188-246: sequential fp32): for rows of at most 64 entries the scores are
// bit-identical to the gold's; longer rows are summed segment-wise as stated (oracle_scores_f32_segmented).
#pragma once
#include <cstdint>
#include <string>
#include <vector>

namespace tkspmv {

constexpr uint32_t SELL_XCOLS = 1024;                  // columns the multi-query kernel is built for
constexpr uint32_t SELL_SEG = 64;                      // entries a lane holds at most; longer rows span adjacent lanes
constexpr uint32_t SELL_PAD_NEUTRAL = SELL_XCOLS;      // LDS slot holding -0.0f
constexpr uint32_t SELL_PAD_ONE = SELL_XCOLS + 1;      // LDS slot holding 1.0f
constexpr uint32_t SELL_NO_ROW = 0xFFFFFFFFu;
constexpr uint16_t SELL_LAST_CHUNK = 0x0001u;
// Entries per lane of a row of `len` entries: the row itself up to SELL_SEG; a longer row is cut into ceil(len / SELL_SEG)
// nearly equal segments, rounded up to whole chunks of 4 (the last segment takes what is left).
inline uint32_t sell_segment_length(uint32_t len) {
    if (len <= SELL_SEG) return len;
    const uint32_t nseg = (len + SELL_SEG - 1) / SELL_SEG;
    return ((len + nseg - 1) / nseg + 3u) & ~3u;
}

// Value type of the chunks: fp32 (1536-byte chunks) or Q1.7 bytes rounded to nearest (768-byte chunks: [64 lanes x 4 u8]
// [64 lanes x 4 u16 column words]; TKSPMV_Q1_7_F32). With byte values the kernel's LDS copy of x is pre-scaled by 2^-7 and
// a lane without a row starts with (byte 1, column PAD_ONE) where that slot holds -inf instead of (value -inf, slot 1.0f).
enum class SellValues : uint32_t { F32 = 4, Q1_7_RND = 1 };  // = bytes per value

// Column words of 12 bits (round 2, byte chunks only -- BASELINE configs[4]): with at most 1022 columns a column word
// (10 bits of column or padding slot, 2 flags) needs 12 bits: the two padding slots move to columns 1022 and 1023, the 256
// words of a chunk lie back to back (entry t at bit 12 t, as in wbscsr.hpp's F32C12) and a byte chunk is 256 + 384 = 640
// bytes instead of 768: 2.5 bytes per entry. TKSPMV_SELL_C12=0 keeps 16-bit words.
constexpr uint32_t SELL_C12_MAX_COLS = 1022;
bool sell_c12_wanted();  // (the environment switch)

struct SellMatrix {
    uint32_t rows = 0, cols = 0;
    uint64_t nnz = 0;
    uint32_t n_slices = 0, n_chunks = 0;
    SellValues values = SellValues::F32;
    uint32_t cw_bits = 16;                   // bits per column word: 16, or 12 (byte chunks of at most 1022 columns)
    uint32_t pad_neutral = SELL_PAD_NEUTRAL, pad_one = SELL_PAD_ONE;  // padding slots (1022 / 1023 with 12-bit words)
    uint32_t packet_bytes = 1536;           // 256 * bytes per value + 256 * cw_bits / 8
    uint64_t padded_entries = 0;            // n_chunks * 256
    std::vector<uint8_t> packets;           // n_chunks * packet_bytes
    std::vector<uint32_t> slice_rows;       // [n_slices][64], stream order
    std::vector<uint32_t> part_first;       // [n_parts] first chunk
    std::vector<uint32_t> part_count;       // [n_parts] chunks
    std::vector<uint32_t> part_slice0;      // [n_parts] index of the partition's first slice (stream order)
    uint64_t stream_bytes() const { return (uint64_t)n_chunks * packet_bytes; }
};

// The layout decisions of a pack -- which segment of which row sits on which lane of which slice, and where every slice
// lies in the stream -- without the nnz-sized fill: shared by the host packer (fill_wsell_host) and the device packer
// (device_pack.hip: sell_scatter_kernel), so the two produce the same bytes by construction of everything but the fill.
struct SellLane {
    uint32_t row, first, n, depth;  // entries [first, first + n) of `row`; depth = index of the segment in its row
    uint32_t tail;                  // 1: last segment of its row (the lane that ends up with the row's score)
};
struct SellPlan {
    std::vector<uint64_t> start;         // [rows + 1] offset of every row's entries in the COO
    std::vector<SellLane> lanes;         // 64 per slice, in the SORTED slice order; row == SELL_NO_ROW: lane without a row
    std::vector<uint32_t> stream_slice;  // [n_slices] sorted-order index of the slice at stream position so
    std::vector<uint32_t> chunk0;        // [n_slices] first chunk of the slice at stream position so
    std::vector<uint32_t> n_chunks_of;   // [n_slices] its chunk count
};
// Fills out everything of `out` but `packets` (validated like pack_wbscsr). Returns an empty string on success.
std::string plan_wsell(uint32_t rows, uint32_t cols, uint64_t nnz, const uint32_t *row, const uint32_t *col,
                       uint32_t n_partitions_hint, SellValues values, SellPlan &plan, SellMatrix &out);
void fill_wsell_host(const SellPlan &plan, const uint32_t *col, const float *val, SellMatrix &out);

// Packs a row-sorted COO (validated like pack_wbscsr). Returns an empty string on success.
std::string pack_wsell(uint32_t rows, uint32_t cols, uint64_t nnz, const uint32_t *row, const uint32_t *col, const float *val,
                       uint32_t n_partitions_hint, SellMatrix &out, SellValues values = SellValues::F32);

// Inverse (tests): the rows in stream order with their entries (padding dropped).
void decode_wsell(const SellMatrix &sm, std::vector<uint32_t> &row, std::vector<uint32_t> &col, std::vector<float> &val);

}  // namespace tkspmv
