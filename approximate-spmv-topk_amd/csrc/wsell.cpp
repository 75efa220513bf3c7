// wsell.cpp -- host packer / decoder of the wave-sliced ELL layout (see wsell.hpp).
#include "wsell.hpp"

#include "wbscsr.hpp"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <limits>

namespace tkspmv {

std::string plan_wsell(uint32_t rows, uint32_t cols, uint64_t nnz, const uint32_t *row, const uint32_t *col,
                       uint32_t n_partitions_hint, SellValues values, SellPlan &plan, SellMatrix &out) {
    if (cols == 0 || cols > SELL_XCOLS) return "the multi-query layout is built for at most 1024 columns";
    if (nnz > 0 && (!row || !col)) return "row/col arrays are NULL";
    if (n_partitions_hint == 0) n_partitions_hint = 1;
    out = SellMatrix();
    plan = SellPlan();
    out.rows = rows;
    out.cols = cols;
    out.nnz = nnz;
    out.values = values;
    const uint32_t vb = (uint32_t)values;
    if (values == SellValues::Q1_7_RND && cols <= SELL_C12_MAX_COLS && sell_c12_wanted()) {
        out.cw_bits = 12;
        out.pad_neutral = 1022;
        out.pad_one = 1023;
    }
    const uint32_t PB = 256u * vb + 256u * out.cw_bits / 8u;
    out.packet_bytes = PB;
    if (nnz == 0) return "";
    for (uint64_t i = 1; i < nnz; ++i)
        if (row[i] < row[i - 1]) return "COO rows are not sorted in non-decreasing order";
    const uint32_t last_row = row[nnz - 1];
    if (last_row >= rows) return "row id out of range (>= rows)";
    std::vector<uint32_t> len((size_t)last_row + 1, 0u);
    for (uint64_t i = 0; i < nnz; ++i) {
        if (col[i] >= cols) return "column id out of range (>= cols)";
        ++len[row[i]];
    }
    std::vector<uint64_t> &start = plan.start;
    start.assign(len.size() + 1, 0);
    for (size_t r = 0; r < len.size(); ++r) start[r + 1] = start[r] + len[r];
    // Units = non-empty rows; a row of more than SELL_SEG entries takes several adjacent lanes, cut into EQUAL segments
    // (sell_segment_length: a 80-entry row becomes 40 + 40, not 64 + 16 padded to 64) and its length class is that segment
    // length, so it sorts among the ordinary rows of that length. Units by length class, longest first, ties by row id
    // (counting sort).
    uint32_t max_class = 0;
    for (uint32_t L : len) max_class = std::max(max_class, sell_segment_length(L));
    std::vector<uint64_t> bucket((size_t)max_class + 2, 0);
    for (uint32_t L : len) {
        if (!L) continue;
        if ((L + SELL_SEG - 1) / SELL_SEG > 64u) return "a row is too long for the multi-query layout (more than 4096 entries)";
        ++bucket[max_class - sell_segment_length(L) + 1];
    }
    for (size_t b = 1; b < bucket.size(); ++b) bucket[b] += bucket[b - 1];
    const uint64_t n_ne = bucket[max_class];  // rows with at least one entry
    std::vector<uint32_t> order(n_ne);
    for (uint32_t r = 0; r < (uint32_t)len.size(); ++r)
        if (len[r]) order[bucket[max_class - sell_segment_length(len[r])]++] = r;

    // Slices: 64 lanes filled in that order; a multi-lane row never straddles two slices (lanes left over stay empty).
    std::vector<SellLane> &lanes = plan.lanes;  // 64 per slice, row == SELL_NO_ROW: lane without a row
    std::vector<uint32_t> slice_chunks;         // chunks of every slice
    {
        uint32_t used = 64;  // lanes used in the open slice (64: none open)
        for (uint64_t u = 0; u < n_ne; ++u) {
            const uint32_t r = order[u], L = len[r], slen = sell_segment_length(L), nseg = (L + slen - 1) / slen;
            if (used + nseg > 64u) {
                if (used < 64u) lanes.resize(lanes.size() + (64u - used), SellLane{SELL_NO_ROW, 0, 0, 0, 0});
                slice_chunks.push_back((slen + 3) / 4);  // the slice's first lane is its longest
                used = 0;
            }
            for (uint32_t g = 0; g < nseg; ++g)
                lanes.push_back(SellLane{r, g * slen, std::min(slen, L - g * slen), g, g + 1 == nseg ? 1u : 0u});
            used += nseg;
        }
        if (used < 64u) lanes.resize(lanes.size() + (64u - used), SellLane{SELL_NO_ROW, 0, 0, 0, 0});
    }
    if (slice_chunks.size() > 0x7FFFFFFFull) return "matrix too large";
    const uint32_t n_slices = (uint32_t)slice_chunks.size();
    uint64_t n_chunks = 0;
    for (uint32_t c : slice_chunks) n_chunks += c;
    if (n_chunks > 0xFFFFFFFFull) return "matrix too large (chunk count overflows 32 bits)";

    // Slices to wave partitions, longest processing time first: the slices come longest first; each goes to the partition
    // holding the fewest chunks so far (ties: lowest partition). No slice is longer than 16 chunks, so the partitions end
    // up within a slice of each other, and every partition gets long and short rows alike (equally strong group maxima
    // for the threshold exchange).
    const uint32_t P = std::min<uint32_t>(n_partitions_hint, n_slices);
    std::vector<std::vector<uint32_t>> part_slices(P);
    {
        std::vector<std::pair<uint64_t, uint32_t>> heap;  // (chunks so far, partition): min-heap
        heap.reserve(P);
        for (uint32_t p = 0; p < P; ++p) heap.emplace_back(0, p);
        auto cmp = [](const std::pair<uint64_t, uint32_t> &a, const std::pair<uint64_t, uint32_t> &b) { return a > b; };
        std::make_heap(heap.begin(), heap.end(), cmp);
        for (uint32_t s = 0; s < n_slices; ++s) {
            std::pop_heap(heap.begin(), heap.end(), cmp);
            auto &top = heap.back();
            part_slices[top.second].push_back(s);
            top.first += slice_chunks[s];
            std::push_heap(heap.begin(), heap.end(), cmp);
        }
    }
    out.n_slices = n_slices;
    out.n_chunks = (uint32_t)n_chunks;
    out.padded_entries = n_chunks * 256;
    out.slice_rows.assign((size_t)n_slices * 64, SELL_NO_ROW);
    out.part_first.resize(P);
    out.part_count.resize(P);
    out.part_slice0.resize(P);
    // stream order: a partition's slices are contiguous; slice `so` of the stream is slice stream_slice[so] of the sorted order
    plan.stream_slice.resize(n_slices);
    plan.chunk0.resize(n_slices);
    plan.n_chunks_of.resize(n_slices);
    uint32_t chunk = 0, slice_out = 0;
    for (uint32_t p = 0; p < P; ++p) {
        out.part_first[p] = chunk;
        out.part_slice0[p] = slice_out;
        for (uint32_t s : part_slices[p]) {
            plan.stream_slice[slice_out] = s;
            plan.chunk0[slice_out] = chunk;
            plan.n_chunks_of[slice_out] = slice_chunks[s];
            for (uint32_t l = 0; l < 64; ++l) {
                const SellLane &ln = lanes[(size_t)s * 64 + l];
                if (ln.row != SELL_NO_ROW && ln.tail) out.slice_rows[(size_t)slice_out * 64 + l] = ln.row;
            }
            chunk += slice_chunks[s];
            ++slice_out;
        }
        out.part_count[p] = chunk - out.part_first[p];
    }
    return "";
}

void fill_wsell_host(const SellPlan &plan, const uint32_t *col, const float *val, SellMatrix &out) {
    const uint32_t vb = (uint32_t)out.values, PB = out.packet_bytes;
    out.packets.assign((size_t)out.n_chunks * PB, 0);
    const float neg_inf = -std::numeric_limits<float>::infinity();
    for (uint32_t so = 0; so < out.n_slices; ++so) {
        const uint32_t s = plan.stream_slice[so], nc = plan.n_chunks_of[so], chunk = plan.chunk0[so];
        for (uint32_t l = 0; l < 64; ++l) {
            const SellLane &ln = plan.lanes[(size_t)s * 64 + l];
            const bool have = ln.row != SELL_NO_ROW;
            for (uint32_t c = 0; c < nc; ++c) {
                uint8_t *pkt = out.packets.data() + (size_t)(chunk + c) * PB;
                for (uint32_t j = 0; j < 4; ++j) {
                    const uint32_t e = 4 * c + j;
                    float v;
                    uint8_t qv;  // byte values
                    uint16_t cw;
                    if (have && e < ln.n) {
                        const uint64_t src = plan.start[ln.row] + ln.first + e;
                        v = val ? val[src] : 1.0f;
                        qv = to_q1_7_rnd(v);
                        cw = (uint16_t)(col[src] << 2);
                    } else if (!have && e == 0) {
                        v = neg_inf;  // a lane without a row: its sum is -inf
                        qv = 1;       // (byte values: the PAD_ONE slot holds -inf)
                        cw = (uint16_t)(out.pad_one << 2);
                    } else {
                        v = 0.0f;  // (+0.0) * (-0.0) = -0.0: leaves every sum as it is
                        qv = 0;
                        cw = (uint16_t)(out.pad_neutral << 2);
                    }
                    if (c + 1 == nc) {  // flags of the slice's last chunk
                        if (j == 0) cw |= SELL_LAST_CHUNK;
                        if (j >= 1) cw |= (uint16_t)((ln.depth >> (2 * (j - 1))) & 3u);  // segment index, 2 bits per word
                    }
                    if (vb == 4u) std::memcpy(pkt + ((size_t)l * 4 + j) * 4, &v, 4);
                    else pkt[(size_t)l * 4 + j] = qv;
                    if (out.cw_bits == 12) colw12_store(pkt + 256u * vb, l * 4 + j, cw);
                    else std::memcpy(pkt + 256u * vb + ((size_t)l * 4 + j) * 2, &cw, 2);
                }
            }
        }
    }
}

std::string pack_wsell(uint32_t rows, uint32_t cols, uint64_t nnz, const uint32_t *row, const uint32_t *col, const float *val,
                       uint32_t n_partitions_hint, SellMatrix &out, SellValues values) {
    SellPlan plan;
    const std::string err = plan_wsell(rows, cols, nnz, row, col, n_partitions_hint, values, plan, out);
    if (!err.empty() || nnz == 0) return err;
    fill_wsell_host(plan, col, val, out);
    return "";
}

bool sell_c12_wanted() {
    const char *f = opt("SELL_C12");
    return !f || atoi(f) != 0;
}

void decode_wsell(const SellMatrix &sm, std::vector<uint32_t> &row, std::vector<uint32_t> &col, std::vector<float> &val) {
    row.clear();
    col.clear();
    val.clear();
    const uint32_t vb = (uint32_t)sm.values, PB = sm.packet_bytes, CW0 = 256u * vb;
    for (size_t p = 0; p < sm.part_first.size(); ++p) {
        uint32_t slice = sm.part_slice0[p];
        uint32_t c0 = sm.part_first[p];
        const uint32_t c_end = c0 + sm.part_count[p];
        while (c0 < c_end) {
            uint32_t nc = 1;  // chunks of this slice: up to the one flagged as last
            auto cw_at = [&](const uint8_t *pkt, uint32_t t) -> uint16_t {  // column word of entry t (lane * 4 + j) of a chunk
                if (sm.cw_bits == 12) return colw12_load(pkt + CW0, t);
                uint16_t cw;
                std::memcpy(&cw, pkt + CW0 + (size_t)t * 2, 2);
                return cw;
            };
            for (;; ++nc) {
                if (cw_at(sm.packets.data() + (size_t)(c0 + nc - 1) * PB, 0) & SELL_LAST_CHUNK) break;
            }
            // segment index of every lane (flags of the last chunk); a lane with segment index d > 0 continues the row of
            // the lane to its left, the row id sits on the row's last lane
            uint32_t depth[64], owner[64];
            const uint8_t *last = sm.packets.data() + (size_t)(c0 + nc - 1) * PB;
            for (uint32_t l = 0; l < 64; ++l) {
                const uint16_t w[4] = {cw_at(last, l * 4), cw_at(last, l * 4 + 1), cw_at(last, l * 4 + 2), cw_at(last, l * 4 + 3)};
                depth[l] = (w[1] & 3u) | ((w[2] & 3u) << 2) | ((w[3] & 3u) << 4);
            }
            for (int l = 63; l >= 0; --l) {
                const uint32_t r = sm.slice_rows[(size_t)slice * 64 + l];
                owner[l] = (r != SELL_NO_ROW) ? r : ((l < 63 && depth[l + 1] != 0) ? owner[l + 1] : SELL_NO_ROW);
            }
            for (uint32_t l = 0; l < 64; ++l) {
                const uint32_t r = owner[l];
                if (r == SELL_NO_ROW) continue;
                for (uint32_t c = 0; c < nc; ++c) {
                    const uint8_t *pkt = sm.packets.data() + (size_t)(c0 + c) * PB;
                    for (uint32_t j = 0; j < 4; ++j) {
                        const uint16_t cw = cw_at(pkt, l * 4 + j);
                        float v;
                        if (vb == 4u) std::memcpy(&v, pkt + ((size_t)l * 4 + j) * 4, 4);
                        else v = from_q1_7(pkt[(size_t)l * 4 + j]);
                        const uint32_t cc = cw >> 2;
                        if (cc >= sm.pad_neutral) continue;  // padding
                        row.push_back(r);
                        col.push_back(cc);
                        val.push_back(v);
                    }
                }
            }
            c0 += nc;
            ++slice;
        }
    }
}

}  // namespace tkspmv
