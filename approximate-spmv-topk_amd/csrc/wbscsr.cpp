// wbscsr.cpp -- host packer / decoder for the wave block-streaming CSR layout (see wbscsr.hpp).
// Role of the reference's SpMV::packet_coo / packet_coo_partition (src/fpga/src/host_spmv_bscsr.cpp:133-248).
#include "wbscsr.hpp"

#include <algorithm>
#include <cstring>

namespace tkspmv {

namespace {

// Greedy fill: walk rows in order, open a new partition when the next row does not fit in `cap` entries.
// Returns the number of partitions; optionally records the first row of each.
uint32_t fill_partitions(const std::vector<uint32_t> &len, uint64_t cap, std::vector<uint32_t> *first_rows) {
    uint32_t parts = 0;
    uint64_t s = 0;
    bool open = false;
    for (uint32_t r = 0; r < (uint32_t)len.size(); ++r) {
        uint64_t L = len[r] ? len[r] : 1;  // empty row -> one placeholder entry
        if (!open || s + L > cap) {
            ++parts;
            if (first_rows) first_rows->push_back(r);
            s = 0;
            open = true;
        }
        s += L;
    }
    return parts;
}

}  // namespace

std::string pack_wbscsr(uint32_t rows, uint32_t cols, uint64_t nnz, const uint32_t *row, const uint32_t *col,
                        const float *val, Precision precision, uint32_t C, uint32_t n_partitions_hint,
                        uint32_t min_packets_per_partition, PackedMatrix &out, int &kind) {
    kind = 1;
    if (C != 4 && C != 8) return "nnz_per_lane must be 4 or 8";
    if (cols == 0 || cols > MAX_COLS) return "cols must be in [1, 16384]";
    if (nnz > 0 && (!row || !col)) return "row/col arrays are NULL";
    if (n_partitions_hint == 0) n_partitions_hint = 1;
    if (min_packets_per_partition == 0) min_packets_per_partition = 1;

    out = PackedMatrix();
    out.rows = rows;
    out.cols = cols;
    out.nnz = nnz;
    out.precision = precision;
    out.C = C;
    out.packet_entries = WAVE * C;
    out.packet_bytes = out.packet_entries * (value_bytes(precision) + 2);

    // Row lengths over [0, last_row]; validates ordering and ranges.
    for (uint64_t i = 1; i < nnz; ++i) {
        if (row[i] < row[i - 1]) {
            kind = 2;
            return "COO rows are not sorted in non-decreasing order";
        }
    }
    uint32_t last_row = 0;
    if (nnz > 0) {
        last_row = row[nnz - 1];
        if (last_row >= rows) return "row id out of range (>= rows)";
    }
    std::vector<uint32_t> len(nnz ? (size_t)last_row + 1 : 0, 0u);
    for (uint64_t i = 0; i < nnz; ++i) {
        if (col[i] >= cols) return "column id out of range (>= cols)";
        ++len[row[i]];
    }
    uint64_t placeholders = 0;
    for (uint32_t L : len) placeholders += (L == 0);
    out.placeholders = placeholders;
    const uint64_t E = nnz + placeholders;
    const uint32_t PE = out.packet_entries;

    if (E == 0) {  // empty matrix: no packets, no partitions
        kind = 0;
        return "";
    }

    // Partition count: never more partitions than packets / min_packets.
    uint64_t total_packets_lb = (E + PE - 1) / PE;
    uint64_t max_parts = std::max<uint64_t>(1, total_packets_lb / min_packets_per_partition);
    uint32_t P = (uint32_t)std::min<uint64_t>(n_partitions_hint, max_parts);
    uint64_t m = std::max<uint64_t>(1, (E + (uint64_t)P * PE - 1) / ((uint64_t)P * PE));
    std::vector<uint32_t> first_rows;
    for (;;) {
        uint32_t used = fill_partitions(len, m * PE, nullptr);
        if (used <= P) break;
        ++m;  // padding pushed us over the wave count: allow one more packet per partition
    }
    fill_partitions(len, m * PE, &first_rows);
    const uint32_t n_parts = (uint32_t)first_rows.size();
    out.packets_per_partition = (uint32_t)m;

    // Packet counts per partition.
    std::vector<uint64_t> row_start(len.size() + 1, 0);  // entry offset of each row in the placeholder-expanded stream
    for (size_t r = 0; r < len.size(); ++r) row_start[r + 1] = row_start[r] + (len[r] ? len[r] : 1);
    out.part_first.resize(n_parts);
    out.part_count.resize(n_parts);
    out.part_row0.resize(n_parts);
    out.part_rows.resize(n_parts);
    uint64_t n_packets = 0;
    for (uint32_t p = 0; p < n_parts; ++p) {
        uint32_t r0 = first_rows[p];
        uint32_t r1 = (p + 1 < n_parts) ? first_rows[p + 1] : (uint32_t)len.size();
        uint64_t entries = row_start[r1] - row_start[r0];
        uint64_t pk = (entries + PE - 1) / PE;
        out.part_first[p] = (uint32_t)n_packets;
        out.part_count[p] = (uint32_t)pk;
        out.part_row0[p] = r0;
        out.part_rows[p] = r1 - r0;
        n_packets += pk;
    }
    if (n_packets > 0xFFFFFFFFull) return "matrix too large (packet count overflows 32 bits)";
    out.n_packets = (uint32_t)n_packets;
    out.packed_entries = n_packets * PE;
    out.packets.assign((size_t)n_packets * out.packet_bytes, 0);
    out.pkt_row.assign((size_t)n_packets, 0);

    const uint32_t vb = value_bytes(precision);
    // Fill packets partition by partition.
    uint64_t src = 0;  // index into the COO
    for (uint32_t p = 0; p < n_parts; ++p) {
        uint32_t r0 = out.part_row0[p], r1 = r0 + out.part_rows[p];
        uint64_t e = 0;  // entry index inside the partition
        uint8_t *base = out.packets.data() + (size_t)out.part_first[p] * out.packet_bytes;
        uint32_t *prow = out.pkt_row.data() + out.part_first[p];
        for (uint32_t r = r0; r < r1; ++r) {
            uint32_t L = len[r];
            uint32_t n_entries = L ? L : 1;
            for (uint32_t j = 0; j < n_entries; ++j, ++e) {
                uint64_t pk = e / PE;
                uint32_t stream_slot = (uint32_t)(e % PE);
                uint32_t slot = slot_to_index(stream_slot, C);
                uint8_t *pkt = base + pk * out.packet_bytes;
                uint16_t cw;
                float v;
                if (L == 0) {
                    cw = COLW_SKIP;
                    v = 0.0f;
                } else {
                    cw = (uint16_t)(col[src] << COLW_COL_SHIFT);
                    v = val ? val[src] : 1.0f;
                    ++src;
                }
                if (j + 1 == n_entries) cw |= COLW_ROW_END;
                // Rows are contiguous, so the first row that ENDS in a packet is the row of its first entry
                // (if that row does not end here, no row does and the value is unused).
                if (stream_slot == 0) prow[pk] = r;
                if (precision == Precision::F32) {
                    std::memcpy(pkt + (size_t)slot * 4, &v, 4);
                } else {
                    pkt[slot] = to_q1_7(v);
                }
                std::memcpy(pkt + (size_t)PE * vb + (size_t)slot * 2, &cw, 2);
            }
        }
    }
    kind = 0;
    return "";
}

void decode_wbscsr(const PackedMatrix &pm, std::vector<uint32_t> &row, std::vector<uint32_t> &col,
                   std::vector<float> &val) {
    row.clear();
    col.clear();
    val.clear();
    const uint32_t PE = pm.packet_entries;
    const uint32_t vb = value_bytes(pm.precision);
    for (size_t p = 0; p < pm.part_first.size(); ++p) {
        uint32_t r = pm.part_row0[p];
        uint32_t rows_left = pm.part_rows[p];
        for (uint32_t k = 0; k < pm.part_count[p] && rows_left; ++k) {
            const uint8_t *pkt = pm.packets.data() + (size_t)(pm.part_first[p] + k) * pm.packet_bytes;
            for (uint32_t ss = 0; ss < PE && rows_left; ++ss) {
                const uint32_t s = slot_to_index(ss, pm.C);
                uint16_t cw;
                std::memcpy(&cw, pkt + (size_t)PE * vb + (size_t)s * 2, 2);
                float v;
                if (pm.precision == Precision::F32)
                    std::memcpy(&v, pkt + (size_t)s * 4, 4);
                else
                    v = from_q1_7(pkt[s]);
                if (!(cw & COLW_SKIP)) {
                    row.push_back(r);
                    col.push_back((uint32_t)(cw >> COLW_COL_SHIFT));
                    val.push_back(v);
                }
                if (cw & COLW_ROW_END) {
                    ++r;
                    --rows_left;
                }
            }
        }
    }
}

}  // namespace tkspmv
