// wbscsr.cpp -- host packer / decoder for the wave block-streaming CSR layout (see wbscsr.hpp).
// Role of the reference's SpMV::packet_coo / packet_coo_partition (src/fpga/src/host_spmv_bscsr.cpp:133-248).
#include "wbscsr.hpp"
#include <cstdlib>

#include <cstdio>
#include <cstring>
#include <new>

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

// The same with a capacity per partition: partition p may take PE x (floor((p + 1) B / P) - floor(p B / P)) entries -- B packets
// dealt out over P partitions, floor(B / P) or ceil(B / P) each (balanced cuts, pack_wbscsr below); partitions beyond P (the caller
// will grow B) take ceil(B / P).
uint32_t fill_partitions_balanced(const std::vector<uint32_t> &len, uint64_t PE, uint64_t B, uint64_t P, std::vector<uint32_t> *first_rows) {
    uint32_t parts = 0;
    uint64_t s = 0, cap = 0;
    bool open = false;
    for (uint32_t r = 0; r < (uint32_t)len.size(); ++r) {
        uint64_t L = len[r] ? len[r] : 1;
        if (!open || s + L > cap) {
            const uint64_t p = parts;
            cap = PE * (p < P ? ((p + 1) * B) / P - (p * B) / P : (B + P - 1) / P);
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

uint64_t small_matrix_packets() {
    if (const char *f = opt("SMALL_PACKETS")) return (uint64_t)atoll(f);
    return SMALL_MATRIX_PACKETS;
}
uint32_t min_packets_per_partition_for(uint64_t nnz, uint32_t C, uint32_t cols) {
    if (const char *f = opt("MIN_PACKETS")) return (uint32_t)std::max(1, atoi(f));
    const uint64_t packets = nnz / (64u * (uint64_t)std::max(C, 1u));
    if (cols > 1024u || packets > small_matrix_packets()) return 4u;
    return packets <= small_matrix_packets() / 10u ? 1u : 2u;  // (up to ~2 packets per streaming wave: one each)
}

std::string pack_wbscsr(uint32_t rows, uint32_t cols, uint64_t nnz, const uint32_t *row, const uint32_t *col,
                        const float *val, Precision precision, uint32_t C, uint32_t n_partitions_hint,
                        uint32_t min_packets_per_partition, PackedMatrix &out, int &kind, uint32_t fixed_width) {
    kind = 1;
    if (C != 4 && C != 8) return "nnz_per_lane must be 4 or 8";
    if (precision == Precision::FIXED26 ? (fixed_width < 8 || fixed_width > FIXED26_MAX_WIDTH || cols > FIXED26_MAX_COLS || C != 4)
        : precision == Precision::FIXED20 ? (fixed_width < 8 || fixed_width > FIXED20_MAX_WIDTH || cols > FIXED20_MAX_COLS)
                                          : (precision == Precision::FIXED ? (fixed_width < 8 || fixed_width > 32) : fixed_width != 0))
        return "fixed_width must be in [8, 32] for fixed-point values (bit-packed: at most 20 / 26 bits, 1024 columns; 26: 4 entries per lane) and 0 otherwise";
    if (cols == 0 || cols > MAX_COLS) return "cols must be in [1, 16384]";
    if (precision == Precision::F32C12 && (cols > F32C12_MAX_COLS || C != 4)) return "12-bit column words need at most 1024 columns and 4 entries per lane";
    if (nnz > 0 && (!row || !col)) return "row/col arrays are NULL";
    if (n_partitions_hint == 0) n_partitions_hint = 1;
    if (min_packets_per_partition == 0) min_packets_per_partition = 1;

    out = PackedMatrix();
    out.rows = rows;
    out.cols = cols;
    out.nnz = nnz;
    out.precision = precision;
    out.fixed_width = fixed_width;
    out.C = C;
    out.packet_entries = WAVE * C;
    out.packet_bytes = packet_bytes_for(precision, out.packet_entries);

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
    // Balanced cuts (round 5). Partitions of m packets each come to fewer than the P asked for whenever E / (P x PE) is not close
    // below an integer -- 125k rows of 20: 3229 partitions of 3 packets for 4064 waves --, and a batch kernel's workgroups then
    // stream 6 or 7 partitions each: the launch waits for the ones with 7 (4.62 against 4.99 us per query with all at 8 of 2-3 packets).
    // Where the uniform cut misses P by more than 1/8, the packets are dealt out instead: B of them over P partitions, floor(B / P) or
    // ceil(B / P) each, B grown from the lower bound until the rows fit. (Measured: at 250k and 500k rows -- 3847 and 3824 uniform
    // partitions, 94 % of the waves -- dealing out gains nothing, 5.98 against 5.83 and 8.5-8.7 against 8.7: those keep the uniform
    // table, from which the kernels derive a wave's range without a load.)
    uint32_t used_uniform = fill_partitions(len, m * PE, nullptr);
    // (not below two packets per partition: 50k rows dealt out one packet per wave measure 3.88 against 3.66 us per query)
    const int bal_opt = opt("BALANCED_CUTS") ? atoi(opt("BALANCED_CUTS")) : 1;  // (2: already where the uniform cut misses P by 1/32 -- tuning runs)
    if (P >= 2 && total_packets_lb >= 2u * (uint64_t)P && bal_opt != 0 &&
        (bal_opt == 2 ? (uint64_t)used_uniform * 32u < (uint64_t)P * 31u : (uint64_t)used_uniform * 8u < (uint64_t)P * 7u)) {
        uint64_t B = std::max<uint64_t>(total_packets_lb, P);
        for (;;) {
            if (fill_partitions_balanced(len, PE, B, P, nullptr) <= P) break;
            B += std::max<uint64_t>(1, B / 64);
        }
        fill_partitions_balanced(len, PE, B, P, &first_rows);
        m = (B + P - 1) / P;
    } else {
        fill_partitions(len, m * PE, &first_rows);
    }
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
                if (precision == Precision::FIXED20) {  // value, column and flags in one dword; no column-word region
                    const uint32_t w = fixed20_word(to_fixed(v, fixed_width), (uint32_t)(cw >> COLW_COL_SHIFT), cw & 3u);
                    std::memcpy(pkt + (size_t)slot * 4, &w, 4);
                    continue;
                }
                if (precision == Precision::F32C12) {
                    std::memcpy(pkt + (size_t)slot * 4, &v, 4);
                    colw12s_store(pkt + (size_t)PE * 4, slot, cw);
                    continue;
                }
                if (precision == Precision::FIXED26) {  // 5 bytes per entry: a dword of the 16-byte plane + 6 bits of the lane's E
                    const uint32_t colv = (uint32_t)(cw >> COLW_COL_SHIFT);
                    const uint32_t w = fixed26_d(to_fixed(v, fixed_width), colv, cw & 3u);
                    std::memcpy(pkt + (size_t)slot * 4, &w, 4);
                    uint32_t e;
                    std::memcpy(&e, pkt + (size_t)PE * 4 + (size_t)(slot >> 2) * 4, 4);
                    e |= fixed26_e(slot & 3u, colv);
                    std::memcpy(pkt + (size_t)PE * 4 + (size_t)(slot >> 2) * 4, &e, 4);
                    continue;
                }
                if (precision == Precision::F32) {
                    std::memcpy(pkt + (size_t)slot * 4, &v, 4);
                } else if (precision == Precision::F16) {
                    const uint16_t hv = to_half(v);
                    std::memcpy(pkt + (size_t)slot * 2, &hv, 2);
                } else if (precision == Precision::FIXED) {
                    const uint32_t q = to_fixed(v, fixed_width);
                    std::memcpy(pkt + (size_t)slot * 4, &q, 4);
                } else if (precision == Precision::Q1_7_RND) {
                    pkt[slot] = to_q1_7_rnd(v);
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
                float v;
                if (pm.precision == Precision::FIXED20) {
                    uint32_t w;
                    std::memcpy(&w, pkt + (size_t)s * 4, 4);
                    cw = (uint16_t)(w & 0xFFFu);
                    v = from_fixed(w & 0xFFFFF000u);
                } else if (pm.precision == Precision::FIXED26) {
                    uint32_t w, e;
                    std::memcpy(&w, pkt + (size_t)s * 4, 4);
                    std::memcpy(&e, pkt + (size_t)PE * 4 + (size_t)(s >> 2) * 4, 4);
                    const uint32_t colv = ((w >> 2) & 15u) | (((e >> (6u * (s & 3u))) & 63u) << 4);
                    cw = (uint16_t)((colv << COLW_COL_SHIFT) | (w & 3u));
                    v = from_fixed(w & 0xFFFFFFC0u);
                } else if (pm.precision == Precision::F32C12) {
                    cw = colw12s_load(pkt + (size_t)PE * 4, s);
                } else {
                    std::memcpy(&cw, pkt + (size_t)PE * vb + (size_t)s * 2, 2);
                }
                if (pm.precision == Precision::FIXED20 || pm.precision == Precision::FIXED26) {
                } else if (pm.precision == Precision::F32 || pm.precision == Precision::F32C12) {
                    std::memcpy(&v, pkt + (size_t)s * 4, 4);
                } else if (pm.precision == Precision::F16) {
                    uint16_t hv;
                    std::memcpy(&hv, pkt + (size_t)s * 2, 2);
                    v = from_half(hv);
                } else if (pm.precision == Precision::FIXED) {
                    uint32_t q;
                    std::memcpy(&q, pkt + (size_t)s * 4, 4);
                    v = from_fixed(q);
                } else {
                    v = from_q1_7(pkt[s]);
                }
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

// ---- binary cache -----------------------------------------------------------------------------------------------------
namespace {
struct FileHeader {  // 128 bytes, little endian (the only byte order this code is built for)
    char magic[8];   // "TKSPMV1\0"
    uint32_t version, precision, C, packet_entries, packet_bytes, n_packets, packets_per_partition, n_parts;
    uint32_t rows, cols;
    uint64_t nnz, packed_entries, placeholders;
    uint64_t payload_bytes, checksum;  // FNV-1a 64 over the payload
    uint32_t fixed_width;              // Precision::FIXED only (0 otherwise; files written before it existed read as 0)
    uint8_t pad[128 - 8 - 11 * 4 - 5 * 8];
};
static_assert(sizeof(FileHeader) == 128, "header layout");
const char MAGIC[8] = {'T', 'K', 'S', 'P', 'M', 'V', '1', '\0'};

uint64_t fnv1a(uint64_t h, const void *p, size_t n) {
    const unsigned char *b = static_cast<const unsigned char *>(p);
    for (size_t i = 0; i < n; ++i) {
        h ^= b[i];
        h *= 1099511628211ull;
    }
    return h;
}
// The checksum walks the payload in 8-byte words where it can: a 0.7 GB stream is hashed in well under a second.
uint64_t fnv1a_words(uint64_t h, const void *p, size_t n) {
    const unsigned char *b = static_cast<const unsigned char *>(p);
    size_t i = 0;
    for (; i + 8 <= n; i += 8) {
        uint64_t w;
        std::memcpy(&w, b + i, 8);
        h ^= w;
        h *= 1099511628211ull;
    }
    return fnv1a(h, b + i, n - i);
}
}  // namespace

std::string save_packed(const PackedMatrix &pm, const char *path) {
    if (!path) return "path is NULL";
    const size_t n_parts = pm.part_first.size();
    if (pm.part_count.size() != n_parts || pm.part_row0.size() != n_parts || pm.part_rows.size() != n_parts ||
        pm.pkt_row.size() != pm.n_packets || pm.packets.size() != pm.stream_bytes())
        return "packed matrix is incomplete (was it already uploaded and dropped?)";
    FileHeader hd;
    std::memset(&hd, 0, sizeof(hd));
    std::memcpy(hd.magic, MAGIC, 8);
    hd.version = 2;  // (2: F32C12 streams carry the split 12-bit plane)
    hd.precision = (uint32_t)pm.precision;
    hd.fixed_width = pm.fixed_width;
    hd.C = pm.C;
    hd.packet_entries = pm.packet_entries;
    hd.packet_bytes = pm.packet_bytes;
    hd.n_packets = pm.n_packets;
    hd.packets_per_partition = pm.packets_per_partition;
    hd.n_parts = (uint32_t)n_parts;
    hd.rows = pm.rows;
    hd.cols = pm.cols;
    hd.nnz = pm.nnz;
    hd.packed_entries = pm.packed_entries;
    hd.placeholders = pm.placeholders;
    const struct {
        const void *p;
        size_t n;
    } parts[] = {{pm.packets.data(), pm.packets.size()},          {pm.pkt_row.data(), pm.pkt_row.size() * 4},
                 {pm.part_first.data(), n_parts * 4},              {pm.part_count.data(), n_parts * 4},
                 {pm.part_row0.data(), n_parts * 4},               {pm.part_rows.data(), n_parts * 4}};
    uint64_t sum = 1469598103934665603ull;
    for (const auto &q : parts) {
        hd.payload_bytes += q.n;
        sum = fnv1a_words(sum, q.p, q.n);
    }
    hd.checksum = sum;
    FILE *f = std::fopen(path, "wb");
    if (!f) return std::string("cannot open ") + path + " for writing";
    bool ok = std::fwrite(&hd, sizeof(hd), 1, f) == 1;
    for (const auto &q : parts) ok = ok && (q.n == 0 || std::fwrite(q.p, 1, q.n, f) == q.n);
    ok = (std::fclose(f) == 0) && ok;
    if (!ok) return std::string("short write to ") + path;
    return "";
}

std::string load_packed(const char *path, PackedMatrix &pm) {
    if (!path) return "path is NULL";
    FILE *f = std::fopen(path, "rb");
    if (!f) return std::string("cannot open ") + path;
    FileHeader hd;
    auto fail = [&](const std::string &m) {
        std::fclose(f);
        return m;
    };
    if (std::fread(&hd, sizeof(hd), 1, f) != 1) return fail("file too short for a header");
    if (std::memcmp(hd.magic, MAGIC, 8) != 0) return fail("not a .tkspmv file (bad magic)");
    if (hd.version != 2) return fail("unsupported .tkspmv version");
    if ((hd.precision != (uint32_t)Precision::F32 && hd.precision != (uint32_t)Precision::Q1_7 &&
         hd.precision != (uint32_t)Precision::F16 && hd.precision != (uint32_t)Precision::FIXED &&
         hd.precision != (uint32_t)Precision::Q1_7_RND && hd.precision != (uint32_t)Precision::FIXED20 &&
         hd.precision != (uint32_t)Precision::F32C12 && hd.precision != (uint32_t)Precision::FIXED26) ||
        (hd.precision == (uint32_t)Precision::F32C12 && (hd.cols > F32C12_MAX_COLS || hd.C != 4)) ||
        (hd.precision == (uint32_t)Precision::FIXED26
             ? (hd.fixed_width < 8 || hd.fixed_width > FIXED26_MAX_WIDTH || hd.cols > FIXED26_MAX_COLS || hd.C != 4)
         : hd.precision == (uint32_t)Precision::FIXED20
             ? (hd.fixed_width < 8 || hd.fixed_width > FIXED20_MAX_WIDTH || hd.cols > FIXED20_MAX_COLS)
             : (hd.precision == (uint32_t)Precision::FIXED ? (hd.fixed_width < 8 || hd.fixed_width > 32) : hd.fixed_width != 0)) ||
        (hd.C != 4 && hd.C != 8) ||
        hd.packet_entries != 64 * hd.C ||
        hd.packet_bytes != packet_bytes_for((Precision)hd.precision, hd.packet_entries) ||
        hd.packed_entries != (uint64_t)hd.n_packets * hd.packet_entries)
        return fail("inconsistent header");
    const uint64_t expect = (uint64_t)hd.n_packets * hd.packet_bytes + (uint64_t)hd.n_packets * 4 + (uint64_t)hd.n_parts * 16;
    if (hd.payload_bytes != expect) return fail("payload size does not match the header");
    PackedMatrix out;
    out.rows = hd.rows;
    out.cols = hd.cols;
    out.nnz = hd.nnz;
    out.precision = (Precision)hd.precision;
    out.fixed_width = hd.fixed_width;
    out.C = hd.C;
    out.packet_entries = hd.packet_entries;
    out.packet_bytes = hd.packet_bytes;
    out.n_packets = hd.n_packets;
    out.packets_per_partition = hd.packets_per_partition;
    out.packed_entries = hd.packed_entries;
    out.placeholders = hd.placeholders;
    try {
        out.packets.resize((size_t)hd.n_packets * hd.packet_bytes);
        out.pkt_row.resize(hd.n_packets);
        out.part_first.resize(hd.n_parts);
        out.part_count.resize(hd.n_parts);
        out.part_row0.resize(hd.n_parts);
        out.part_rows.resize(hd.n_parts);
    } catch (const std::bad_alloc &) {
        return fail("out of host memory");
    }
    struct {
        void *p;
        size_t n;
    } parts[] = {{out.packets.data(), out.packets.size()},      {out.pkt_row.data(), out.pkt_row.size() * 4},
                 {out.part_first.data(), out.part_first.size() * 4}, {out.part_count.data(), out.part_count.size() * 4},
                 {out.part_row0.data(), out.part_row0.size() * 4},   {out.part_rows.data(), out.part_rows.size() * 4}};
    uint64_t sum = 1469598103934665603ull;
    for (auto &q : parts) {
        if (q.n != 0 && std::fread(q.p, 1, q.n, f) != q.n) return fail("file is truncated");
        sum = fnv1a_words(sum, q.p, q.n);
    }
    if (std::fgetc(f) != EOF) return fail("trailing bytes after the payload");
    std::fclose(f);
    if (sum != hd.checksum) return "checksum mismatch (corrupted file)";
    // structural checks the kernels rely on: packet ranges of the partitions are in bounds, row ids below `rows`
    for (uint32_t p = 0; p < hd.n_parts; ++p) {
        if ((uint64_t)out.part_first[p] + out.part_count[p] > hd.n_packets) return "partition table out of range";
    }
    for (uint32_t p = 0; p < hd.n_packets; ++p) {
        if (out.pkt_row[p] >= hd.rows) return "packet row table out of range";
    }
    // column ids: the kernels read x[col] from LDS without a bounds check
    if (hd.cols == 0 || hd.cols > MAX_COLS) return "column count out of range";
    {
        const size_t vbytes = (size_t)hd.packet_entries * value_bytes((Precision)hd.precision);
        for (uint32_t p = 0; p < hd.n_packets; ++p) {
            const uint8_t *cwp = out.packets.data() + (size_t)p * hd.packet_bytes + vbytes;
            for (uint32_t s = 0; s < hd.packet_entries; ++s) {
                uint16_t cw;
                if (hd.precision == (uint32_t)Precision::FIXED20) {
                    uint32_t w;
                    std::memcpy(&w, out.packets.data() + (size_t)p * hd.packet_bytes + (size_t)s * 4, 4);
                    cw = (uint16_t)(w & 0xFFFu);
                } else if (hd.precision == (uint32_t)Precision::FIXED26) {
                    uint32_t w, e;
                    const uint8_t *pk = out.packets.data() + (size_t)p * hd.packet_bytes;
                    std::memcpy(&w, pk + (size_t)s * 4, 4);
                    std::memcpy(&e, pk + (size_t)hd.packet_entries * 4 + (size_t)(s >> 2) * 4, 4);
                    cw = (uint16_t)(((((w >> 2) & 15u) | (((e >> (6u * (s & 3u))) & 63u) << 4)) << COLW_COL_SHIFT) | (w & 3u));
                } else if (hd.precision == (uint32_t)Precision::F32C12) {
                    cw = colw12s_load(cwp, s);
                } else {
                    std::memcpy(&cw, cwp + (size_t)s * 2, 2);
                }
                if ((uint32_t)(cw >> COLW_COL_SHIFT) >= hd.cols) return "column id out of range in the packet stream";
            }
        }
    }
    pm = std::move(out);
    return "";
}

}  // namespace tkspmv
