// device_pack.hpp -- the wave-BSCSR packer on the GPU (see device_pack.hip). Same output as pack_wbscsr(), byte for byte.
#pragma once
#include <cstdint>
#include <string>

#include "wbscsr.hpp"
#include "wsell.hpp"

namespace tkspmv {

struct DevicePacked {
    PackedMatrix meta;             // everything but `packets` / `pkt_row`, which stay on the device (see download_device_packed)
    uint8_t *d_packets = nullptr;  // [n_packets * packet_bytes] device memory, owned by the caller after a successful pack
    uint32_t *d_pkt_row = nullptr; // [n_packets]
    bool keep_coo = false;         // in: leave the column and value arrays of the COO in HBM (d_col, d_val) for pack_wsell_device
    uint32_t *d_col = nullptr;     // [nnz] only with keep_coo; freed by free_device_packed
    float *d_val = nullptr;        // [nnz] only with keep_coo and values given
    double upload_ms = 0.0;        // host-to-device copy of the COO
    double kernels_ms = 0.0;       // everything after it (lengths, scans, cuts, scatter; small copies of the side tables)
};

// Packs a row-sorted COO given as HOST arrays on the current HIP device. Same validation, same error messages and the
// same `kind` (0 ok, 1 invalid, 2 not sorted) as pack_wbscsr.
std::string pack_wbscsr_device(uint32_t rows, uint32_t cols, uint64_t nnz, const uint32_t *row, const uint32_t *col,
                               const float *val, Precision precision, uint32_t C, uint32_t n_partitions_hint,
                               uint32_t min_packets_per_partition, uint32_t fixed_width, DevicePacked &out, int &kind);
// Copies the stream and the packet row table back into meta.packets / meta.pkt_row (tests, tkspmv_pack_device).
std::string download_device_packed(DevicePacked &dp);
void free_device_packed(DevicePacked &dp);

// The wave-sliced ELL layout of the multi-query kernel: the plan on the host (plan_wsell), the fill on the GPU.
struct DeviceSell {
    SellMatrix meta;               // everything but `packets` (see download_device_sell)
    uint8_t *d_packets = nullptr;  // [n_chunks * packet_bytes] device memory, owned by the caller after a successful pack
    double plan_ms = 0.0, upload_ms = 0.0, kernels_ms = 0.0;
};
// row/col/val: the COO as HOST arrays (the plan reads row and col); d_col/d_val: the same column and value arrays in
// HBM if the caller has them there already (DevicePacked::keep_coo), else NULL and they are uploaded here.
std::string pack_wsell_device(uint32_t rows, uint32_t cols, uint64_t nnz, const uint32_t *row, const uint32_t *col,
                              const float *val, uint32_t n_partitions_hint, SellValues values, const uint32_t *d_col,
                              const float *d_val, DeviceSell &out);
std::string download_device_sell(DeviceSell &ds);
void free_device_sell(DeviceSell &ds);

}  // namespace tkspmv
