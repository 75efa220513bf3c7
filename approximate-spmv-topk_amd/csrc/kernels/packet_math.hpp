// kernels/packet_math.hpp -- Per-packet arithmetic of the wave-BSCSR kernels: DPP helpers, threshold exchange helpers, products, in-lane segmented sums, the clipped cross-lane scan, the candidate path.
// Part of engine.hip (one translation unit: included there in this order; device code only).
#pragma once
#include "select.hpp"

namespace tkspmv {

// DPP lane movement (gfx950 keeps the GFX9 controls): lanes without a valid source receive 0.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_zero(float src) {
    return __builtin_bit_cast(
        float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, src), CTRL, ROW_MASK, 0xF, true));
}
constexpr int DPP_ROW_SHR1 = 0x111, DPP_ROW_SHR2 = 0x112, DPP_ROW_SHR4 = 0x114, DPP_ROW_SHR8 = 0x118;
constexpr int DPP_WAVE_SHL1 = 0x130, DPP_WAVE_SHR1 = 0x138, DPP_ROW_BCAST15 = 0x142, DPP_ROW_BCAST31 = 0x143;

__device__ __forceinline__ float max2(float a, float b) {  // one v_max_f32 (fmaxf puts a canonicalising v_max in front of each operand)
    float r;
    asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
// Wave-wide maximum with DPP (result uniform, returned through an SGPR).
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_keep(float v) {  // lanes without a valid source keep their own value
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, v), __builtin_bit_cast(int, v),
                                                                 CTRL, ROW_MASK, 0xF, false));
}
__device__ __forceinline__ float wave_max(float v) {
    v = fmaxf(v, v);  // (one canonicalisation; the steps below are plain v_max)
    v = max2(v, dpp_keep<DPP_ROW_SHR1, 0xF>(v));
    v = max2(v, dpp_keep<DPP_ROW_SHR2, 0xF>(v));
    v = max2(v, dpp_keep<DPP_ROW_SHR4, 0xF>(v));
    v = max2(v, dpp_keep<DPP_ROW_SHR8, 0xF>(v));
    v = max2(v, dpp_keep<DPP_ROW_BCAST15, 0xA>(v));
    v = max2(v, dpp_keep<DPP_ROW_BCAST31, 0xC>(v));
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}
__device__ __forceinline__ uint32_t wave_min_u32(uint32_t v) {
    auto mv = [](uint32_t a, uint32_t b) { return b < a ? b : a; };
    v = mv(v, (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, DPP_ROW_SHR1, 0xF, 0xF, false));
    v = mv(v, (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, DPP_ROW_SHR2, 0xF, 0xF, false));
    v = mv(v, (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, DPP_ROW_SHR4, 0xF, 0xF, false));
    v = mv(v, (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, DPP_ROW_SHR8, 0xF, 0xF, false));
    v = mv(v, (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, DPP_ROW_BCAST15, 0xA, 0xF, false));
    v = mv(v, (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, DPP_ROW_BCAST31, 0xC, 0xF, false));
    return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}

// Threshold exchange, reader side. One wave: (1) issue the loads of the published maxima early, (2) much later
// stage them in LDS and reduce: tau = min over sets of (max over the set's groups).
struct TauRegs {
    uint32_t k[MAX_GM];
};
__device__ __forceinline__ void tau_issue(const StreamParams &P, uint32_t lane, TauRegs &t) {
    // gmax is allocated with MAX_GM * 64 entries (zero beyond n_groups_pub), so no bounds predicate is needed;
    // whole 256-B rows beyond the used part are skipped with a uniform branch.
#pragma unroll
    for (int i = 0; i < MAX_GM; ++i) {
        t.k[i] = 0u;
        if (64u * i < P.n_groups_pub)
            t.k[i] = __hip_atomic_load(&P.gmax[lane + 64u * i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}
// tau = (lower bound within 2^-8 relative of) the k-th largest published maximum: the maxima are scores of distinct
// rows, so k of them at or above tau prove that the k-th best score overall is at least tau.
__device__ __forceinline__ float tau_from_maxima(const StreamParams &P, const TauRegs &t, const float min_units) {
    const uint32_t rows_used = (P.n_groups_pub + 63u) >> 6;
    uint32_t key;
    if (rows_used <= 1) key = kth_largest_prefix<1, 17>(t.k, P.k);
    else if (rows_used <= 2) key = kth_largest_prefix<2, 17>(t.k, P.k);
    else if (rows_used <= 4) key = kth_largest_prefix<4, 17>(t.k, P.k);
    else if (rows_used <= 8) key = kth_largest_prefix<8, 17>(t.k, P.k);
    else key = kth_largest_prefix<16, 17>(t.k, P.k);
    float tau = min_units;
    if (key != 0u) {
        const float f = key_to_float(key);
        tau = f > tau ? f : tau;
    }
    return tau;
}

// Writer side: lanes 0..gpw-1 of the calling wave push the workgroup's group maxima (kept in LDS) to gmax.
__device__ __forceinline__ void publish_group_max(const StreamParams &P, uint32_t bid, uint32_t lane, uint32_t *misc) {
    if (lane < P.gpw) {
        const uint32_t g = bid * P.gpw + lane;
        const uint32_t key = misc[MISC_GRPMAX + lane];
        if (g < P.n_groups_pub && key > misc[MISC_PUBLISHED + lane]) {
            misc[MISC_PUBLISHED + lane] = key;
            // single writer per slot (this workgroup): a write-through store, no memory-side read-modify-write
            __hip_atomic_store(&P.gmax[g], key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

// ------------------------------------------------------------------------------------------------------------
// The fused streaming kernel
// ------------------------------------------------------------------------------------------------------------
// ---- single-instruction helpers ---------------------------------------------------------------------------------
// hipcc turns mask arithmetic back into v_cmp + v_cndmask (+ s_nop hazards); these keep it at one VALU op each.
// All are plain VGPR -> VGPR VALU operations (no hazard besides the DPP one noted at `tail` below).
template <int BIT>
__device__ __forceinline__ uint32_t bit_mask(uint32_t w) {  // all ones iff bit BIT of w is set
    uint32_t r;
    asm("v_bfe_i32 %0, %1, %2, 1" : "=v"(r) : "v"(w), "n"(BIT));
    return r;
}
__device__ __forceinline__ float mask_select(uint32_t m, float if_set, float if_clear) {  // bitwise m ? a : b
    float r;
    asm("v_bfi_b32 %0, %1, %2, %3" : "=v"(r) : "v"(m), "v"(if_set), "v"(if_clear));
    return r;
}
__device__ __forceinline__ float mask_clear(uint32_t m, float a) {  // a where m is clear, +0.0 where set
    float r;
    asm("v_bfi_b32 %0, %1, 0, %2" : "=v"(r) : "v"(m), "v"(a));
    return r;
}
__device__ __forceinline__ float max3(float a, float b, float c) {
    float r;
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}

// LDS reads at an integer byte address (address space 3): the gathers of x[col] compute their addresses with integer
// instructions (field extraction fused with the base of the x copy in use), so the pointer is an integer.
typedef __attribute__((address_space(3))) const float lds_cfloat;
typedef __attribute__((address_space(3))) const uint32_t lds_cu32;
__device__ __forceinline__ float lds_f32(uint32_t byte_addr) { return *reinterpret_cast<lds_cfloat *>((uintptr_t)byte_addr); }
__device__ __forceinline__ uint32_t lds_u32(uint32_t byte_addr) { return *reinterpret_cast<lds_cu32 *>((uintptr_t)byte_addr); }
__device__ __forceinline__ uint32_t lds_addr_of(const void *p) {  // LDS byte address of a pointer into a __shared__ object
    return (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const void *)p;
}
// (w & mask) | base in one instruction; mask lives in a VGPR (VOP3 takes no literal on gfx9 and only one scalar operand)
__device__ __forceinline__ uint32_t and_or(uint32_t w, uint32_t mask_vgpr, uint32_t base_sgpr) {
    uint32_t r;
    asm("v_and_or_b32 %0, %1, %2, %3" : "=v"(r) : "v"(w), "v"(mask_vgpr), "s"(base_sgpr));
    return r;
}

// Finished-row sums of one packet as seen by one lane, in full: what the candidate path, the SpMV-only variant and the
// per-partition lists need. fl: bit j = ROW_END of the lane's entry j, bit 8 + j = its SKIP flag (placeholder of an empty row).
template <int C>
struct RowSums {
    float rs[C];
    uint32_t fl;
    float best_any;  // max over the lane's row ends, placeholders included
    __device__ __forceinline__ bool end(int j) const { return (fl >> j) & 1u; }
    __device__ __forceinline__ bool valid(int j) const { return ((fl >> j) & 0x101u) == 1u; }
};

// What the reduction leaves behind on the hot path: nothing is selected or compared per entry until some lane's trigger
// reaches the threshold (expand() then produces the RowSums).
//   s[j] : in-lane running sum of entry j's row segment (restarted behind every row end inside the lane; lane 0's entry 0
//          includes the carry of the previous packet)
//   S    : the row sum at the lane's FIRST row end = inclusive scan value of the lane below + the lane's head
//   m[j] : all ones where entry j ends a row
// Every finished row of the lane has its sum in {S, s[1], ..., s[C-1]} (the first row end yields S, later ones their own
// s[j]); the other members of that set are partial sums of rows still running.
template <int C>
struct Reduced {
    float s[C];
    float S;
    uint32_t m[C];
};

// Products are in p; in-lane segmented sums, cross-lane segmented scan; updates the packet carry.
// Arithmetic (mirrored statement for statement by oracle_packed_scores in oracle/oracle.c):
//   p_0 += carry on lane 0;  s_0 = p_0,  s_j = (end_{j-1} ? +0 : s_{j-1}) + p_j
//   tail = end_{C-1} ? +0 : s_{C-1};   head = s at the lane's first row end
//   vv = clipped Kogge-Stone scan of tail over the 64 lanes (never across a lane that holds a row end)
//   row sum at the lane's first row end = vv[lane-1] + head, at its later row ends = s_j; carry' = vv[63]
// INT: the C "floats" (and the carry) hold u32 fixed-point words; every sum is an integer add (wrapping at 2^32 = 2.0 in
// Q1.31: the reference's real_type sums wrap the same way), lane movement and masking are bitwise either way.
template <bool INT>
__device__ __forceinline__ float add_rn(float a, float b) {
    if (INT) return __uint_as_float(__float_as_uint(a) + __float_as_uint(b));
    return __fadd_rn(a, b);
}
template <int C, bool INT = false>
__device__ __forceinline__ Reduced<C> reduce_core(float (&p)[C], const uint32_t (&m)[C], const bool has_end, float &carry) {
    Reduced<C> R;
    p[0] = __builtin_amdgcn_inverse_ballot_w64(1ull) ? add_rn<INT>(p[0], carry) : p[0];  // lane 0 only
    R.s[0] = p[0];
#pragma unroll
    for (int j = 1; j < C; ++j) R.s[j] = add_rn<INT>(mask_clear(m[j - 1], R.s[j - 1]), p[j]);
    float head = R.s[C - 1];
#pragma unroll
    for (int j = C - 2; j >= 0; --j) head = mask_select(m[j], R.s[j], head);
    float tail;
    // (the DPP instruction that reads `tail` next needs two wait states after a VALU write; the compiler does
    //  not look inside asm, hence the explicit s_nop)
    asm("v_bfi_b32 %0, %1, 0, %2\n\ts_nop 1" : "=v"(tail) : "v"(m[C - 1]), "v"(R.s[C - 1]));

    // Lane masks of the clipped scan, computed once on the scalar unit from H = lanes holding a row end:
    //   M_d  : no row end in lanes (l-d, l]                     (steps row_shr:1,2,4,8)
    //   P16  : no row end in [first lane of l's 16-lane row, l]  (step row_bcast:15; only rows 1 and 3 have a source)
    //   P32  : no row end in [lane 32, l]                        (step row_bcast:31; only lanes 32..63 have a source)
    // Round 5: 15 scalar instructions where round 4's doubling chain with its clipping constants took 25.
    // * M_d is the plain doubling chain M_2d = M_d & (M_d << d). It runs across the 16-lane rows, which is immaterial: the lanes it
    //   concerns (lane % 16 < d) receive 0 from the DPP shift whatever their mask says.
    // * P16 and P32 are "trailing ones" of a field of M1 = ~H: f & ~(f + 1) keeps exactly the ones below the field's first zero --
    //   the lanes from the row's (half's) first lane up to the first row end. One add serves both 16-bit fields (rows 1 and 3; the
    //   fields between them are cleared first, so a carry out of row 1 stops in row 2's empty field); P32 is the same on the upper
    //   32-bit word.
    const uint64_t H = __ballot(has_end);
    const uint64_t M1 = ~H;
    const uint64_t M2 = M1 & (M1 << 1);
    const uint64_t M4 = M2 & (M2 << 2);
    const uint64_t M8 = M4 & (M4 << 4);
    const uint64_t X16 = M1 & 0xFFFF0000FFFF0000ull;
    const uint64_t P16 = X16 & ~(X16 + 0x0001000000010000ull);
    const uint32_t Z32 = (uint32_t)(M1 >> 32);
    const uint64_t P32 = (uint64_t)(Z32 & ~(Z32 + 1u)) << 32;

    float vv = tail;
    {
        float t;
        t = add_rn<INT>(vv, dpp_zero<DPP_ROW_SHR1, 0xF>(vv));
        vv = __builtin_amdgcn_inverse_ballot_w64(M1) ? t : vv;
        t = add_rn<INT>(vv, dpp_zero<DPP_ROW_SHR2, 0xF>(vv));
        vv = __builtin_amdgcn_inverse_ballot_w64(M2) ? t : vv;
        t = add_rn<INT>(vv, dpp_zero<DPP_ROW_SHR4, 0xF>(vv));
        vv = __builtin_amdgcn_inverse_ballot_w64(M4) ? t : vv;
        t = add_rn<INT>(vv, dpp_zero<DPP_ROW_SHR8, 0xF>(vv));
        vv = __builtin_amdgcn_inverse_ballot_w64(M8) ? t : vv;
        // The two broadcast steps write only the rows that have a source (row_mask); the add is fused into the DPP instruction
        // (the compiler's own form moves a zero, moves the lane, then adds: two more instructions per step) and the lanes of the
        // other rows keep a stale t that the step's mask, cleared for those rows, never selects. (s_nop 1: a DPP read of a
        // register written by the previous vector instruction needs two wait states; the compiler does not look inside asm.)
        if (INT) asm("s_nop 1\n\tv_add_u32_dpp %0, %1, %1 row_bcast:15 row_mask:0xa bank_mask:0xf" : "+v"(t) : "v"(vv));
        else asm("s_nop 1\n\tv_add_f32_dpp %0, %1, %1 row_bcast:15 row_mask:0xa bank_mask:0xf" : "+v"(t) : "v"(vv));  // lane 15 -> row 1, lane 47 -> row 3
        vv = __builtin_amdgcn_inverse_ballot_w64(P16) ? t : vv;
        if (INT) asm("s_nop 1\n\tv_add_u32_dpp %0, %1, %1 row_bcast:31 row_mask:0xc bank_mask:0xf" : "+v"(t) : "v"(vv));
        else asm("s_nop 1\n\tv_add_f32_dpp %0, %1, %1 row_bcast:31 row_mask:0xc bank_mask:0xf" : "+v"(t) : "v"(vv));  // lane 31 -> rows 2 and 3
        vv = __builtin_amdgcn_inverse_ballot_w64(P32) ? t : vv;
    }
    const float cin = dpp_zero<DPP_WAVE_SHR1, 0xF>(vv);  // lane l-1's inclusive sum; 0 for lane 0
    R.S = add_rn<INT>(cin, head);
    carry = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, vv), 63));
#pragma unroll
    for (int j = 0; j < C; ++j) R.m[j] = m[j];
    return R;
}

// The hot-path trigger: an upper bound of every finished row's score in this lane, three instructions. It is the maximum of
// the set {S, s[1], ..., s[C-1]}, which contains every finished row's sum; its other members are partial sums of rows that
// are still running, so it may fire for a packet without a passing row (the candidate path then finds none) but can never
// miss one. With non-negative products a partial sum reaches the threshold only if its row will.
// INT: the words are converted to fp32 score units first (round to nearest even, what C's (float)u32 does), like the row
// sums themselves.
template <int C, bool INT>
__device__ __forceinline__ float trigger_of(const Reduced<C> &R) {
    float c[C];
    c[0] = R.S;
#pragma unroll
    for (int j = 1; j < C; ++j) c[j] = R.s[j];
    if (INT) {
#pragma unroll
        for (int j = 0; j < C; ++j) c[j] = (float)__float_as_uint(c[j]);
    }
    const float NEG_INF = -__builtin_huge_valf();
    float best = max3(c[0], C > 1 ? c[1] : NEG_INF, C > 2 ? c[2] : NEG_INF);
#pragma unroll
    for (int j = 3; j < C; j += 2) best = max3(best, c[j], c[j + 1 < C ? j + 1 : j]);  // (v_max3: no canonicalising v_max in front)
    return best;
}

// The full row sums of a packet (candidate path, SpMV-only variant, exact group maxima): fl as in RowSums.
template <int C, bool INT>
__device__ __forceinline__ RowSums<C> expand(const Reduced<C> &R, const uint32_t fl) {
    RowSums<C> out;
    out.rs[0] = R.S;  // if entry 0 ends a row it is the lane's first row end
    uint32_t o = R.m[0];
#pragma unroll
    for (int j = 1; j < C; ++j) {
        out.rs[j] = mask_select(o, R.s[j], R.S);  // an earlier end in the lane => s_j
        o |= R.m[j];
    }
    if (INT) {
#pragma unroll
        for (int j = 0; j < C; ++j) out.rs[j] = (float)__float_as_uint(out.rs[j]);  // fixed-point word -> score units
    }
    out.fl = fl;
    const float NEG_INF = -__builtin_huge_valf();
    float best = NEG_INF;
#pragma unroll
    for (int j = 0; j < C; ++j) best = fmaxf(best, mask_select(R.m[j], out.rs[j], NEG_INF));
    out.best_any = best;
    return out;
}

// A and B of the split 12-bit plane out of the two dwords a lane loaded (even lane: [A][B_even | B_odd << 16], odd lane:
// [B_even | B_odd << 16][A]): two selects and one field extraction.
__device__ __forceinline__ void split_ab(const uint32_t d0, const uint32_t d1, uint32_t &A, uint32_t &B) {
    const bool odd = (threadIdx.x & 1u) != 0u;  // (lane parity: a wave is 64 consecutive threads)
    A = odd ? d1 : d0;
    const uint32_t bw = odd ? d0 : d1;
    B = odd ? bw >> 16 : bw & 0xFFFFu;
}

// Flag word (RowSums::fl) of the lane's entries of a packet.
template <int C, int QM>
__device__ __forceinline__ uint32_t packet_flags(const Pkt<C, value_type_of(QM)> &cur) {
    constexpr int VT = value_type_of(QM);
    if (VT == 4) {  // split 12-bit plane: ROW_END 0-3 at bits 12-15 of the halfword, SKIP 0, 1 in the dword, 2, 3 in the halfword
        uint32_t A, B;
        split_ab(cur.cw[0], cur.cw[1], A, B);
        return (B >> 12) | ((A & 3u) << 8) | ((B & 3u) << 10);
    }
    uint32_t fl = 0u;
#pragma unroll
    for (int j = 0; j < C; ++j) {
        const uint32_t w = (VT == 3 || VT == 6) ? __float_as_uint(cur.v[(VT == 3 || VT == 6) ? j : 0]) : (cur.cw[j >> 1] >> (16 * (j & 1)));
        fl |= ((w & 1u) << j) | (((w >> 1) & 1u) << (8 + j));
    }
    return fl;
}

// Products from a packet and the x vector staged in LDS (at LDS byte address xbase), then the reduction.
template <int C, int QM>
__device__ __forceinline__ Reduced<C> reduce_packet(const Pkt<C, value_type_of(QM)> &cur, float &carry, const uint32_t xbase,
                                                    const uint32_t fixed_mask = 0u) {
    constexpr int VT = value_type_of(QM);
    float p[C];
    uint32_t m[C];
    if (QM == 6) {
        // Bit-packed fixed point: word = value (bits 31..12, the top 20 bits of its Q1.31 word) | column << 2 | flags. x is
        // staged as a 20-bit integer (Q1.19); both factors fit the full-rate 24-bit multipliers; the 40-bit product is
        // Q2.38, of which bits 38..7 are the product in Q1.31 (integer part wrapped to one bit), masked to the width.
        uint32_t any = 0u;
#pragma unroll
        for (int j = 0; j < C; ++j) {
            const uint32_t w = __float_as_uint(cur.v[VT == 3 ? j : 0]);
            const uint32_t xq = lds_u32(xbase + (w & 0xFFCu));
            uint32_t v20, hi;
            asm("v_bfe_u32 %0, %1, 12, 20" : "=v"(v20) : "v"(w));
            asm("v_mul_hi_u32_u24 %0, %1, %2" : "=v"(hi) : "v"(v20), "v"(xq));
            p[j] = __uint_as_float(__builtin_amdgcn_alignbit(hi, __umul24(v20, xq), 7) & fixed_mask);
            m[j] = bit_mask<0>(w);
            any |= w;
        }
        return reduce_core<C, true>(p, m, (any & 1u) != 0u, carry);
    }
    if (QM == 8) {
        // Five bytes per entry (wbscsr.hpp FIXED26): D_j = value (bits 31..6) | column bits 3..0 << 2 | flags, E = column bits 9..4
        // of entry j at bits 6 j. The arithmetic is QM 4's: Q1.31 words, 24-bit multipliers up to 24 bits (x staged shifted down
        // by 8), quarter-rate 32-bit multiplies for 25 and 26.
        static_assert(QM != 8 || C == 4, "FIXED26 is built for 4 entries per lane");
        uint32_t any = 0u;
        const uint32_t E = cur.cw[0];
#pragma unroll
        for (int j = 0; j < C; ++j) {
            const uint32_t w = __float_as_uint(cur.v[VT == 6 ? j : 0]);
            uint32_t hi6;
            asm("v_bfe_u32 %0, %1, %2, 6" : "=v"(hi6) : "v"(E), "n"(6 * j));
            const uint32_t xq = lds_u32(xbase + ((w & 0x3Cu) | (hi6 << 6)));
            const uint32_t vq = w & 0xFFFFFFC0u;
            if (fixed_mask & 0xFFu) {
                p[j] = __uint_as_float(__builtin_amdgcn_alignbit(__umulhi(vq, xq), vq * xq, 31) & fixed_mask);
            } else {
                uint32_t hi;
                const uint32_t v24 = vq >> 8;
                asm("v_mul_hi_u32_u24 %0, %1, %2" : "=v"(hi) : "v"(v24), "v"(xq));
                p[j] = __uint_as_float(__builtin_amdgcn_alignbit(hi, __umul24(v24, xq), 15) & fixed_mask);
            }
            m[j] = bit_mask<0>(w);
            any |= w;
        }
        return reduce_core<C, true>(p, m, (any & 1u) != 0u, carry);
    }
    if (QM == 7) {
        // fp32 values, split 12-bit plane: A = col0 << 2 | col1 << 12 | col2 << 22 | SKIP0 | SKIP1 << 1,
        // B = col3 << 2 | SKIP2 | SKIP3 << 1 | ROW_END0..3 << 12. One instruction per LDS address where the field sits at
        // bit 2 (mask + base of the x copy), two where it has to be shifted down first; the row-end masks are single bit
        // extractions of B, and "this lane holds a row end" is one compare.
        static_assert(QM != 7 || C == 4, "the split 12-bit plane is built for 4 entries per lane");
        uint32_t mask = 0xFFCu;
        asm("" : "+v"(mask));  // (kept in a register: v_and_or_b32 takes no literal)
        uint32_t A, B;
        split_ab(cur.cw[0], cur.cw[VT == 4 ? 1 : 0], A, B);
        const uint32_t a0 = and_or(A, mask, xbase), a1 = and_or(A >> 10, mask, xbase), a2 = and_or(A >> 20, mask, xbase),
                       a3 = and_or(B, mask, xbase);
        p[0] = __fmul_rn(cur.v[VT == 4 ? 0 : 0], lds_f32(a0));
        p[C > 1 ? 1 : 0] = __fmul_rn(cur.v[VT == 4 ? 1 : 0], lds_f32(a1));
        p[C > 2 ? 2 : 0] = __fmul_rn(cur.v[VT == 4 ? 2 : 0], lds_f32(a2));
        p[C > 3 ? 3 : 0] = __fmul_rn(cur.v[VT == 4 ? 3 : 0], lds_f32(a3));
        m[0] = bit_mask<12>(B);
        m[C > 1 ? 1 : 0] = bit_mask<13>(B);
        m[C > 2 ? 2 : 0] = bit_mask<14>(B);
        m[C > 3 ? 3 : 0] = bit_mask<15>(B);
        return reduce_core<C, false>(p, m, B > 0xFFFu, carry);
    }
    uint32_t any = 0u;
#pragma unroll
    for (int j = 0; j < C; ++j) {
        const uint32_t word = cur.cw[j >> 1];
        const uint32_t off = xbase + ((j & 1) ? ((word >> 16) & 0xFFFCu) : (word & 0xFFFCu));  // LDS byte address of x[col]
        m[j] = (j & 1) ? bit_mask<16>(word) : bit_mask<0>(word);
        if ((j & 1) == 0) any |= word;
        if (QM == 5) {
            // x is staged as fp32 scaled by 2^-7: byte * (x / 128), one conversion and one multiply per entry
            p[j] = __fmul_rn(ubyte_to_float(cur.vq[VT == 1 ? (j >> 2) : 0], j & 3), lds_f32(off));
        } else if (VT == 1) {
            // x is staged as Q1.7 integers; product truncated to Q1.7 and wrapped to 8 bits, exact in fp32
            const uint32_t xq = lds_u32(off);
            const uint32_t vq = (cur.vq[VT == 1 ? (j >> 2) : 0] >> (8 * (j & 3))) & 255u;
            // both factors are below 2^8: the 24-bit multiply is exact (and full rate; v_mul_lo_u32 is quarter rate)
            const uint32_t t = __umul24(vq, xq);
            p[j] = (float)(QM == 2 ? (t >> 7) : ((t >> 7) & 255u));  // wide mode: no wrap
        } else if (VT == 2) {
            const uint32_t hw = cur.vq[VT == 2 ? (j >> 1) : 0];
            const _Float16 hv = __builtin_bit_cast(_Float16, (uint16_t)((j & 1) ? (hw >> 16) : (hw & 0xFFFFu)));
            p[j] = __fmul_rn((float)hv, lds_f32(off));  // the conversion is exact
        } else if (QM == 4) {
            // both factors are Q1.31 words: the 64-bit product is Q2.62; bits 31..62 are the product in Q1.31 (its integer
            // part wrapped to one bit, like an assignment to real_type), masked down to the W-1 fraction bits kept
            const uint32_t xq = lds_u32(off);
            const uint32_t vq = __float_as_uint(cur.v[VT == 0 ? j : 0]);
            if (fixed_mask & 0xFFu) {
                p[j] = __uint_as_float(__builtin_amdgcn_alignbit(__umulhi(vq, xq), vq * xq, 31) & fixed_mask);
            } else {
                // W <= 24: the low 8 bits of every word are zero and x was staged shifted down by 8, so both factors are
                // 24-bit integers (Q1.23) and the full-rate 24-bit multipliers give the 48-bit product (Q2.46), of which
                // bits 15..46 are the product in Q1.31 (v_mul_lo/hi_u32 run at quarter rate)
                uint32_t hi;
                const uint32_t v24 = vq >> 8;
                asm("v_mul_hi_u32_u24 %0, %1, %2" : "=v"(hi) : "v"(v24), "v"(xq));
                p[j] = __uint_as_float(__builtin_amdgcn_alignbit(hi, __umul24(v24, xq), 15) & fixed_mask);
            }
        } else {
            p[j] = __fmul_rn(cur.v[VT == 0 ? j : 0], lds_f32(off));
        }
    }
    // (`any` holds every column word of the lane: the odd entries' flags sit at bit 16 of the same words)
    return reduce_core<C, QM == 4>(p, m, (any & 0x00010001u) != 0u, carry);  // (QM 6 returned above)
}
template <int QM>
constexpr bool int_sums() { return QM == 4 || QM == 6 || QM == 8; }

template <int C, int QM>
__device__ __forceinline__ float row_score(const RowSums<C> &R, int j) {  // strict Q1.7: the 8-bit wrap of the row sum
    return QM == 1 ? q17_wrap(R.rs[j]) : R.rs[j];
}
template <int C, int QM>
__device__ __forceinline__ float lane_best(const RowSums<C> &R) {  // placeholders excluded
    float best = -__builtin_huge_valf();
#pragma unroll
    for (int j = 0; j < C; ++j) {
        const float sc = row_score<C, QM>(R, j);
        best = (R.valid(j) && sc > best) ? sc : best;
    }
    return best;
}

// Number of row ends in lower lanes (=> row id of this lane's first row end is rb + that).
template <int C>
__device__ __forceinline__ uint32_t ends_below(const RowSums<C> &R) {
    uint32_t below = 0;
#pragma unroll
    for (int j = 0; j < C; ++j) {
        const uint64_t b = __ballot(R.end(j));
        below += __builtin_amdgcn_mbcnt_hi((uint32_t)(b >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)b, 0u));
    }
    return below;
}

// Filter a wave's private candidate list against the threshold: lane l holds entries l, l+64, ... (EPL per lane);
// keep[] / pos[] tell which survive and where they go in the compacted order. Returns the number kept.
template <uint32_t EPL>
struct ListScan {
    uint2 e[EPL];
    uint32_t pos[EPL];
    bool keep[EPL];
};
template <uint32_t EPL>
__device__ __forceinline__ uint32_t scan_list(const uint2 *wcand, uint32_t n, float tau, uint32_t lane, ListScan<EPL> &L) {
    uint32_t total = 0;
#pragma unroll
    for (uint32_t u = 0; u < EPL; ++u) {
        const uint32_t i = lane + 64u * u;
        L.e[u] = make_uint2(0u, 0u);
        if (i < n) L.e[u] = wcand[i];
        L.keep[u] = i < n && __uint_as_float(L.e[u].x) >= tau;
        const uint64_t b = __ballot(L.keep[u]);
        L.pos[u] = total + __builtin_amdgcn_mbcnt_hi((uint32_t)(b >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)b, 0u));
        total += (uint32_t)__popcll(b);
    }
    return total;
}
// Drop what the (risen) threshold has made obsolete. All reads are issued before any write (LDS executes a wave's
// instructions in order), so writing the kept entries to the front cannot clobber an entry still to be read.
template <uint32_t EPL>
__device__ __forceinline__ uint32_t compact_list(uint2 *wcand, uint32_t n, float tau, uint32_t lane) {
    ListScan<EPL> L;
    const uint32_t kept = scan_list<EPL>(wcand, n, tau, lane, L);
#pragma unroll
    for (uint32_t u = 0; u < EPL; ++u)
        if (L.keep[u]) wcand[L.pos[u]] = L.e[u];
    return kept;
}

// Candidate path (rare once tau has converged). Every streaming wave owns a private list of WAVE_CAP entries in LDS
// (its length lives in an SGPR: no atomic, no other wave involved). A full list is first compacted against the
// current threshold; only what still does not fit goes to the shared overflow list in global memory, with ONE
// atomic per wave and packet. One LDS atomic raises the group maximum (the server wave pushes it to global memory).
// drop (workgroup-local thresholds, which the selection checks): a row that does not fit the list even after the compaction is
// not sent to a list in global memory but dropped, and the largest dropped score goes on record one step up (MISC_BOUND, an LDS
// atomic max; the workgroup's record of "everything I dropped lies strictly below this"): the selection's check fails unless k
// candidates reach every such record, and a failed check sends the query through the exact path.
template <int C, int QM, uint32_t WAVE_CAP, bool STATS = true>
__device__ __forceinline__ float offer_candidates(const StreamParams &P, const RowSums<C> &R, uint32_t rb, float tau,
                                                 uint32_t lane, uint32_t grp_local, bool publishes, uint2 *wcand,
                                                 uint32_t &wcnt, uint32_t *misc, const bool drop = false) {
    bool pass[C];
    uint32_t slot[C];
    uint32_t total = 0;
    const uint32_t below = ends_below<C>(R);
#pragma unroll
    for (int j = 0; j < C; ++j) {
        pass[j] = R.valid(j) && row_score<C, QM>(R, j) >= tau;
        const uint64_t pb = __ballot(pass[j]);
        slot[j] = total + __builtin_amdgcn_mbcnt_hi((uint32_t)(pb >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)pb, 0u));
        total += (uint32_t)__popcll(pb);
    }
    const float best = lane_best<C, QM>(R);
    const float wmax = wave_max(best >= tau ? best : -__builtin_huge_valf());
    if (total == 0u) return -__builtin_huge_valf();  // only placeholders of empty rows tripped the trigger
    if (lane == 0) {
        if (publishes)
            (void)__hip_atomic_fetch_max(&misc[MISC_GRPMAX + grp_local], order_key(wmax), __ATOMIC_RELAXED,
                                         __HIP_MEMORY_SCOPE_WORKGROUP);
        if (STATS && P.dbg) {  // TKSPMV_STATS=1: summed into global memory when the wave finishes
            atomicAdd(&misc[MISC_SLOW_CNT], 1u);
            atomicAdd(&misc[MISC_CAND_CNT], total);
        }
    }
    if (wcnt + total > WAVE_CAP) wcnt = compact_list<WAVE_CAP / 64u>(wcand, wcnt, tau, lane);
    const uint32_t base = wcnt;
    const uint32_t first_ovf = base < WAVE_CAP ? WAVE_CAP : base;  // list position of the first overflowing row
    uint32_t gbase = 0u;
    if (base + total > WAVE_CAP) {
        if (drop) {
            float dmax = -__builtin_huge_valf();
#pragma unroll
            for (int j = 0; j < C; ++j)
                if (pass[j] && base + slot[j] >= WAVE_CAP) dmax = fmaxf(dmax, row_score<C, QM>(R, j));
            dmax = wave_max(dmax);
            if (lane == 0)
                (void)__hip_atomic_fetch_max(&misc[MISC_BOUND], order_key(dmax) + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        } else {
            if (P.ovf_gate_lds != 0u) {  // (a list shared by several queries of the launch: its earlier users' selections must have finished)
                typedef __attribute__((address_space(3))) uint32_t lds_word;
                while (__builtin_amdgcn_readfirstlane(__hip_atomic_load(reinterpret_cast<lds_word *>((uintptr_t)P.ovf_gate_lds), __ATOMIC_RELAXED,
                                                                        __HIP_MEMORY_SCOPE_WORKGROUP)) != P.ovf_need)
                    __builtin_amdgcn_s_sleep(8);
                P.ovf_gate_lds = 0u;  // (open for the rest of this wave's query)
            }
            if (lane == 0) gbase = atomicAdd(P.ovf_count, base + total - first_ovf);
            gbase = __builtin_amdgcn_readfirstlane(gbase);
        }
    }
    uint32_t r = rb + below;
#pragma unroll
    for (int j = 0; j < C; ++j) {
        if (pass[j]) {
            const uint32_t pos = base + slot[j];
            if (pos < WAVE_CAP) {
                wcand[pos] = make_uint2(__float_as_uint(row_score<C, QM>(R, j)), r);
            } else if (!drop) {
                const uint32_t gp = gbase + (pos - first_ovf);
                if (gp < P.ovf_cap) st_agent(&P.ovf_cand[gp], pack_cand(__float_as_uint(row_score<C, QM>(R, j)), r));
            }
        }
        r += R.end(j) ? 1u : 0u;
    }
    wcnt = base + total < WAVE_CAP ? base + total : WAVE_CAP;
    return wmax;  // the best of the rows appended (wave-uniform): the batch kernel's workgroup-local thresholds build on it
}

}  // namespace tkspmv
