// Pieces of the fp32 LDS-DMA ring shared by the GEMM (gemm.hip) and the 3x3 convolution ring
// (conv.hip): stage geometry, fragment reads with the stage in the instruction immediate, the MFMA
// slab, the scalar-base form of the LDS-DMA.
#pragma once
#include "qarig_common.h"

namespace qarig {

typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef __attribute__((address_space(1))) const void* glb_ptr_t;

constexpr int DMA_OP_FLOATS = 128 * BK;              // 8 KB per operand per stage
constexpr int DMA_STAGE_FLOATS = 2 * DMA_OP_FLOATS;  // A then B

__device__ __forceinline__ unsigned lds_addr(const float* p) {
    return (unsigned)(uintptr_t)(__attribute__((address_space(3))) const float*)p;
}

struct Frags8 {
    f32x4 a0l, a0h, a1l, a1h, b0l, b0h, b1l, b1h;
};
// lgkmcnt(0) with the fragments passing THROUGH the wait: the inline-asm reads that produced them
// have no completion the compiler knows of, so without the data dependency nothing but instruction
// scheduling barriers keeps their first use behind the wait.
__device__ __forceinline__ void frags_wait(Frags8& f) {
    asm volatile("s_waitcnt lgkmcnt(0)"
                 : "+v"(f.a0l), "+v"(f.a0h), "+v"(f.a1l), "+v"(f.a1h), "+v"(f.b0l), "+v"(f.b0h), "+v"(f.b1l), "+v"(f.b1h)
                 :: "memory");
}
__device__ __forceinline__ void frags_mma(Acc& acc, const Frags8& f) {
#pragma unroll
    for (int s = 0; s < 8; ++s) {
        const float a0 = s < 4 ? f.a0l[s & 3] : f.a0h[s & 3];
        const float a1 = s < 4 ? f.a1l[s & 3] : f.a1h[s & 3];
        const float b0 = s < 4 ? f.b0l[s & 3] : f.b0h[s & 3];
        const float b1 = s < 4 ? f.b1l[s & 3] : f.b1h[s & 3];
        acc.t[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc.t[0][0], 0, 0, 0);
        acc.t[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc.t[0][1], 0, 0, 0);
        acc.t[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc.t[1][0], 0, 0, 0);
        acc.t[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc.t[1][1], 0, 0, 0);
    }
}

// The fp32 MFMA and the vector ALU do not overlap on this chip, so every address
// computation inside the k-loop is paid in full.  The loop below is unrolled over the 4 = lcm(2
// register sets, 4 stages) tiles of a period: every LDS address is (per-lane base VGPR, computed
// once) + an instruction immediate, every global address (uniform base advanced on the scalar
// unit) + (per-lane 32-bit offset, computed once).
struct FragBase {
    unsigned a0, a1, b0, b1;   // KC: the two 16-B chunks of this lane's row; XC: sub-tiles 0 and 1
};
template <int OFF>
__device__ __forceinline__ void rd_kc(unsigned c0, unsigned c1, f32x4& lo, f32x4& hi) {
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(lo) : "v"(c0), "n"(OFF));
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(hi) : "v"(c1), "n"(OFF));
}
template <int OFF>
__device__ __forceinline__ void rd_xc(unsigned base, f32x4& lo, f32x4& hi) {
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    static_assert(OFF % 256 == 0 && OFF / 256 + 14 < 256, "ds_read2st64 offsets are 8-bit units of 256 B");
    constexpr int U = OFF / 256;
    f32x2 p0, p1, p2, p3;
    asm volatile("ds_read2st64_b32 %0, %1 offset0:%2 offset1:%3" : "=v"(p0) : "v"(base), "n"(U), "n"(U + 2));
    asm volatile("ds_read2st64_b32 %0, %1 offset0:%2 offset1:%3" : "=v"(p1) : "v"(base), "n"(U + 4), "n"(U + 6));
    asm volatile("ds_read2st64_b32 %0, %1 offset0:%2 offset1:%3" : "=v"(p2) : "v"(base), "n"(U + 8), "n"(U + 10));
    asm volatile("ds_read2st64_b32 %0, %1 offset0:%2 offset1:%3" : "=v"(p3) : "v"(base), "n"(U + 12), "n"(U + 14));
    lo = f32x4{p0.x, p0.y, p1.x, p1.y};
    hi = f32x4{p2.x, p2.y, p3.x, p3.y};
}
template <bool AKC, bool BKC, int S>
__device__ __forceinline__ void frags_read_s(Frags8& f, const FragBase& fb) {
    constexpr int OA = S * DMA_STAGE_FLOATS * 4, OB = OA + DMA_OP_FLOATS * 4;
    if (AKC) rd_kc<OA>(fb.a0, fb.a1, f.a0l, f.a0h); else rd_xc<OA>(fb.a0, f.a0l, f.a0h);
    if (BKC) rd_kc<OB>(fb.b0, fb.b1, f.b0l, f.b0h); else rd_xc<OB>(fb.b0, f.b0l, f.b0h);
    if (AKC) rd_kc<OA + 32 * 64>(fb.a0, fb.a1, f.a1l, f.a1h); else rd_xc<OA>(fb.a1, f.a1l, f.a1h);
    if (BKC) rd_kc<OB + 32 * 64>(fb.b0, fb.b1, f.b1l, f.b1h); else rd_xc<OB>(fb.b1, f.b1l, f.b1h);
}
template <bool KC>
__device__ __forceinline__ void frag_bases(const float* lds, int wq, int x, int h, unsigned& p0, unsigned& p1) {
    if (KC) {
        const int row = wq * 64 + x, sw = (row >> 2) & 3;
        const unsigned base = lds_addr(lds + row * 16);
        p0 = base + (((2 * h) ^ sw) << 4);
        p1 = base + (((2 * h + 1) ^ sw) << 4);
    } else {
        p0 = lds_addr(lds + (8 * h) * 128 + wq * 64 + x);
        p1 = p0 + 32 * 4;
    }
}
// per-lane byte offset of DMA instruction q (of 8) of an operand tile from the tile's first element
template <bool KC>
__device__ __forceinline__ unsigned dma_lane_off(int64_t ld, int q, int lane) {
    if (KC) {
        const int r = 16 * q + (lane >> 2);
        const int c = (lane & 3) ^ ((r >> 2) & 3);
        return (unsigned)((r * ld + 4 * c) * 4);
    }
    const int k = 2 * q + (lane >> 5);
    return (unsigned)((k * ld + (lane & 31) * 4) * 4);
}

// One LDS-DMA instruction in its (scalar base + 32-bit lane offset) form: the k advance of the
// base stays on the scalar unit.  (The builtin takes a 64-bit vector address: one 64-bit vector add
// per instruction.)  M0 = LDS destination of lane 0; s_nop: the M0-write -> LDS-DMA wait state.
// M0 is not declared clobbered (a reserved register): nothing else in these kernels lives in it.
__device__ __forceinline__ void dma16_saddr(const char* base, unsigned lane_off, unsigned lds_dst) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1"
                 :: "v"(lane_off), "s"(base), "s"(lds_dst) : "memory");
}

constexpr int PF_STAGES = 4;

}  // namespace qarig
