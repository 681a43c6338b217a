// Device helpers shared by the Winograd F(2x2x2, 3x3x3) kernels (conv_wino.hip: forward / data gradient,
// conv_wino_wgrad.hip: weight gradient).  gfx950 only.
#pragma once
#include "common.h"

namespace mis_wino {

using mis_dma::i32x4;

typedef float f32x2 __attribute__((ext_vector_type(2)));

// Packed fp32 adds on register pairs, written as asm: hipcc scalarises <2 x float> arithmetic whose results are read
// one half at a time (the MFMA operands), and it sinks a C++ transform of chunk s+1 into the next iteration, in front
// of the MFMAs that consume it.  Volatile asm keeps the program order of the slots below.
#ifndef MIS_WINO_CXX_TRANSFORM
#define MIS_WINO_CXX_TRANSFORM 0
#endif
#if MIS_WINO_CXX_TRANSFORM & 1
__device__ __forceinline__ f32x2 pk_add(f32x2 a, f32x2 b) { return a + b; }
__device__ __forceinline__ f32x2 pk_sub(f32x2 a, f32x2 b) { return a - b; }
#else
__device__ __forceinline__ f32x2 pk_add(f32x2 a, f32x2 b) {
    f32x2 r; asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r;
}
__device__ __forceinline__ f32x2 pk_sub(f32x2 a, f32x2 b) {
    f32x2 r; asm volatile("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(b)); return r;
}
#endif
// rows 2h, 2h+1 of an accumulator tile, read from its AGPRs at this point of the program
__device__ __forceinline__ f32x2 acc_pair(const f32x4& q, int h) {
    float lo, hi;
    asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(lo) : "a"(q[2 * h]));
    asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(hi) : "a"(q[2 * h + 1]));
    return f32x2{lo, hi};
}
// B^T d = (d0 - d2, d1 + d2, d2 - d1, d1 - d3) on four pairs (elementwise)
__device__ __forceinline__ void bt4(f32x2& a, f32x2& b, f32x2& c, f32x2& d) {
    const f32x2 t0 = pk_sub(a, c), t3 = pk_sub(b, d), t1 = pk_add(b, c), t2 = pk_sub(c, b);
    a = t0; b = t1; c = t2; d = t3;
}
// ... and inside two pairs p0 = (d0, d1), p1 = (d2, d3): op_sel picks the halves
__device__ __forceinline__ void bt4_inner(f32x2& p0, f32x2& p1) {
#if MIS_WINO_CXX_TRANSFORM & 2
    const f32x2 c0 = {p0[0] - p1[0], p0[1] + p1[0]}, c1 = {p1[0] - p0[1], p0[1] - p1[1]};
    p0 = c0; p1 = c1;
    return;
#endif
    f32x2 q0, q1;
    asm volatile("v_pk_add_f32 %0, %1, %2 op_sel_hi:[1,0] neg_lo:[0,1]" : "=v"(q0) : "v"(p0), "v"(p1));            // (d0 - d2, d1 + d2)
    // (d2 - d1, d1 - d3) with the half-crossing operand as src0.  HARDWARE ERRATUM (gfx950, round 5): a packed fp32 op whose LOW
    // result lane takes the low half of src0 and the HIGH half of another source register (op_sel:[0,1...]) returns wrong values
    // while a foreign wave on the same SIMD runs v_mfma_f32_16x16x32_bf16 (scripts/ubench/pk_hazard.hip: a quarter of the lanes
    // differ from the quiet run, for v_pk_add / v_pk_mul / v_pk_fma alike; op_sel:[1,0], op_sel:[1,1] and same-register sources
    // are fine, fp32-MFMA and VALU neighbours too).  The form this line had until round 5 -- src0 = p1, src1 = p0, op_sel:[0,1]
    // -- made every Winograd convolution that shared a SIMD with SwinUnet's bf16x3 attention waves return wrong rows
    // (cross teaching; scripts/interference.py); scripts/check_pk_opsel.py rejects the pattern in the ISA of the built library (Makefile `all`, CPU test).
    asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[1,1] neg_lo:[1,0] neg_hi:[0,1]" : "=v"(q1) : "v"(p0), "v"(p1));
    p0 = q0; p1 = q1;
}

// The 4x4x4 patch of a lane lives in 32 register pairs u[(z*4 + y)*2 + x/2] = (x even, x odd) -- what one ds_read2_b32
// of two neighbouring floats delivers.  Every pass is packed adds (v_pk_add_f32), 96 instructions instead of 192.
// 24 units of 4 instructions: 0..7 x pass of two (z, y) rows, 8..15 y pass of (z, x pair), 16..23 z pass of (y, x pair).
template <int U>
__device__ __forceinline__ void in_unit(f32x2 (&u)[32]) {
    if constexpr (U < 8) {
        bt4_inner(u[U * 4], u[U * 4 + 1]);
        bt4_inner(u[U * 4 + 2], u[U * 4 + 3]);
    } else if constexpr (U < 16) {
        constexpr int z = (U - 8) / 2, xp = (U - 8) % 2, b = z * 8 + xp;
        bt4(u[b], u[b + 2], u[b + 4], u[b + 6]);
    } else if constexpr (U < 24) {
        constexpr int j = U - 16;            // (y, x pair)
        bt4(u[j], u[8 + j], u[16 + j], u[24 + j]);
    }
}

template <int T, int END>
__device__ __forceinline__ void in_units(f32x2 (&u)[32]) {
    if constexpr (T < END) { in_unit<T>(u); in_units<T + 1, END>(u); }
}

// dma_dwordx4 with the uniform part of the address in the instruction's scalar offset (no per-lane add)
__device__ __forceinline__ void dma_dwordx4_s(unsigned lds_byte, unsigned voff, unsigned soff, i32x4 rsrc) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\t"
                 "buffer_load_dwordx4 %2, %3, %4 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "s"(lds_byte), "v"(voff), "s"(rsrc), "s"(soff) : "memory");
}

__device__ __forceinline__ void dma_dword_s(unsigned lds_byte, unsigned voff, unsigned soff, i32x4 rsrc) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\t"
                 "buffer_load_dword %2, %3, %4 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "s"(lds_byte), "v"(voff), "s"(rsrc), "s"(soff) : "memory");
}

template <int N> struct vmwait { static __device__ __forceinline__ void go() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); } };


}  // namespace mis_wino
