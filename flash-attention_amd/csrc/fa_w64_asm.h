// Inline-asm helpers shared by the 64-rows-per-wave kernels (fa_fwd_w64.hip, fa_bwd_w64.hip): accumulator registers that are
// owned by the asm (named literally, never C++ values), the drains hipcc does not insert for asm MFMAs, and a few
// single-instruction VALU helpers.  The including file defines FA_W64_CLOB = the clobber list of ITS asm-owned accumulators
// (fa_fwd_w64_regs.h) first; the helpers live in an anonymous namespace because their bodies differ with that list.
#pragma once
#include <type_traits>
#include <utility>

#include "fa_device.h"

#ifndef FA_W64_CLOB
#error "define FA_W64_CLOB (accumulator clobber list) before including fa_w64_asm.h"
#endif

namespace fa {
namespace {

template <int N> using ICw = std::integral_constant<int, N>;
template <int... I, class F> FA_DEVINL void static_for_impl(std::integer_sequence<int, I...>, F&& f) { (f(ICw<I>{}), ...); }
template <int N, class F> FA_DEVINL void static_for(F&& f) { static_for_impl(std::make_integer_sequence<int, N>{}, f); }

// hipcc neither sees the asm MFMAs' latency nor pads their hazards: an MFMA result may be read by a non-MFMA instruction
// only 12+ wait states after issue.  The hand-placed step keeps that distance by construction; every other reader drains.
// The drained VGPR tuples are tied operands of the drain: a "memory" clobber does not order register-only instructions, so
// without the data dependence hipcc may schedule a reader above the nops.
FA_DEVINL void mfma_drain_v(f32x16& a, f32x16& b) { asm volatile("s_nop 15\n\ts_nop 3" : "+v"(a), "+v"(b)); }
FA_DEVINL void mfma_drain_acc() { asm volatile("s_nop 15\n\ts_nop 3" ::: FA_W64_CLOB); }   // volatile asm keeps its order among the asm accessors
// accumulator register N: write / read / multiply by a per-lane factor (N is a compile-time constant)
template <int N> FA_DEVINL void acc_write(float x) { asm volatile("v_accvgpr_write_b32 a[%c1], %0" : : "v"(x), "i"(N) : FA_W64_CLOB); }
template <int N> FA_DEVINL float acc_read() { float x; asm volatile("v_accvgpr_read_b32 %0, a[%c1]" : "=v"(x) : "i"(N) : FA_W64_CLOB); return x; }
template <int N> FA_DEVINL void acc_scale(float f) {
  float t;
  asm volatile("v_accvgpr_read_b32 %0, a[%c2]\n\ts_nop 0\n\tv_mul_f32 %0, %0, %1\n\ts_nop 0\n\tv_accvgpr_write_b32 a[%c2], %0" : "=&v"(t) : "v"(f), "i"(N) : FA_W64_CLOB);
}
// two registers per statement, their instructions interleaved: no instruction consumes its predecessor's result, so no pads
// (3 instructions per register instead of 5 -- this is cold code, but two copies of 128 registers' worth sit in the tile loop's body)
template <int N> FA_DEVINL void acc_scale2(float f) {
  float t, u;
  asm volatile("v_accvgpr_read_b32 %0, a[%c3]\n\tv_accvgpr_read_b32 %1, a[%c4]\n\tv_mul_f32 %0, %0, %2\n\tv_mul_f32 %1, %1, %2\n\t"
               "v_accvgpr_write_b32 a[%c3], %0\n\tv_accvgpr_write_b32 a[%c4], %1"
               : "=&v"(t), "=&v"(u) : "v"(f), "i"(N), "i"(N + 1) : FA_W64_CLOB);
}
template <int N0, int... I> FA_DEVINL void acc_scale_range(float f, std::integer_sequence<int, I...>) {
  static_assert(sizeof...(I) % 2 == 0, "even register count");
  ((I % 2 == 0 ? acc_scale2<N0 + I>(f) : (void)0), ...);
}
template <int N0, int... I> FA_DEVINL void acc_zero_range(std::integer_sequence<int, I...>) { (acc_write<N0 + I>(0.f), ...); }
// accumulator tuples T0 .. (16 registers each) = 0 by the matrix pipe: D = 0 . 0 + 0 (zf = four zero registers, freshly written: two wait states first)
template <int T> FA_DEVINL void acc_zero_tuple_mfma(const u32x4& zf) {
  asm volatile("v_mfma_f32_32x32x16_bf16 a[%c1:%c2], %0, %0, 0" : : "v"(zf), "i"(16 * T), "i"(16 * T + 15) : FA_W64_CLOB);
}
template <int T0, int... I> FA_DEVINL void acc_zero_tuples_mfma(const u32x4& zf, std::integer_sequence<int, I...>) {
  asm volatile("s_nop 1" : : "v"(zf));
  (acc_zero_tuple_mfma<T0 + I>(zf), ...);
}
// (elements are copied to scalars first: __builtin_bit_cast applied directly to an ext_vector element lvalue reads element 0)
template <int N0> FA_DEVINL void acc_read8(float (&y)[8]) {   // eight registers per statement (hipcc pads every asm statement with an s_nop)
  asm volatile("v_accvgpr_read_b32 %0, a[%c8]\n\tv_accvgpr_read_b32 %1, a[%c9]\n\tv_accvgpr_read_b32 %2, a[%c10]\n\tv_accvgpr_read_b32 %3, a[%c11]\n\t"
               "v_accvgpr_read_b32 %4, a[%c12]\n\tv_accvgpr_read_b32 %5, a[%c13]\n\tv_accvgpr_read_b32 %6, a[%c14]\n\tv_accvgpr_read_b32 %7, a[%c15]"
               : "=v"(y[0]), "=v"(y[1]), "=v"(y[2]), "=v"(y[3]), "=v"(y[4]), "=v"(y[5]), "=v"(y[6]), "=v"(y[7])
               : "i"(N0), "i"(N0 + 1), "i"(N0 + 2), "i"(N0 + 3), "i"(N0 + 4), "i"(N0 + 5), "i"(N0 + 6), "i"(N0 + 7) : FA_W64_CLOB);
}
template <int N0> FA_DEVINL void acc_read_tuple(f32x16& x) {
  float lo[8], hi[8];
  acc_read8<N0>(lo);
  acc_read8<N0 + 8>(hi);
#pragma unroll
  for (int i = 0; i < 8; ++i) { x[i] = lo[i]; x[8 + i] = hi[i]; }
}
template <int N0> FA_DEVINL void acc_write_frag(u32x4 w) {
  const unsigned w0 = w[0], w1 = w[1], w2 = w[2], w3 = w[3];
  acc_write<N0 + 0>(__builtin_bit_cast(float, w0)); acc_write<N0 + 1>(__builtin_bit_cast(float, w1));
  acc_write<N0 + 2>(__builtin_bit_cast(float, w2)); acc_write<N0 + 3>(__builtin_bit_cast(float, w3));
}
// single-instruction float helpers (clang would canonicalise the asm MFMA outputs in front of fmaxf)
FA_DEVINL float vmax3(float a, float b, float c) { float r; asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }
FA_DEVINL float vmax2(float a, float b) { float r; asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
// max of a value with the one held by lane (l ^ 32): copy, swap halves, max -- one statement so that the permlane hazard pad
// (2 wait states between a VALU write of an operand and v_permlane32_swap) sits inside it; no canonicalising v_max
FA_DEVINL float vhalf_max(float x) {
  float t;
  asm("v_mov_b32 %1, %0\n\ts_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\tv_max_f32 %0, %0, %1" : "+v"(x), "=&v"(t));
  return x;
}
// NOTE on asm helpers: hipcc's hazard recogniser does not look inside an asm statement.  gfx950 needs one wait state between
// a transcendental (v_exp_f32) and a non-transcendental VALU that consumes its result; an asm v_add_f32 placed right behind
// the v_exp that feeds it reads a stale register (seen as run-to-run different row sums).  So nothing that consumes a v_exp
// result is asm: the row sums are plain C++ adds, and this translation unit is built with -fno-slp-vectorize (build.py) --
// the SLP vectoriser otherwise packs them into v_pk_add_f32 (slower beside MFMAs) and moves them out of their gaps.


}  // namespace
}  // namespace fa
