// wm_device.h — device-side building blocks shared by the gfx950 kernels: operand types, 8-wide K fragments,
// the mma32 wrapper over MFMA 16x16 (bf16 / f16 / exact f32), wave64 reductions, activation functions.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace wm {

typedef __bf16 bf16;
typedef _Float16 f16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(4))) _Float16 f16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) float f32x8;

// ---- K fragment: 8 consecutive K elements of one row, held by lane (row = lane&15, kgroup = lane>>4) -------
// For 16-bit types this is exactly the A/B operand of v_mfma_f32_16x16x32_{bf16,f16}
// (A[row l&15][k = 8(l>>4)+j]).  For fp32 the same ownership feeds eight v_mfma_f32_16x16x4_f32, MFMA j taking
// element j of every lane (contracting k = 8g + j over g = 0..3) — A and B use the same map, so the sum over the
// 32-deep K chunk is complete and each product is an exact fp32 fma (no reduced-precision path on gfx950).
template <typename T> struct Frag;
template <> struct Frag<float> { f32x8 v; };
template <> struct Frag<bf16> { bf16x8 v; };
template <> struct Frag<f16> { f16x8 v; };

template <typename T> __device__ __forceinline__ Frag<T> load_frag(const T* p);  // p -> 8 contiguous elements
template <> __device__ __forceinline__ Frag<float> load_frag<float>(const float* p) {
    Frag<float> f;
    f32x4 a = *reinterpret_cast<const f32x4*>(p), b = *reinterpret_cast<const f32x4*>(p + 4);
    f.v = f32x8{a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
    return f;
}
template <> __device__ __forceinline__ Frag<bf16> load_frag<bf16>(const bf16* p) {
    Frag<bf16> f;
    f.v = *reinterpret_cast<const bf16x8*>(p);
    return f;
}
template <> __device__ __forceinline__ Frag<f16> load_frag<f16>(const f16* p) {
    Frag<f16> f;
    f.v = *reinterpret_cast<const f16x8*>(p);
    return f;
}

template <typename T> __device__ __forceinline__ T from_f32(float x) { return (T)x; }

// fragment from 8 fp32 values (activations normalised on the fly)
template <typename T> __device__ __forceinline__ Frag<T> make_frag(const float (&x)[8]) {
    Frag<T> f;
#pragma unroll
    for (int j = 0; j < 8; ++j) f.v[j] = from_f32<T>(x[j]);
    return f;
}

// acc(16x16 f32, C/D map: col = lane&15, row = 4*(lane>>4)+reg) += A(16 x 32) · B(32 x 16)
__device__ __forceinline__ f32x4 mma32(const Frag<float>& a, const Frag<float>& b, f32x4 c) {
#pragma unroll
    for (int j = 0; j < 8; ++j) c = __builtin_amdgcn_mfma_f32_16x16x4f32(a.v[j], b.v[j], c, 0, 0, 0);
    return c;
}
__device__ __forceinline__ f32x4 mma32(const Frag<bf16>& a, const Frag<bf16>& b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.v, b.v, c, 0, 0, 0);
}
__device__ __forceinline__ f32x4 mma32(const Frag<f16>& a, const Frag<f16>& b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(a.v, b.v, c, 0, 0, 0);
}

// ---- wave64 cross-lane helpers ---------------------------------------------------------------------------------
template <int CTRL> __device__ __forceinline__ float dpp_f32(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
// all-reduce (sum) over aligned groups of 8 / 16 lanes using DPP only (no LDS crossbar)
__device__ __forceinline__ float group_sum8(float v) {
    v += dpp_f32<0xB1>(v);   // quad_perm [1,0,3,2]
    v += dpp_f32<0x4E>(v);   // quad_perm [2,3,0,1]
    v += dpp_f32<0x141>(v);  // row_half_mirror
    return v;
}
__device__ __forceinline__ float group_sum16(float v) {
    v = group_sum8(v);
    v += dpp_f32<0x140>(v);  // row_mirror
    return v;
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// ---- activations -----------------------------------------------------------------------------------------------
// mode 0: tanh approximation, the reference's formula and constants (whisper_tensor.mojo:288-308);
// mode 1: exact erf (HF).  Accurate libm-grade tanhf/erff: no fast-math on the parity path.
__device__ __forceinline__ float gelu_f(float x, int mode) {
    if (mode == 0) {
        const float SQRT_2_PI = 0.79788456f, COEFF = 0.044715f;
        float x3 = x * x * x;
        float inner = SQRT_2_PI * (x + COEFF * x3);
        return 0.5f * x * (1.0f + tanhf(inner));
    }
    return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f));
}

// 16-bit operand modes: the tanh form as one exp2 + one reciprocal, 0.5x(1+tanh(u)) = x / (1 + exp(-2u)) with
// -2u·log2(e) = x·(K1 + K2·x²) folded into two constants — the accurate tanhf costs ~40 VALU instructions per element, and even
// the 9-instruction form of round 1 made fc1's GELU epilogue (147 M elements at 64 clips) cost more than its MFMA main loop.
// Relative error ~1e-6, far below the bf16/f16 rounding of the stored result.  x -> -inf gives exp2 = inf, 1/inf = 0: -0.
__device__ __forceinline__ float gelu_fast(float x, int mode) {
    if (mode == 0) {
        constexpr float K1 = -2.0f * 0.79788456f * 1.4426950408889634f, K2 = K1 * 0.044715f;
        const float e = x * __builtin_fmaf(x * x, K2, K1);
        return x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(e));
    }
    return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f));
}
// the same on four values: the polynomial, the +1 and the final product as packed fp32 (v_pk_mul / v_pk_fma / v_pk_add)
__device__ __forceinline__ f32x4 gelu_fast4(f32x4 x, int mode) {
    if (mode == 0) {
        constexpr float K1 = -2.0f * 0.79788456f * 1.4426950408889634f, K2 = K1 * 0.044715f;
        const f32x4 k1 = f32x4{K1, K1, K1, K1}, k2 = f32x4{K2, K2, K2, K2}, one = f32x4{1.f, 1.f, 1.f, 1.f};
        const f32x4 e = x * __builtin_elementwise_fma(x * x, k2, k1);
        f32x4 d;
#pragma unroll
        for (int r = 0; r < 4; ++r) d[r] = __builtin_amdgcn_exp2f(e[r]);
        d += one;
#pragma unroll
        for (int r = 0; r < 4; ++r) d[r] = __builtin_amdgcn_rcpf(d[r]);
        return x * d;
    }
    f32x4 y;
#pragma unroll
    for (int r = 0; r < 4; ++r) y[r] = gelu_fast(x[r], mode);
    return y;
}

}  // namespace wm
