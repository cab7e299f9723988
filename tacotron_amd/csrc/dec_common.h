// Device helpers shared by the decoder-step kernels (decoder.hip) and the persistent decoder (dec_persistent.hip).
#pragma once
#include <hip/hip_runtime.h>

namespace taco {

typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float dsigmoid(float x) { return 1.f / (1.f + expf(-x)); }
__device__ __forceinline__ float dtanh(float x) {
    const float e = expf(2.f * x);
    return 1.f - 2.f / (e + 1.f);
}

// tanh for the attention scores (T_in*256 evaluations per utterance-step): hardware exp2 + rcp,
// 1 - 2/(2^(2x log2 e) + 1); absolute error ~2e-7, saturates cleanly. The GRU epilogues keep expf.
__device__ __forceinline__ float fast_tanh(float x) {
    const float e = __builtin_amdgcn_exp2f(x * 2.8853900817779268f);
    return 1.f - 2.f * __builtin_amdgcn_rcpf(e + 1.f);
}

__device__ __forceinline__ float wave_sum(float x) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) x += __shfl_xor(x, o);
    return x;
}
__device__ __forceinline__ float wave_max(float x) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) x = fmaxf(x, __shfl_xor(x, o));
    return x;
}

}  // namespace taco
