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

// Wave-wide reductions on the DPP cross-lane path (no LDS round trips: a __shfl_xor butterfly is six dependent
// ds_bpermute, ~100 cycles each): quad swaps, half-row and row mirrors give every lane its 16-lane row total,
// the four row totals are combined through readlane. Every lane returns the same value.
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true));
}
__device__ __forceinline__ float wave_sum(float x) {
    x += dpp_mov<0xB1>(x);    // quad_perm [1,0,3,2]
    x += dpp_mov<0x4E>(x);    // quad_perm [2,3,0,1]
    x += dpp_mov<0x141>(x);   // row_half_mirror
    x += dpp_mov<0x140>(x);   // row_mirror
    return (__int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), 0)) +
            __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), 16))) +
           (__int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), 32)) +
            __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), 48)));
}
__device__ __forceinline__ float wave_max(float x) {
    x = fmaxf(x, dpp_mov<0xB1>(x));
    x = fmaxf(x, dpp_mov<0x4E>(x));
    x = fmaxf(x, dpp_mov<0x141>(x));
    x = fmaxf(x, dpp_mov<0x140>(x));
    return fmaxf(fmaxf(__int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), 0)),
                       __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), 16))),
                 fmaxf(__int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), 32)),
                       __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), 48))));
}

}  // namespace taco
