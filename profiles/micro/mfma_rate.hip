// Micro-benchmark: issue rate of the fp32 MFMA shapes on gfx950 (cycles per instruction per SIMD, one and two waves per
// SIMD, four independent accumulator chains):  v_mfma_f32_4x4x1_16B_f32, 16x16x4, 32x32x2.
//   hipcc --offload-arch=gfx950 -O3 -o profiles/micro/mfma_rate profiles/micro/mfma_rate.hip && ./profiles/micro/mfma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f16 __attribute__((ext_vector_type(16)));
constexpr int ITERS = 2048;
template <int KIND>
__global__ void k(float* out, float a0) {
    const float a = a0 + threadIdx.x, b = a0 * 0.5f + threadIdx.x;
    f4 c[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
    f16 d[2] = {};
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (KIND == 0) c[j] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c[j], 0, 0, 0);
            if (KIND == 1) c[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c[j], 0, 0, 0);
            if (KIND == 2) d[j & 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, d[j & 1], 0, 0, 0);
        }
    }
    float r = 0;
    for (int j = 0; j < 4; ++j) r += c[j][0] + c[j][3];
    r += d[0][0] + d[1][5];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}
template <int KIND>
void run(const char* name, float* out) {
    for (int wps : {1, 2}) {
        hipLaunchKernelGGL(k<KIND>, dim3(256), dim3(256 * wps), 0, 0, out, 1.0f);
        hipDeviceSynchronize();
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0);
        hipLaunchKernelGGL(k<KIND>, dim3(256), dim3(256 * wps), 0, 0, out, 1.0f);
        hipEventRecord(e1); hipDeviceSynchronize();
        float ms = 0; hipEventElapsedTime(&ms, e0, e1);
        // per SIMD: wps waves x ITERS x 4 instructions
        printf("%-28s waves/SIMD %d: %.3f ms = %.1f cycles per instruction per SIMD at 2.4 GHz\n", name, wps, ms,
               ms * 1e-3 * 2.4e9 / ((double)ITERS * 4 * wps));
    }
}
int main() {
    float* out; hipMalloc(&out, 256 * 1024 * sizeof(float));
    run<0>("v_mfma_f32_4x4x1_16B_f32", out);
    run<1>("v_mfma_f32_16x16x4_f32", out);
    run<2>("v_mfma_f32_32x32x2_f32", out);
    return 0;
}
