// Micro-benchmark: issue rate of fp32 VALU forms on gfx950 (cycles per wave-instruction per SIMD), at 1, 2 and 4
// waves per SIMD: v_add_f32, v_fma_f32, v_pk_add_f32, v_pk_mul_f32, v_pk_fma_f32. Independent accumulators (8 chains).
//   hipcc --offload-arch=gfx950 -O3 -o profiles/micro/valu_rate profiles/micro/valu_rate.hip && ./profiles/micro/valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));
constexpr int ITERS = 4096, CH = 8;

template <int KIND>
__global__ void k(float* out, unsigned long long* cyc) {
    f2 a[CH];
    float s[CH];
    const float x = out[threadIdx.x & 3] + 1.0f;
    const f2 xv = {x, x + 1.f};
#pragma unroll
    for (int i = 0; i < CH; ++i) { a[i] = f2{(float)i, (float)(i + 1)} + xv; s[i] = (float)i + x; }
    const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int i = 0; i < CH; ++i) {
            if (KIND == 0) asm volatile("v_add_f32 %0, %0, %1" : "+v"(s[i]) : "v"(x));
            if (KIND == 1) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(s[i]) : "v"(x));
            if (KIND == 2) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(xv));
            if (KIND == 3) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(xv));
            if (KIND == 4) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(a[i]) : "v"(xv));
            if (KIND == 5) asm volatile("v_pk_add_f32 %0, %0, %1 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]" : "+v"(a[i]) : "v"(xv));
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
    float r = 0.f;
#pragma unroll
    for (int i = 0; i < CH; ++i) r += a[i].x + a[i].y + s[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
    if (threadIdx.x == 0) { cyc[blockIdx.x] = t1 - t0; cyc[256 + blockIdx.x] = r1 - r0; }
}

template <int KIND>
void run(const char* name, float* out, unsigned long long* cyc) {
    for (int wps : {1, 2, 4}) {            // waves per SIMD: one block per CU of 4 * wps waves
        const int threads = 256 * wps;
        hipLaunchKernelGGL(k<KIND>, dim3(256), dim3(threads), 0, 0, out, cyc);
        hipDeviceSynchronize();
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0);
        hipLaunchKernelGGL(k<KIND>, dim3(256), dim3(threads), 0, 0, out, cyc);
        hipEventRecord(e1);
        hipDeviceSynchronize();
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        unsigned long long h[512];
        hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
        double avg = 0, rt = 0;
        for (int i = 0; i < 256; ++i) { avg += (double)h[i]; rt += (double)h[256 + i]; }
        avg /= 256; rt /= 256;
        // memtime ticks per wave-instruction issued on one SIMD: wps waves each issue ITERS*CH instructions
        // s_memrealtime: constant 100 MHz; the loop's wall time per wave = rt * 10 ns
        printf("%-22s waves/SIMD %d: %.2f memtime ticks per instruction per SIMD (%.3f ms launch, %.0f ticks/wave, loop %.1f us by the 100 MHz clock = %.2f ns per instruction per SIMD)\n",
               name, wps, avg / ((double)ITERS * CH * wps), ms, avg, rt * 0.01, rt * 10.0 / ((double)ITERS * CH * wps));
    }
}

int main() {
    float* out; unsigned long long* cyc;
    hipMalloc(&out, 256 * 1024 * sizeof(float)); hipMemset(out, 0, 256 * 1024 * sizeof(float));
    hipMalloc(&cyc, 512 * sizeof(unsigned long long));
    run<0>("v_add_f32", out, cyc);
    run<1>("v_fma_f32", out, cyc);
    run<2>("v_pk_add_f32", out, cyc);
    run<3>("v_pk_mul_f32", out, cyc);
    run<4>("v_pk_fma_f32", out, cyc);
    run<5>("v_pk_add_f32 op_sel/neg", out, cyc);
    return 0;
}
