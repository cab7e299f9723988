// Diagnostic: cost of a chain of N dependent small kernels on one stream, eager vs hipGraph replay.
// build: hipcc --offload-arch=gfx950 -O3 -o launch_chain launch_chain.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ void k_empty() {}
__global__ void k_touch(float* p) { if (threadIdx.x == 0) p[blockIdx.x] += 1.f; }
// load -> small compute -> store, one float4 per thread (like a decoder-step stage without MFMA)
__global__ void k_stream(const float4* __restrict__ a, float4* __restrict__ b, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { float4 v = a[i]; v.x += 1.f; b[i] = v; }
}

template <typename F>
double time_chain(hipStream_t s, int n, bool graph, F launch) {
    hipGraph_t g = nullptr; hipGraphExec_t ge = nullptr;
    if (graph) {
        hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal);
        for (int i = 0; i < n; ++i) launch(i);
        hipStreamEndCapture(s, &g);
        hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
    }
    auto run = [&]() { if (graph) hipGraphLaunch(ge, s); else for (int i = 0; i < n; ++i) launch(i); };
    run(); hipStreamSynchronize(s);
    double best = 1e30;
    for (int rep = 0; rep < 5; ++rep) {
        auto t0 = std::chrono::high_resolution_clock::now();
        run(); hipStreamSynchronize(s);
        auto t1 = std::chrono::high_resolution_clock::now();
        best = std::min(best, std::chrono::duration<double, std::micro>(t1 - t0).count());
    }
    if (ge) hipGraphExecDestroy(ge);
    if (g) hipGraphDestroy(g);
    return best / n;
}

int main() {
    hipStream_t s; CK(hipStreamCreate(&s));
    float *a, *b; CK(hipMalloc(&a, 64 << 20)); CK(hipMalloc(&b, 64 << 20));
    CK(hipMemset(a, 0, 64 << 20)); CK(hipMemset(b, 0, 64 << 20));
    const int N = 2000;
    for (int graph = 0; graph < 2; ++graph) {
        printf("%s:\n", graph ? "hipGraph replay" : "eager");
        printf("  empty 1x64            %.2f us/kernel\n", time_chain(s, N, graph, [&](int) { hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, s); }));
        printf("  empty 32x512          %.2f us/kernel\n", time_chain(s, N, graph, [&](int) { hipLaunchKernelGGL(k_empty, dim3(32), dim3(512), 0, s); }));
        printf("  empty 256x512         %.2f us/kernel\n", time_chain(s, N, graph, [&](int) { hipLaunchKernelGGL(k_empty, dim3(256), dim3(512), 0, s); }));
        printf("  touch 32x64           %.2f us/kernel\n", time_chain(s, N, graph, [&](int) { hipLaunchKernelGGL(k_touch, dim3(32), dim3(64), 0, s, a); }));
        printf("  stream 32x512 (256KB) %.2f us/kernel\n", time_chain(s, N, graph, [&](int i) {
            hipLaunchKernelGGL(k_stream, dim3(32), dim3(512), 0, s, (const float4*)((i & 1) ? a : b), (float4*)((i & 1) ? b : a), 32 * 512); }));
        printf("  stream 64x512 (512KB) %.2f us/kernel\n", time_chain(s, N, graph, [&](int i) {
            hipLaunchKernelGGL(k_stream, dim3(64), dim3(512), 0, s, (const float4*)((i & 1) ? a : b), (float4*)((i & 1) ? b : a), 64 * 512); }));
    }
    return 0;
}
