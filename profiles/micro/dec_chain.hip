// Diagnostic: the decoder step's kernel chain (11 dec_gemm + 1 dec_attention per step) on dummy data, graph replay,
// timed per stage family by leaving stages out.  Includes the product kernels directly.
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I../../tacotron_amd/csrc -o dec_chain dec_chain.hip
#include "../../tacotron_amd/csrc/decoder.hip"
#include <chrono>
#include <cstdio>
using namespace taco;
void Timing::begin(int, hipStream_t) {}
void Timing::end(int, hipStream_t) {}

int main(int argc, char** argv) {
    const int B = argc > 1 ? atoi(argv[1]) : 32, T_in = 150, S = 200, Bp = (B + 15) / 16 * 16;
    hipStream_t s; (void)hipStreamCreate(&s);
    float* pool; (void)hipMalloc(&pool, 256 << 20); (void)hipMemset(pool, 0, 256 << 20);
    size_t off = 0;
    auto take = [&](size_t n) { float* p = pool + off; off += (n + 63) / 64 * 64; return p; };
    float* W = take(2 << 20);   // 8 MB of packed weights; each GEMM of a step reads its own region
    size_t woff = 0;
    float *keys = take((size_t)B * T_in * 256), *vals = take((size_t)B * T_in * 256), *v = take(256);
    float *go = take(80 * Bp), *ctx = take(256 * Bp), *hatt = take(256 * Bp), *h1 = take(256 * Bp), *h2 = take(256 * Bp);
    float *p1 = take(256 * Bp), *p2 = take(128 * Bp), *rh = take(256 * Bp), *u = take(256 * Bp), *q = take(256 * Bp);
    float *d0 = take(256 * Bp), *o1 = take(256 * Bp), *o2 = take(256 * Bp), *outQ = take((size_t)S * 400 * Bp);
    float* al = take((size_t)S * B * T_in);
    int32_t* nz = (int32_t*)take((size_t)S * Bp);
    float* bias = take(512);
    auto gemm = [&](const float* s0, int k0, const float* s1, int k1, int N, int epi, float* out, float* hh) {
        DecGemmArgs a{}; a.B = B; a.Bpad = Bp; a.src0 = s0; a.k16_0 = k0; a.src1 = s1; a.k16_1 = k1; a.Wp = W + woff; woff += (size_t)(k0 + k1) * 16 * N; if (woff > (size_t)(1500000)) woff = 0; a.N = N;
        a.bias = bias; a.epi = epi; a.out = out; a.u = u; a.h = hh; a.nz = nz;
        launch_dec_gemm(a, s, nullptr);
    };
    auto step = [&](int st, int mask) {
        const float* prev = st == 0 ? go : outQ + ((size_t)(st - 1) * 400 + 320) * Bp;
        if (mask & 1) {
            gemm(prev, 5, ctx, 16, 256, DE_RELU, p1, nullptr);
            gemm(p1, 16, nullptr, 0, 128, DE_RELU, p2, nullptr);
            gemm(p2, 8, hatt, 16, 512, DE_GATES, rh, hatt);
            gemm(p2, 8, rh, 16, 256, DE_CAND, nullptr, hatt);
            gemm(hatt, 16, nullptr, 0, 256, DE_LINEAR, q, nullptr);
        }
        if (mask & 2) {
            DecAttnArgs at{q, keys, vals, v, ctx, al + (size_t)st * B * T_in, B, Bp, T_in, 256, 256};
            launch_dec_attention(at, s, nullptr);
        }
        if (mask & 4) {
            gemm(hatt, 16, ctx, 16, 256, DE_LINEAR, d0, nullptr);
            gemm(d0, 16, h1, 16, 512, DE_GATES, rh, h1);
            gemm(d0, 16, rh, 16, 256, DE_CAND, nullptr, h1);
            gemm(d0, 16, h2, 16, 512, DE_GATES, rh, h2);
            gemm(d0, 16, rh, 16, 256, DE_CAND, nullptr, h2);
            gemm(h2, 16, nullptr, 0, 400, DE_OUT, outQ + (size_t)st * 400 * Bp, nullptr);
        }
    };
    for (int mask : {7, 5, 2}) {
        hipGraph_t g; hipGraphExec_t ge;
        (void)hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal);
        for (int st = 0; st < S; ++st) step(st, mask);
        (void)hipStreamEndCapture(s, &g);
        (void)hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
        (void)hipGraphLaunch(ge, s); (void)hipStreamSynchronize(s);
        double best = 1e30;
        for (int rep = 0; rep < 5; ++rep) {
            auto t0 = std::chrono::high_resolution_clock::now();
            (void)hipGraphLaunch(ge, s); (void)hipStreamSynchronize(s);
            auto t1 = std::chrono::high_resolution_clock::now();
            best = std::min(best, std::chrono::duration<double, std::micro>(t1 - t0).count());
        }
        const int nk = ((mask & 1) ? 5 : 0) + ((mask & 2) ? 1 : 0) + ((mask & 4) ? 6 : 0);
        printf("B=%d mask=%d: %.1f us/step, %.2f us/kernel (%d kernels/step)\n", B, mask, best / S, best / S / nk, nk);
        (void)hipGraphExecDestroy(ge); (void)hipGraphDestroy(g);
    }
    return 0;
}
