// bigru_scan: the sequential half of tf.nn.bidirectional_dynamic_rnn(GRUCell(128), GRUCell(128),
// sequence_length) of the CBHG (model/modules.py:92-98).  The input halves of both cells'
// gate/candidate matmuls are hoisted into one conv_gemm ([B*T,128] @ [128,768] + biases); this
// kernel runs the recurrence, one 512-thread workgroup per (utterance, direction), with the
// recurrent weights (128x256 gates + 128x128 candidate = 192 KB) resident in VGPRs for all T
// steps (96 per thread) and the state in LDS.  Pure latency chain: 2 barriers per step.
//
// TF GRUCell: g = sigmoid([x,h] Wg + bg); r,u = split(g); c = tanh([x, r*h] Wc + bc);
// h' = u*h + (1-u)*c.  With sequence_length, steps t >= len emit 0 and keep the state; the
// backward cell runs over the first `len` steps reversed (A.6).
#include "kernels.h"
#include "dec_common.h"

namespace taco {

// gate non-linearities on the hardware exp2 / rcp (absolute error ~2e-7, saturate cleanly): they sit on the
// recurrence's dependency chain, the library expf + IEEE divide are several times the instructions
__device__ __forceinline__ float sigmoidf_(float x) {
    return __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(x * -1.4426950408889634f));
}
__device__ __forceinline__ float tanhf_(float x) { return fast_tanh(x); }
typedef float f32x2 __attribute__((ext_vector_type(2)));

__global__ __launch_bounds__(BIGRU_THREADS) void bigru_scan_kernel(
    const float* __restrict__ xp, const float* __restrict__ Pg, const float* __restrict__ Pc,
    const int32_t* __restrict__ lens, float* __restrict__ out, int T) {
    constexpr int H = BIGRU_H;           // 128
    constexpr int NT = BIGRU_THREADS;    // 512
    constexpr int KG = 2 * H * H / NT;   // 64 gate weights per thread (256 outputs x 128 k / 512)
    constexpr int KC = H * H / NT;       // 32 candidate weights per thread
    __shared__ __attribute__((aligned(16))) float hs[2][H];
    __shared__ __attribute__((aligned(16))) float rh[H];
    __shared__ __attribute__((aligned(16))) float us[H];

    const int b = blockIdx.x;
    const int dir = blockIdx.y;
    const int tid = threadIdx.x;
    int len = lens ? lens[b] : T;
    len = len < 0 ? 0 : (len > T ? T : len);

    // packed weights: Pg[dir][i][tid], Pc[dir][i][tid]
    float wg[KG], wc[KC];
#pragma unroll
    for (int i = 0; i < KG; ++i) wg[i] = Pg[((size_t)dir * KG + i) * NT + tid];
#pragma unroll
    for (int i = 0; i < KC; ++i) wc[i] = Pc[((size_t)dir * KC + i) * NT + tid];

    const int og = tid >> 1, half = tid & 1;   // gates: output og (0..255), k in [64*half, +64)
    const int oc = tid >> 2, q = tid & 3;      // cand : output oc (0..127), k in [32*q, +32)
    if (tid < H) { hs[0][tid] = 0.f; hs[1][tid] = 0.f; }

    const float* xrow = xp + (size_t)b * T * (6 * H) + dir * 3 * H;
    float* orow = out + (size_t)b * T * (2 * H) + dir * H;

    // zero the outputs of masked steps
    for (int i = tid; i < (T - len) * H; i += NT) {
        const int t = len + i / H;
        orow[(size_t)t * 2 * H + (i % H)] = 0.f;
    }

    auto tpos = [&](int s) { return dir == 0 ? s : len - 1 - s; };
    float xg = 0.f, xc = 0.f;
    if (len > 0) {
        const float* xr = xrow + (size_t)tpos(0) * 6 * H;
        if (half == 0) xg = xr[og];
        if (q == 0) xc = xr[2 * H + oc];
    }
    __syncthreads();

    int cur = 0;
    for (int s = 0; s < len; ++s) {
        // prefetch next step's input projections
        float xg_n = 0.f, xc_n = 0.f;
        if (s + 1 < len) {
            const float* xr = xrow + (size_t)tpos(s + 1) * 6 * H;
            if (half == 0) xg_n = xr[og];
            if (q == 0) xc_n = xr[2 * H + oc];
        }
        const float* h = hs[cur];
        // ---- gates
        {
            const float4* h4 = reinterpret_cast<const float4*>(h + 64 * half);
            f32x2 a01 = {0.f, 0.f}, a23 = {0.f, 0.f};     // packed fp32 FMAs: two MACs per instruction
#pragma unroll
            for (int i = 0; i < KG / 4; ++i) {
                const float4 hv = h4[i];
                a01 = __builtin_elementwise_fma(f32x2{wg[4 * i + 0], wg[4 * i + 1]}, f32x2{hv.x, hv.y}, a01);
                a23 = __builtin_elementwise_fma(f32x2{wg[4 * i + 2], wg[4 * i + 3]}, f32x2{hv.z, hv.w}, a23);
            }
            float acc = (a01.x + a01.y) + (a23.x + a23.y);
            acc += dpp_mov<0xB1>(acc);             // lane ^ 1 (DPP quad_perm, no LDS round trip)
            if (half == 0) {
                const float g = sigmoidf_(acc + xg);
                if (og < H) rh[og] = g * h[og];
                else us[og - H] = g;
            }
        }
        __syncthreads();
        // ---- candidate + state update
        {
            const float4* r4 = reinterpret_cast<const float4*>(rh + 32 * q);
            f32x2 a01 = {0.f, 0.f}, a23 = {0.f, 0.f};
#pragma unroll
            for (int i = 0; i < KC / 4; ++i) {
                const float4 rv = r4[i];
                a01 = __builtin_elementwise_fma(f32x2{wc[4 * i + 0], wc[4 * i + 1]}, f32x2{rv.x, rv.y}, a01);
                a23 = __builtin_elementwise_fma(f32x2{wc[4 * i + 2], wc[4 * i + 3]}, f32x2{rv.z, rv.w}, a23);
            }
            float acc = (a01.x + a01.y) + (a23.x + a23.y);
            acc += dpp_mov<0xB1>(acc);             // lane ^ 1
            acc += dpp_mov<0x4E>(acc);             // lane ^ 2
            if (q == 0) {
                const float c = tanhf_(acc + xc);
                const float u = us[oc];
                const float hn = u * h[oc] + (1.f - u) * c;
                hs[cur ^ 1][oc] = hn;
                orow[(size_t)tpos(s) * 2 * H + oc] = hn;
            }
        }
        __syncthreads();
        cur ^= 1;
        xg = xg_n;
        xc = xc_n;
    }
}

void launch_bigru_scan(const float* xp, const float* Pg, const float* Pc, const int32_t* lens, float* out,
                       int B, int T, hipStream_t s, Timing* tm) {
    if (tm) tm->begin(KK_BIGRU, s);
    hipLaunchKernelGGL(bigru_scan_kernel, dim3(B, 2), dim3(BIGRU_THREADS), 0, s, xp, Pg, Pc, lens, out, T);
    if (tm) tm->end(KK_BIGRU, s);
}

}  // namespace taco
