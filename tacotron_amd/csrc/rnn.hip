// bigru_scan: the sequential half of tf.nn.bidirectional_dynamic_rnn(GRUCell(128), GRUCell(128),
// sequence_length) of the CBHG (model/modules.py:92-98).  The input halves of both cells'
// gate/candidate matmuls are hoisted into one conv_gemm ([B*T,128] @ [128,768] + biases); this
// kernel runs the recurrence, one 512-thread workgroup per (utterance, direction), with the
// recurrent weights (128x256 gates + 128x128 candidate = 192 KB) resident in VGPRs for all T
// steps (96 per thread) and the state in LDS.  Pure latency chain: 2 barriers per step; per step a lane reads 16 + 16
// state values from LDS and meets the other k-slices of its outputs through DPP.
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

    // gates: lane = (k-slice ks of 16, group of 4 outputs); candidate: (k-slice of 16, group of 2 outputs).
    // A lane reads only its 16 state values (4 ds_read_b128): the LDS return path, not the FMAs, bounds a step when
    // every lane pulls 64 of them (8 waves x 16 x 1 KB per step); the 8 k-slices of an output meet through three DPP
    // adds (lane^1, lane^2, half-row mirror), and lanes ks < 4 (ks < 2) finish one output each.
    const int ks = tid & 7, grp = tid >> 3;
    const int og = 4 * grp + (ks & 3);         // gate output finished by this lane when ks < 4 (0..255)
    const int oc = 2 * grp + (ks & 1);         // candidate output finished by this lane when ks < 2 (0..127)
    if (tid < H) { hs[0][tid] = 0.f; hs[1][tid] = 0.f; }

    const float* xrow = xp + (size_t)b * T * (6 * H) + dir * 3 * H;
    float* orow = out + (size_t)b * T * (2 * H) + dir * H;

    // zero the outputs of masked steps
    for (int i = tid; i < (T - len) * H; i += NT) {
        const int t = len + i / H;
        orow[(size_t)t * 2 * H + (i % H)] = 0.f;
    }

    auto tpos = [&](int s) { return dir == 0 ? s : len - 1 - s; };
    // The input projections are read once, straight from HBM: a load issued one step ahead is still in flight when
    // its value is needed (and vmcnt retires in order, behind the output stores), so they are fetched in chunks of
    // U steps, a whole chunk (>= U steps of arithmetic) before their first use.
    constexpr int U = 8;
    float xg_c[U], xc_c[U], xg_n[U], xc_n[U];
#define BIGRU_LOAD_CHUNK(S0, G, C)                                                  \
    _Pragma("unroll") for (int u = 0; u < U; ++u) {                                 \
        const int sl = min((S0) + u, len - 1);    /* clamped: past the end unused */ \
        const float* xr = xrow + (size_t)tpos(sl) * 6 * H;                          \
        G[u] = xr[og];                                                              \
        C[u] = xr[2 * H + oc];                                                      \
    }
    if (len > 0) { BIGRU_LOAD_CHUNK(0, xg_c, xc_c) }
    __syncthreads();

    auto sum8 = [](float x) {
        x += dpp_mov<0xB1>(x);                 // lane ^ 1
        x += dpp_mov<0x4E>(x);                 // lane ^ 2
        x += dpp_mov<0x141>(x);                // row_half_mirror: the other quad of the 8
        return x;
    };

    for (int s0 = 0; s0 < len; s0 += U) {
        if (s0 + U < len) { BIGRU_LOAD_CHUNK(s0 + U, xg_n, xc_n) }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int s = s0 + u;
            if (s >= len) break;
            const int cur = s & 1;
            const float* h = hs[cur];
            // ---- gates
            {
                const float4* h4 = reinterpret_cast<const float4*>(h + 16 * ks);
                const float h_own = h[og & (H - 1)];       // for r*h: read up front, not behind the sigmoid
                const float4 v0 = h4[0], v1 = h4[1], v2 = h4[2], v3 = h4[3];
                const f32x2 hp[8] = {{v0.x, v0.y}, {v0.z, v0.w}, {v1.x, v1.y}, {v1.z, v1.w},
                                     {v2.x, v2.y}, {v2.z, v2.w}, {v3.x, v3.y}, {v3.z, v3.w}};
                float sj[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    f32x2 a = {0.f, 0.f};
#pragma unroll
                    for (int i = 0; i < 8; ++i)
                        a = __builtin_elementwise_fma(f32x2{wg[16 * j + 2 * i], wg[16 * j + 2 * i + 1]}, hp[i], a);
                    sj[j] = sum8(a.x + a.y);
                }
                if (ks < 4) {
                    const float acc = (ks & 2) ? ((ks & 1) ? sj[3] : sj[2]) : ((ks & 1) ? sj[1] : sj[0]);
                    const float g = sigmoidf_(acc + xg_c[u]);
                    if (og < H) rh[og] = g * h_own;
                    else us[og - H] = g;
                }
            }
            __syncthreads();
            // ---- candidate + state update
            {
                const float4* r4 = reinterpret_cast<const float4*>(rh + 16 * ks);
                const float uu = us[oc], h_own = h[oc];
                const float4 v0 = r4[0], v1 = r4[1], v2 = r4[2], v3 = r4[3];
                const f32x2 rp[8] = {{v0.x, v0.y}, {v0.z, v0.w}, {v1.x, v1.y}, {v1.z, v1.w},
                                     {v2.x, v2.y}, {v2.z, v2.w}, {v3.x, v3.y}, {v3.z, v3.w}};
                float sj[2];
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    f32x2 a = {0.f, 0.f};
#pragma unroll
                    for (int i = 0; i < 8; ++i)
                        a = __builtin_elementwise_fma(f32x2{wc[16 * j + 2 * i], wc[16 * j + 2 * i + 1]}, rp[i], a);
                    sj[j] = sum8(a.x + a.y);
                }
                if (ks < 2) {
                    const float c = tanhf_((ks & 1 ? sj[1] : sj[0]) + xc_c[u]);
                    const float hn = uu * h_own + (1.f - uu) * c;
                    hs[cur ^ 1][oc] = hn;
                    orow[(size_t)tpos(s) * 2 * H + oc] = hn;
                }
            }
            __syncthreads();
        }
#pragma unroll
        for (int u = 0; u < U; ++u) { xg_c[u] = xg_n[u]; xc_c[u] = xc_n[u]; }
    }
#undef BIGRU_LOAD_CHUNK
}

// The same recurrence for any number of cells G (hparams *_gru_num_cells other than the reference's 128): one
// 256-thread workgroup per (utterance, direction); the recurrent weights do not fit a register image in general, so
// they are read from L2 every step, k-major ([G][2G] gates, [G][G] candidate: coalesced over the outputs); state,
// r*h and u in LDS. Functional reach, not a tuned path: ~G/8 us per step.
constexpr int BIGRU_GEN_THREADS = 256;
__global__ __launch_bounds__(BIGRU_GEN_THREADS) void bigru_scan_generic_kernel(
    const float* __restrict__ xp, const float* __restrict__ Wg, const float* __restrict__ Wc,
    const int32_t* __restrict__ lens, float* __restrict__ out, int T, int G) {
    extern __shared__ __attribute__((aligned(16))) float gsm[];
    float* hs = gsm;                // [2][G]
    float* rh = gsm + 2 * G;        // [G]
    float* us = rh + G;             // [G]
    const int b = blockIdx.x, dir = blockIdx.y, tid = threadIdx.x;
    int len = lens ? lens[b] : T;
    len = len < 0 ? 0 : (len > T ? T : len);
    const float* wg = Wg + (size_t)dir * G * 2 * G;
    const float* wc = Wc + (size_t)dir * G * G;
    const float* xrow = xp + (size_t)b * T * (6 * G) + dir * 3 * G;
    float* orow = out + (size_t)b * T * (2 * G) + dir * G;
    for (int i = tid; i < 2 * G; i += BIGRU_GEN_THREADS) hs[i] = 0.f;
    for (int i = tid; i < (T - len) * G; i += BIGRU_GEN_THREADS) orow[(size_t)(len + i / G) * 2 * G + (i % G)] = 0.f;
    __syncthreads();
    for (int s = 0; s < len; ++s) {
        const int t = dir == 0 ? s : len - 1 - s;
        const float* h = hs + (s & 1) * G;
        float* hn_s = hs + ((s & 1) ^ 1) * G;
        const float* xr = xrow + (size_t)t * 6 * G;
        for (int o = tid; o < 2 * G; o += BIGRU_GEN_THREADS) {
            float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
            for (int k = 0; k < G; k += 4) {
                a0 = fmaf(h[k], wg[(size_t)k * 2 * G + o], a0);
                a1 = fmaf(h[k + 1], wg[(size_t)(k + 1) * 2 * G + o], a1);
                a2 = fmaf(h[k + 2], wg[(size_t)(k + 2) * 2 * G + o], a2);
                a3 = fmaf(h[k + 3], wg[(size_t)(k + 3) * 2 * G + o], a3);
            }
            const float g = sigmoidf_(((a0 + a1) + (a2 + a3)) + xr[o]);
            if (o < G) rh[o] = g * h[o];
            else us[o - G] = g;
        }
        __syncthreads();
        for (int o = tid; o < G; o += BIGRU_GEN_THREADS) {
            float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
            for (int k = 0; k < G; k += 4) {
                a0 = fmaf(rh[k], wc[(size_t)k * G + o], a0);
                a1 = fmaf(rh[k + 1], wc[(size_t)(k + 1) * G + o], a1);
                a2 = fmaf(rh[k + 2], wc[(size_t)(k + 2) * G + o], a2);
                a3 = fmaf(rh[k + 3], wc[(size_t)(k + 3) * G + o], a3);
            }
            const float c = tanhf_(((a0 + a1) + (a2 + a3)) + xr[2 * G + o]);
            const float uu = us[o];
            const float hn = uu * h[o] + (1.f - uu) * c;
            hn_s[o] = hn;
            orow[(size_t)t * 2 * G + o] = hn;
        }
        __syncthreads();
    }
}

// The same recurrence as a THROUGHPUT variant (taco_set_bigru_mode 1; 128 cells): one 512-thread workgroup scans one
// direction of SIXTEEN utterances in lockstep, the recurrent products on v_mfma_f32_16x16x4_f32 (exact f32, a fixed k
// order that does not depend on the batch). The weights are the A operand and stay in registers for all T steps (wave
// w owns the 16 cells 16 w .. 16 w + 15: its reset, update and candidate tiles, 3 x 32 = 96 registers per lane); the
// state is the B operand. A lane holds its C/D fragment (cells n0 .. n0 + 3 of utterance `col`) as one float4, and
// that float4, stored as quad (n0 / 4, col) of a [32 quads][16 utterances] LDS image, is at once what an MFMA group
// reads back as B: lane l takes quad (4 g + l / 16, l % 16) and feeds its component m to MFMA (g, m), which therefore
// sums over k = 16 g + 4 (l / 16) + m - the weights are packed to that order (api.hip). Per step and wave: 8 + 8
// ds_read_b128, 96 MFMAs (6144 MFMA cycles a SIMD: ~2.6 us a step, against 0.64 us for the one-utterance kernel), 2
// barriers; but 4 workgroups scan a batch of 32 instead of 64, ~12 CU-ms instead of ~49 per post-net scan: the
// variant for several batches in flight, where CU time, not latency, is what a scan costs. Same TF gate order (reset
// gate applied to the state BEFORE the candidate matmul, modules.py:92-98 / A.5), same masking (A.6): a column past
// its length keeps its state and emits nothing (its rows were zeroed up front); the backward direction walks each
// column's own first `len` steps reversed.
constexpr int BIGRU_MFMA_THREADS = 512;
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(BIGRU_MFMA_THREADS) void bigru_scan_mfma_kernel(
    const float* __restrict__ xp, const float* __restrict__ Pm, const int32_t* __restrict__ lens,
    float* __restrict__ out, int B, int T) {
    constexpr int H = BIGRU_H;
    static_assert(H == 128 && BIGRU_MFMA_THREADS == 512, "eight waves of 16 cells each");
    __shared__ __attribute__((aligned(16))) f32x4 hq[2][32 * 16];     // state, [quad of cells][utterance]
    __shared__ __attribute__((aligned(16))) f32x4 rhq[32 * 16];       // r * h
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int dir = blockIdx.y;
    const int col = lane & 15, rq = lane >> 4;
    const int b = blockIdx.x * 16 + col;
    const bool has_b = b < B;
    const int bc = has_b ? b : B - 1;                                // (addresses only: a padding column never stores)
    auto clamp_len = [&](int i) { const int l = lens ? lens[i] : T; return l < 0 ? 0 : (l > T ? T : l); };
    const int len = has_b ? clamp_len(b) : 0;
    int maxlen = 0;
    for (int c = 0; c < 16; ++c) {
        const int bb = blockIdx.x * 16 + c;
        if (bb < B) maxlen = max(maxlen, clamp_len(bb));
    }
    // zero the outputs of masked steps (this direction's half of the rows past each column's length)
    for (int c = 0; c < 16; ++c) {
        const int bb = blockIdx.x * 16 + c;
        if (bb >= B) break;
        const int lc = clamp_len(bb);
        float* orow = out + ((size_t)bb * T + lc) * (2 * H) + dir * H;
        for (int i = tid; i < (T - lc) * (H / 4); i += BIGRU_MFMA_THREADS)
            *reinterpret_cast<f32x4*>(orow + (size_t)(i / (H / 4)) * 2 * H + 4 * (i % (H / 4))) = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    float wa[96];
#pragma unroll
    for (int i = 0; i < 96; ++i) wa[i] = Pm[(((size_t)dir * 8 + wave) * 96 + i) * 64 + lane];
    const int n0 = 16 * wave + 4 * rq;                               // the lane's four cells
    const int myq = (4 * wave + rq) * 16 + col;                      // ... as a quad of the LDS images
    f32x4 h_own = {0.f, 0.f, 0.f, 0.f};
    hq[0][myq] = h_own;
    const float* xb = xp + (size_t)bc * T * (6 * H) + dir * 3 * H + n0;
    float* ob = out + (size_t)bc * T * (2 * H) + dir * H + n0;
    auto tpos = [&](int s) { const int sc = min(s, len - 1); return len > 0 ? (dir == 0 ? sc : len - 1 - sc) : 0; };
    auto loadx = [&](int s, f32x4& xr, f32x4& xu, f32x4& xc) {
        const float* p = xb + (size_t)tpos(s) * (6 * H);
        xr = *reinterpret_cast<const f32x4*>(p);
        xu = *reinterpret_cast<const f32x4*>(p + H);
        xc = *reinterpret_cast<const f32x4*>(p + 2 * H);
    };
    f32x4 xr, xu, xc, xr_n, xu_n, xc_n;
    loadx(0, xr, xu, xc);
    __syncthreads();
    for (int s = 0; s < maxlen; ++s) {
        const int cur = s & 1;
        loadx(s + 1, xr_n, xu_n, xc_n);                              // (one step ahead: a step is ~2.6 us)
        f32x4 ar = {0.f, 0.f, 0.f, 0.f}, au = ar;
#pragma unroll
        for (int g = 0; g < 8; ++g) {
            const f32x4 hb = hq[cur][64 * g + lane];
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                ar = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[4 * g + m], hb[m], ar, 0, 0, 0);
                au = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[32 + 4 * g + m], hb[m], au, 0, 0, 0);
            }
        }
        f32x4 r, u;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            r[j] = sigmoidf_(ar[j] + xr[j]);
            u[j] = sigmoidf_(au[j] + xu[j]);
        }
        rhq[myq] = r * h_own;
        __syncthreads();
        f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = a0;                    // (two chains: dependent MFMAs do not issue back to back)
#pragma unroll
        for (int g = 0; g < 8; g += 2) {
            const f32x4 r0 = rhq[64 * g + lane], r1 = rhq[64 * (g + 1) + lane];
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[64 + 4 * g + m], r0[m], a0, 0, 0, 0);
                a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[64 + 4 * (g + 1) + m], r1[m], a1, 0, 0, 0);
            }
        }
        const f32x4 ac = a0 + a1;
        if (s < len) {
            f32x4 hn;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float c = tanhf_(ac[j] + xc[j]);
                hn[j] = u[j] * h_own[j] + (1.f - u[j]) * c;
            }
            h_own = hn;
            *reinterpret_cast<f32x4*>(ob + (size_t)tpos(s) * (2 * H)) = hn;
        }
        hq[cur ^ 1][myq] = h_own;
        xr = xr_n; xu = xu_n; xc = xc_n;
        __syncthreads();
    }
}

void launch_bigru_scan_mfma(const float* xp, const float* Pm, const int32_t* lens, float* out, int B, int T,
                            hipStream_t s, Timing* tm) {
    if (tm) tm->begin(KK_BIGRU, s);
    hipLaunchKernelGGL(bigru_scan_mfma_kernel, dim3((B + 15) / 16, 2), dim3(BIGRU_MFMA_THREADS), 0, s, xp, Pm, lens, out,
                       B, T);
    if (tm) tm->end(KK_BIGRU, s);
}

// A third form (taco_set_bigru_mode 2; 128 cells): FOUR utterances per 512-thread workgroup on
// v_mfma_f32_4x4x1_16B_f32 - sixteen independent 4x4 outer products an instruction: a block's four rows are four cells,
// its four columns the four utterances, K = 1. One MFMA covers 64 rows (lane l supplies the A operand of row l) times 4
// utterances for one k, in 8 cycles, so a step's 128 x 384 products per utterance cost 96 MFMAs a wave, 2 waves a SIMD
// = 1536 MFMA cycles: the time of the one-utterance kernel's step (whose packed FMAs are what bounds it), for four
// utterances. Wave w owns cells 16 w .. 16 w + 15 and splits K so that its 64 rows are always full: in the gates phase
// [reset | update | reset | update] x 16 cells, the lower 32 rows over k < 64, the upper over k >= 64 (the B operand is
// per block, so the upper blocks are fed h[64 + k]); in the candidate phase the 16 cells over four quarters of k. The
// partial sums and the two gates meet through lane exchanges (v_permlane32_swap, v_permlane16_swap: no LDS round trip)
// in a fixed order. A lane's C/D fragment is four consecutive cells of one utterance (a float4); stored as quad
// (cell / 4, utterance) of a [32 quads][4 utterances] LDS image it is what the next phase reads back as the B operands
// of four consecutive k. Same gate order, masking and reversed backward walk as the other forms.
constexpr int BIGRU_M4_THREADS = 512;
__global__ __launch_bounds__(BIGRU_M4_THREADS) void bigru_scan_mfma4_kernel(
    const float* __restrict__ xp, const float* __restrict__ Pm4, const int32_t* __restrict__ lens,
    float* __restrict__ out, int B, int T) {
    constexpr int H = BIGRU_H;
    static_assert(H == 128 && BIGRU_M4_THREADS == 512, "eight waves of 16 cells each");
    __shared__ __attribute__((aligned(16))) f32x4 hq[2][32 * 4];      // state, [quad of cells][utterance]
    __shared__ __attribute__((aligned(16))) f32x4 rhq[32 * 4];        // r * h
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int dir = blockIdx.y;
    const int q = lane >> 4, cq = (lane & 15) >> 2, col = lane & 3;   // row group, cell quad, utterance
    const int b = blockIdx.x * 4 + col;
    const bool has_b = b < B;
    const int bc = has_b ? b : B - 1;
    auto clamp_len = [&](int i) { const int l = lens ? lens[i] : T; return l < 0 ? 0 : (l > T ? T : l); };
    const int len = has_b ? clamp_len(b) : 0;
    int maxlen = 0;
    for (int c = 0; c < 4; ++c) {
        const int bb = blockIdx.x * 4 + c;
        if (bb < B) maxlen = max(maxlen, clamp_len(bb));
    }
    for (int c = 0; c < 4; ++c) {                                     // zero the outputs of masked steps
        const int bb = blockIdx.x * 4 + c;
        if (bb >= B) break;
        const int lc = clamp_len(bb);
        float* orow = out + ((size_t)bb * T + lc) * (2 * H) + dir * H;
        for (int i = tid; i < (T - lc) * (H / 4); i += BIGRU_M4_THREADS)
            *reinterpret_cast<f32x4*>(orow + (size_t)(i / (H / 4)) * 2 * H + 4 * (i % (H / 4))) = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    float wa[96];
#pragma unroll
    for (int i = 0; i < 96; ++i) wa[i] = Pm4[(((size_t)dir * 8 + wave) * 96 + i) * 64 + lane];
    const int cell0 = 16 * wave + 4 * cq;                             // the lane's four cells
    const int myq = (4 * wave + cq) * 4 + col;                        // ... as a quad of the LDS images
    f32x4 h_own = {0.f, 0.f, 0.f, 0.f};
    if (q == 0) hq[0][myq] = h_own;
    const float* xb = xp + (size_t)bc * T * (6 * H) + dir * 3 * H + cell0;
    float* ob = out + (size_t)bc * T * (2 * H) + dir * H + cell0;
    auto tpos = [&](int s) { const int sc = min(s, len - 1); return len > 0 ? (dir == 0 ? sc : len - 1 - sc) : 0; };
    // the value of lane l ^ 32 / l ^ 16: one v_permlane32_swap / v_permlane16_swap (no LDS round trip)
    auto x32 = [&](float x) {
        const auto r = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, x), __builtin_bit_cast(unsigned, x),
                                                        false, false);
        return __builtin_bit_cast(float, (q & 2) ? r[0] : r[1]);
    };
    auto x16 = [&](float x) {
        const auto r = __builtin_amdgcn_permlane16_swap(__builtin_bit_cast(unsigned, x), __builtin_bit_cast(unsigned, x),
                                                        false, false);
        return __builtin_bit_cast(float, (q & 1) ? r[0] : r[1]);
    };
    // the lane's input projections: its gate (reset in the even row groups, update in the odd) and the candidate
    constexpr int U = 4;
    f32x4 xg_c[U], xc_c[U], xg_n[U], xc_n[U];
    auto load_chunk = [&](int s0, f32x4 (&G)[U], f32x4 (&C)[U]) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const float* p = xb + (size_t)tpos(s0 + u) * (6 * H);
            G[u] = *reinterpret_cast<const f32x4*>(p + (q & 1) * H);
            C[u] = *reinterpret_cast<const f32x4*>(p + 2 * H);
        }
    };
    load_chunk(0, xg_c, xc_c);
    __syncthreads();
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    for (int s0 = 0; s0 < maxlen; s0 += U) {
        load_chunk(s0 + U, xg_n, xc_n);
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int s = s0 + u;
            if (s >= maxlen) break;
            const int cur = s & 1;
            // ---- gates: rows [reset | update] x 16 cells over k < 64 (lanes < 32) / k >= 64 (lanes >= 32)
            f32x4 hb[16];
#pragma unroll
            for (int g = 0; g < 16; ++g) hb[g] = hq[cur][4 * (g + 16 * (q >> 1)) + col];
            f32x4 acc[4] = {zero4, zero4, zero4, zero4};
#pragma unroll
            for (int g = 0; g < 16; ++g) {
#pragma unroll
                for (int m = 0; m < 4; ++m)
                    acc[m] = __builtin_amdgcn_mfma_f32_4x4x1f32(wa[4 * g + m], hb[g][m], acc[m], 0, 0, 0);
            }
            const f32x4 ah = (acc[0] + acc[1]) + (acc[2] + acc[3]);
            f32x4 gate, other;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float o = x32(ah[j]);
                const float t = (q & 2) ? o + ah[j] : ah[j] + o;       // (k < 64) + (k >= 64), in every lane
                gate[j] = sigmoidf_(t + xg_c[u][j]);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) other[j] = x16(gate[j]);
            const f32x4 r = (q & 1) ? other : gate, uu = (q & 1) ? gate : other;
            if (q == 0) rhq[myq] = r * h_own;
            __syncthreads();
            // ---- candidate: rows 16 cells x four quarters of k (row group q: k in [32 q, 32 q + 32))
            f32x4 rb[8];
#pragma unroll
            for (int g = 0; g < 8; ++g) rb[g] = rhq[4 * (g + 8 * q) + col];
            f32x4 a2[4] = {zero4, zero4, zero4, zero4};
#pragma unroll
            for (int g = 0; g < 8; ++g) {
#pragma unroll
                for (int m = 0; m < 4; ++m)
                    a2[m] = __builtin_amdgcn_mfma_f32_4x4x1f32(wa[64 + 4 * g + m], rb[g][m], a2[m], 0, 0, 0);
            }
            const f32x4 ap = (a2[0] + a2[1]) + (a2[2] + a2[3]);
            f32x4 ac;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float o1 = x16(ap[j]);
                const float s01 = (q & 1) ? o1 + ap[j] : ap[j] + o1;   // quarters (0 + 1) or (2 + 3)
                const float o2 = x32(s01);
                ac[j] = (q & 2) ? o2 + s01 : s01 + o2;                 // (0 + 1) + (2 + 3)
            }
            if (s < len) {
                f32x4 hn;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float c = tanhf_(ac[j] + xc_c[u][j]);
                    hn[j] = uu[j] * h_own[j] + (1.f - uu[j]) * c;
                }
                h_own = hn;
                if (q == 1) *reinterpret_cast<f32x4*>(ob + (size_t)tpos(s) * (2 * H)) = hn;
            }
            if (q == 0) hq[cur ^ 1][myq] = h_own;
            __syncthreads();
        }
#pragma unroll
        for (int u = 0; u < U; ++u) { xg_c[u] = xg_n[u]; xc_c[u] = xc_n[u]; }
    }
}

void launch_bigru_scan_mfma4(const float* xp, const float* Pm4, const int32_t* lens, float* out, int B, int T,
                             hipStream_t s, Timing* tm) {
    if (tm) tm->begin(KK_BIGRU, s);
    hipLaunchKernelGGL(bigru_scan_mfma4_kernel, dim3((B + 3) / 4, 2), dim3(BIGRU_M4_THREADS), 0, s, xp, Pm4, lens, out,
                       B, T);
    if (tm) tm->end(KK_BIGRU, s);
}

void launch_bigru_scan(const float* xp, const float* Pg, const float* Pc, const int32_t* lens, float* out,
                       int B, int T, int G, hipStream_t s, Timing* tm) {
    if (tm) tm->begin(KK_BIGRU, s);
    if (G == BIGRU_H)
        hipLaunchKernelGGL(bigru_scan_kernel, dim3(B, 2), dim3(BIGRU_THREADS), 0, s, xp, Pg, Pc, lens, out, T);
    else
        hipLaunchKernelGGL(bigru_scan_generic_kernel, dim3(B, 2), dim3(BIGRU_GEN_THREADS), (size_t)4 * G * sizeof(float),
                           s, xp, Pg, Pc, lens, out, T, G);
    if (tm) tm->end(KK_BIGRU, s);
}

}  // namespace taco
