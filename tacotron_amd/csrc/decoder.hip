// Decoder-step kernels: the body of dynamic_decode(BasicDecoder(output_cell, TestingHelper))
// (model/decoder.py:18-40, model/rnn_wrappers.py:24-26,51-53, model/helpers.py:66-78;
// step maths in SURVEY.md A.7).
//
// Data layout ("Q layout"): every per-step vector of the batch (previous frame, context, GRU
// states, prenet activations, query, ...) is stored batch-minor as float4 quads,
//     act[(k/4) * Bpad + b] = {x_b[k], x_b[k+1], x_b[k+2], x_b[k+3]},   Bpad = 16*ceil(B/16),
// which is exactly the C/D fragment of v_mfma_f32_16x16x4_f32 when the weights are the A
// operand and the activations the B operand (lane l: column b = l&15, rows n = 4*(l>>4)+reg),
// and - with the K order inside a 16-deep step permuted so that MFMA i consumes
// k = 4*(l>>4) + i - also its B operand.  Every global access is a coalesced dwordx4; the
// weights are re-packed once (taco_create) as Wp[N/16][K/16][64 lanes][4].
//
// dec_gemm: one 16(n) x 16(b) tile per workgroup, K split over the 8 waves (fixed chunks,
// fixed-order LDS reduction => results do not depend on batch size or placement), fused
// epilogues for the prenet (ReLU [+dropout replay]), GRU gates (sigmoid, r*h, u), GRU
// candidate (tanh, state update, residual) and the r-frame output projection (+stop test).
#include "kernels.h"
#include "dec_common.h"

namespace taco {

constexpr int DG_WAVES = 8;       // K is split over 8 waves (fixed chunks => fixed summation order)
constexpr int DG_THREADS = 64 * DG_WAVES;
constexpr int DG_DEPTH = 4;       // k16-steps whose loads are issued together per wave

__global__ __launch_bounds__(DG_THREADS) void dec_gemm_kernel(DecGemmArgs a) {
    __shared__ f32x4 red[DG_WAVES][64];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int ntile = blockIdx.x;
    const int b0 = blockIdx.y * 16;
    const int nk16 = a.k16_0 + a.k16_1;
    const int chunk = (nk16 + DG_WAVES - 1) / DG_WAVES;
    const int ks_begin = wave * chunk;
    const int ks_end = min(nk16, ks_begin + chunk);

    const float4* Wp = reinterpret_cast<const float4*>(a.Wp) + ((size_t)ntile * nk16) * 64 + lane;
    const float4* s0 = reinterpret_cast<const float4*>(a.src0);
    const float4* s1 = reinterpret_cast<const float4*>(a.src1);
    const int lb = b0 + (lane & 15);
    const int lq = lane >> 4;
    // epilogue operands of wave 0 are requested up front so their latency hides under the K loop
    const int n = ntile * 16 + 4 * lq;                 // first of this lane's 4 output rows
    const size_t e = (size_t)(n >> 2) * a.Bpad + lb;   // float4 index in a Q-layout buffer of dimension N
    f32x4 pbias = {0.f, 0.f, 0.f, 0.f}, ph = pbias, pu = pbias, pres = pbias;
    if (wave == 0) {
        if (a.bias) pbias = *reinterpret_cast<const f32x4*>(a.bias + n);
        if (a.epi == DE_GATES) {
            if (n < (a.N >> 1)) ph = reinterpret_cast<const f32x4*>(a.h)[e];
        } else if (a.epi == DE_CAND) {
            ph = reinterpret_cast<const f32x4*>(a.h)[e];
            pu = reinterpret_cast<const f32x4*>(a.u)[e];
            if (a.res_out) pres = reinterpret_cast<const f32x4*>(a.res)[e];
        }
    }
    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
    for (int base = ks_begin; base < ks_end; base += DG_DEPTH) {
        float4 w[DG_DEPTH], x[DG_DEPTH];
        // all loads of this group are issued before the first MFMA waits on one
#pragma unroll
        for (int i = 0; i < DG_DEPTH; ++i) {
            const int ks = base + i;
            if (ks < ks_end) {
                w[i] = Wp[(size_t)ks * 64];
                x[i] = ks < a.k16_0 ? s0[(size_t)(4 * ks + lq) * a.Bpad + lb]
                                    : s1[(size_t)(4 * (ks - a.k16_0) + lq) * a.Bpad + lb];
            } else {
                w[i] = make_float4(0.f, 0.f, 0.f, 0.f);
                x[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
#pragma unroll
        for (int i = 0; i < DG_DEPTH; ++i) {
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(w[i].x, x[i].x, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(w[i].y, x[i].y, acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(w[i].z, x[i].z, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(w[i].w, x[i].w, acc1, 0, 0, 0);
        }
    }
    red[wave][lane] = acc0 + acc1;
    __syncthreads();
    if (wave != 0) return;
    f32x4 v = red[0][lane];
#pragma unroll
    for (int w = 1; w < DG_WAVES; ++w) v += red[w][lane];

    v += pbias;
    switch (a.epi) {
        case DE_LINEAR:
            reinterpret_cast<f32x4*>(a.out)[e] = v;
            break;
        case DE_RELU: {
            v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
            if (a.keep && lb < a.B) {
                const float4 k = *reinterpret_cast<const float4*>(a.keep + (size_t)lb * a.keep_ld + n);
                v.x *= 2.f * k.x; v.y *= 2.f * k.y; v.z *= 2.f * k.z; v.w *= 2.f * k.w;
            }
            reinterpret_cast<f32x4*>(a.out)[e] = v;
            break;
        }
        case DE_GATES: {
            f32x4 g;
            g.x = dsigmoid(v.x); g.y = dsigmoid(v.y); g.z = dsigmoid(v.z); g.w = dsigmoid(v.w);
            const int H = a.N >> 1;
            if (n < H) {
                reinterpret_cast<f32x4*>(a.out)[e] = g * ph;                         // r * h
            } else {
                reinterpret_cast<f32x4*>(a.u)[e - (size_t)(H >> 2) * a.Bpad] = g;   // u
            }
            break;
        }
        case DE_CAND: {
            f32x4 c;
            c.x = dtanh(v.x); c.y = dtanh(v.y); c.z = dtanh(v.z); c.w = dtanh(v.w);
            const f32x4 hn = pu * ph + (1.f - pu) * c;
            reinterpret_cast<f32x4*>(a.h)[e] = hn;
            if (a.res_out) reinterpret_cast<f32x4*>(a.res_out)[e] = hn + pres;
            break;
        }
        case DE_OUT: {
            reinterpret_cast<f32x4*>(a.out)[e] = v;   // the stop test reads these back once, in dec_nonzero_kernel
            break;
        }
    }
}

void launch_dec_gemm(const DecGemmArgs& a, hipStream_t s, Timing* tm) {
    if (tm) tm->begin(KK_DEC_GEMM, s);
    hipLaunchKernelGGL(dec_gemm_kernel, dim3(a.N / 16, a.Bpad / 16), dim3(DG_THREADS), 0, s, a);
    if (tm) tm->end(KK_DEC_GEMM, s);
}

// ---------------------------------------------------------------------------------------
// Bahdanau attention for one step (TF _bahdanau_score + softmax + context; decoder.py:20):
//   e_t = sum_d v[d] * tanh(keys[t,d] + q[d])  over ALL T_in (no memory_sequence_length, A.8)
//   a = softmax(e);  ctx = sum_t a_t * values[t,:]
// One workgroup per utterance; a wave owns time steps t = wave, wave+NW, ...; each lane owns
// d-quads lane, lane+64, ...; wave butterflies for the score sums, block reductions in LDS.
constexpr int ATTN_THREADS = 1024;
constexpr int ATTN_WAVES = ATTN_THREADS / 64;
constexpr int ATTN_DEPTH = 5;     // time steps per wave whose loads are in flight together

__global__ __launch_bounds__(ATTN_THREADS) void dec_attention_kernel(DecAttnArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* e = smem;                                                  // [T_in]
    f32x4* cred = reinterpret_cast<f32x4*>(smem + ((a.T_in + 3) & ~3)); // [ATTN_WAVES][Dv/4]
    const int b = blockIdx.x;
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int AQ = a.A >> 2, VQ = a.Dv >> 2;

    // ---- scores: wave owns t = wave + 16*j; lane owns d-quads lane, lane+64, ...
    const float4* q4 = reinterpret_cast<const float4*>(a.q);
    const float4* v4 = reinterpret_cast<const float4*>(a.v);
    const float4* kb = reinterpret_cast<const float4*>(a.keys) + (size_t)b * a.T_in * AQ;
    for (int tb = wave; tb < a.T_in; tb += ATTN_WAVES * ATTN_DEPTH) {
        float sc[ATTN_DEPTH];
#pragma unroll
        for (int j = 0; j < ATTN_DEPTH; ++j) sc[j] = 0.f;
        for (int dq = lane; dq < AQ; dq += 64) {
            float4 k[ATTN_DEPTH];
#pragma unroll
            for (int j = 0; j < ATTN_DEPTH; ++j) {
                const int t = tb + ATTN_WAVES * j;
                k[j] = t < a.T_in ? kb[(size_t)t * AQ + dq] : make_float4(0.f, 0.f, 0.f, 0.f);
            }
            const float4 q = q4[(size_t)dq * a.Bpad + b];
            const float4 vv = v4[dq];
#pragma unroll
            for (int j = 0; j < ATTN_DEPTH; ++j)
                sc[j] += vv.x * fast_tanh(k[j].x + q.x) + vv.y * fast_tanh(k[j].y + q.y) +
                         vv.z * fast_tanh(k[j].z + q.z) + vv.w * fast_tanh(k[j].w + q.w);
        }
#pragma unroll
        for (int j = 0; j < ATTN_DEPTH; ++j) {
            const float s = wave_sum(sc[j]);
            const int t = tb + ATTN_WAVES * j;
            if (lane == 0 && t < a.T_in) e[t] = s;
        }
    }
    __syncthreads();
    // ---- softmax over T_in (max-subtracted, as tf.nn.softmax), by wave 0 alone: no block barriers
    if (wave == 0) {
        float mx = -INFINITY;
        for (int t = lane; t < a.T_in; t += 64) mx = fmaxf(mx, e[t]);
        mx = wave_max(mx);
        float sum = 0.f;
        for (int t = lane; t < a.T_in; t += 64) {
            const float p = expf(e[t] - mx);
            e[t] = p;
            sum += p;
        }
        sum = wave_sum(sum);
        const float inv = 1.f / sum;
        float* arow = a.align + (size_t)b * a.T_in;
        for (int t = lane; t < a.T_in; t += 64) {
            const float p = e[t] * inv;
            e[t] = p;
            arow[t] = p;
        }
    }
    __syncthreads();
    // ---- context: wave owns t = wave + 16*j, partial sums reduced across waves in fixed order
    const float4* vb = reinterpret_cast<const float4*>(a.values) + (size_t)b * a.T_in * VQ;
    for (int dq = lane; dq < VQ; dq += 64) {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        for (int tb = wave; tb < a.T_in; tb += ATTN_WAVES * ATTN_DEPTH) {
            float4 x[ATTN_DEPTH];
#pragma unroll
            for (int j = 0; j < ATTN_DEPTH; ++j) {
                const int t = tb + ATTN_WAVES * j;
                x[j] = t < a.T_in ? vb[(size_t)t * VQ + dq] : make_float4(0.f, 0.f, 0.f, 0.f);
            }
#pragma unroll
            for (int j = 0; j < ATTN_DEPTH; ++j) {
                const int t = tb + ATTN_WAVES * j;
                const float p = t < a.T_in ? e[t] : 0.f;
                acc.x = fmaf(p, x[j].x, acc.x); acc.y = fmaf(p, x[j].y, acc.y);
                acc.z = fmaf(p, x[j].z, acc.z); acc.w = fmaf(p, x[j].w, acc.w);
            }
        }
        cred[wave * VQ + dq] = acc;
    }
    __syncthreads();
    for (int dq = tid; dq < VQ; dq += ATTN_THREADS) {
        f32x4 acc = cred[dq];
#pragma unroll
        for (int w = 1; w < ATTN_WAVES; ++w) acc += cred[w * VQ + dq];
        reinterpret_cast<f32x4*>(a.ctx)[(size_t)dq * a.Bpad + b] = acc;
    }
}

void launch_dec_attention(const DecAttnArgs& a, hipStream_t s, Timing* tm) {
    if (tm) tm->begin(KK_DEC_ATTN, s);
    const size_t shm = (size_t)((a.T_in + 3) & ~3) * 4 + (size_t)ATTN_WAVES * (a.Dv / 4) * 16;
    hipLaunchKernelGGL(dec_attention_kernel, dim3(a.B), dim3(ATTN_THREADS), shm, s, a);
    if (tm) tm->end(KK_DEC_ATTN, s);
}

// ---------------------------------------------------------------------------------------
// Finalisation: stop test bookkeeping, [S][NR/4][Bpad][4] -> mel [B][S*NR],
// alignS [S][B][T_in] -> align [B][T_in][S] (model/tacotron.py:51).
// nz[s][b] = any(out[s][:, b] != 0) from the step outputs (stop test of helpers.py:75). `gran` = 1: quads are
// stored as {x0,tag,x1,tag,x2,tag,x3,tag} (persistent decoder), else plain float4.
__global__ void dec_nonzero_kernel(const float* __restrict__ outQ, int NRq, int Bpad, int gran, int32_t* __restrict__ nz) {
    const int s = blockIdx.x;
    const int qf = gran ? 8 : 4;
    const float* p = outQ + (size_t)s * NRq * Bpad * qf;
    for (int b = threadIdx.x; b < Bpad; b += blockDim.x) {
        int any = 0;
        for (int r = 0; r < NRq; ++r) {
            const float* q = p + ((size_t)r * Bpad + b) * qf;
            if (gran) any |= (q[0] != 0.f) | (q[2] != 0.f) | (q[4] != 0.f) | (q[6] != 0.f);
            else any |= (q[0] != 0.f) | (q[1] != 0.f) | (q[2] != 0.f) | (q[3] != 0.f);
        }
        nz[(size_t)s * Bpad + b] = any;
    }
}

__global__ __launch_bounds__(1024) void dec_steps_kernel(const int32_t* __restrict__ nz, int S, int B, int Bpad,
                                                         int32_t* steps_run) {
    // dynamic_decode: finished_b |= all(out_b == 0) (sticky); the loop stops after the first step at which every
    // utterance is finished. first_zero[b] = first step with an all-zero output; steps = max_b first_zero + 1.
    extern __shared__ int32_t first_zero[];
    for (int b = threadIdx.x; b < B; b += blockDim.x) first_zero[b] = S;
    __syncthreads();
    for (int i = threadIdx.x; i < S * Bpad; i += blockDim.x) {
        const int s = i / Bpad, b = i - s * Bpad;
        if (b < B && nz[i] == 0) atomicMin(&first_zero[b], s);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        int steps = 0;
        for (int b = 0; b < B; ++b) steps = max(steps, first_zero[b] + 1);
        *steps_run = min(steps, S);
    }
}

__global__ void dec_mel_kernel(const float* __restrict__ outQ, const int32_t* __restrict__ steps_run, int S, int NR,
                               int B, int Bpad, int gran, float* __restrict__ mel) {
    const int steps = *steps_run;
    const long total = (long)B * S * NR;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int n = (int)(i % NR);
        const long r = i / NR;
        const int s = (int)(r % S);
        const int b = (int)(r / S);
        float v = 0.f;
        if (s < steps) {
            const size_t q = ((size_t)s * (NR >> 2) + (n >> 2)) * Bpad + b;
            v = gran ? outQ[q * 8 + 2 * (n & 3)] : outQ[q * 4 + (n & 3)];
        }
        mel[i] = v;
    }
}

__global__ void dec_align_kernel(const float* __restrict__ alignS, const int32_t* __restrict__ steps_run, int S, int B,
                                 int T_in, float* __restrict__ align) {
    const int steps = *steps_run;
    const long total = (long)B * T_in * S;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int s = (int)(i % S);
        const long r = i / S;
        const int t = (int)(r % T_in);
        const int b = (int)(r / T_in);
        align[i] = s < steps ? alignS[((size_t)s * B + b) * T_in + t] : 0.f;
    }
}

// targets [B, S*r, M] -> teachQ[st][M/4][Bpad] quads of frame (st+1)*r - 1, st < S (model/helpers.py:15,42:
// TrainingHelper feeds every r-th target frame); padded columns zero
__global__ void dec_pack_teacher_kernel(const float* __restrict__ targets, float4* __restrict__ teachQ, int S, int r,
                                        int M, int B, int Bpad) {
    const long total = (long)S * (M / 4) * Bpad;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int b = (int)(i % Bpad);
        const int q = (int)((i / Bpad) % (M / 4));
        const int st = (int)(i / ((long)Bpad * (M / 4)));
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (b < B) v = *reinterpret_cast<const float4*>(targets + ((size_t)b * S * r + (size_t)(st + 1) * r - 1) * M + 4 * q);
        teachQ[i] = v;
    }
}
void launch_dec_pack_teacher(const float* targets, float* teachQ, int S, int r, int M, int B, int Bpad, hipStream_t s) {
    const long total = (long)S * (M / 4) * Bpad;
    const int blocks = (int)std::min<long>(1024, (total + 255) / 256);
    hipLaunchKernelGGL(dec_pack_teacher_kernel, dim3(blocks), dim3(256), 0, s, targets, (float4*)teachQ, S, r, M, B, Bpad);
}

void launch_dec_finalize(const float* outQ, int gran, const float* alignS, int32_t* nz, int S, int NR, int B, int Bpad,
                         int T_in, float* mel, float* align, int32_t* steps_run, hipStream_t s, Timing* tm,
                         bool stop_test) {
    if (tm) tm->begin(KK_DEC_MISC, s);
    if (stop_test) hipLaunchKernelGGL(dec_nonzero_kernel, dim3(S), dim3(64), 0, s, outQ, NR / 4, Bpad, gran, nz);
    else (void)hipMemsetAsync(nz, 0x01, (size_t)S * Bpad * sizeof(int32_t), s);   // every step "non-zero": no stop
    hipLaunchKernelGGL(dec_steps_kernel, dim3(1), dim3(1024), (size_t)B * 4, s, nz, S, B, Bpad, steps_run);
    long total = (long)B * S * NR;
    int blocks = (int)std::min<long>(2048, (total + 255) / 256);
    hipLaunchKernelGGL(dec_mel_kernel, dim3(blocks), dim3(256), 0, s, outQ, steps_run, S, NR, B, Bpad, gran, mel);
    total = (long)B * T_in * S;
    blocks = (int)std::min<long>(2048, (total + 255) / 256);
    hipLaunchKernelGGL(dec_align_kernel, dim3(blocks), dim3(256), 0, s, alignS, steps_run, S, B, T_in, align);
    if (tm) tm->end(KK_DEC_MISC, s);
}

}  // namespace taco
