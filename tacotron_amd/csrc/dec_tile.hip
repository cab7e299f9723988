// dec_tile: the whole dynamic_decode loop (model/decoder.py:18-40, model/rnn_wrappers.py:24-26,51-53,
// model/helpers.py:66-78; step maths SURVEY.md A.7) as ONE launch in which every 16-utterance batch tile is owned by
// ONE workgroup - the throughput form of the decode loop (taco_set_decode_graph mode 3).
//
// Why: dec_persistent (mode 2) minimises the LATENCY of a decode: 32 workgroups per tile keep the decoder's weights in
// registers and hand activations to each other 11 times per step; each of them owns a whole CU (512 registers per
// lane) and mostly waits, so a B = 32 decode holds 64 of the 256 CUs for ~8 ms: half of the chip's CU-time per batch
// with several batches in flight. Here a tile's step runs inside one workgroup: no inter-workgroup hand-off exists at
// all (stages are separated by workgroup barriers), the weights (6 MB per step) stream from L2 / Infinity Cache straight
// into MFMA A-operand fragments, and the workgroup is small enough (8 waves, <= 128 registers per lane, < 8 KB of LDS)
// that workgroups of other kernels (the GEMMs and the vocoder of other batches) run beside it on the same CU. A tile's
// step is then bound by the fp32 matrix pipe of the one CU it runs on (16 utterances x 1.5 M MAC = 23.5 k
// v_mfma_f32_16x16x4_f32 = 188 k cycles over four SIMDs) plus the attention's sweep over the tile's keys and values:
// ~4x the latency of mode 2 per decode for ~1/15 of its CU-time. A caller that keeps enough batches in flight
// (bench.py, SynthesisPipeline) gets the chip back for the GEMMs; a single utterance or batch alone wants mode 2.
//
// Arithmetic: the stage bodies of the per-stage kernels (decoder.hip: same packed weights, same Q-layout buffers, same
// epilogues), with the K sum of an output tile taken by one wave in ascending order (two alternating accumulators),
// so an utterance's result does not depend on the batch it is in; attention: one wave per utterance, rows in ascending
// order. Differs from modes 0-2 by fp32 summation order only.
#include <atomic>

#include "kernels.h"
#include "dec_common.h"

namespace taco {

constexpr int DT_WAVES = 8;
constexpr int DT_THREADS = 64 * DT_WAVES;
constexpr int DT_DEPTH = 4;            // k16-steps per chunk; two chunks are in flight per wave (double buffer)
constexpr int DT_ATT_DEPTH = 6;        // key / value rows of the attention whose loads are in flight together

struct DtFrag {
    float4 w[DT_DEPTH], x[DT_DEPTH];
};

// One GEMM stage for this workgroup's batch tile: wave w takes output tiles w, w + 8, ...; its (tile, chunk) pairs form
// one stream whose loads run a chunk ahead of the MFMAs, across tile boundaries too. The chunk body is branch-free (so
// that the compiler can count the loads in flight instead of draining them): a k16-step beyond the stage's K re-reads
// the tile's last weights against the zero quads of the <GO> frame, and an issue beyond the wave's last chunk re-reads
// that chunk (never consumed). All activations of the per-stage path live in one allocation (`region`), read through
// one buffer resource: wave-uniform offset per k16-step + one per-lane offset.
__device__ __forceinline__ void dt_stage(const DecTileStage& st, unsigned src0_off, float* outp, float* const* buf,
                                         __amdgpu_buffer_rsrc_t rs, const float* region, int b0, int Bpad, int wave,
                                         int lane) {
    const int nk16 = st.k16_0 + st.k16_1;
    const int nch = (nk16 + DT_DEPTH - 1) / DT_DEPTH;
    const int ntiles = st.N >> 4;
    const int my_tiles = ntiles > wave ? (ntiles - wave + DT_WAVES - 1) / DT_WAVES : 0;
    if (my_tiles == 0) return;
    const int lb = b0 + (lane & 15), lq = lane >> 4;
    const unsigned lane_bytes = (unsigned)(lq * Bpad + lb) * 16u;
    const unsigned step_bytes = (unsigned)(4 * Bpad) * 16u;              // one k16-step of a Q-layout buffer
    const unsigned off0 = src0_off;                                       // byte offsets in `region`
    const unsigned off1 = st.src1 >= 0 ? (unsigned)((buf[st.src1] - region) * 4) : 0u;
    const unsigned offz = (unsigned)((buf[DTB_GO] - region) * 4);         // zeros (never written)
    const float4* Wp = reinterpret_cast<const float4*>(st.Wp) + lane;
    const int k16_0 = st.k16_0;
    const int total = my_tiles * nch;

    auto issue = [&](DtFrag& f, int c) {                                  // chunk c of the wave's stream (clamped)
        c = c < total ? c : total - 1;
        const int t = c / nch, ch = c - t * nch;
        const float4* wt = Wp + (size_t)(wave + DT_WAVES * t) * nk16 * 64;
#pragma unroll
        for (int i = 0; i < DT_DEPTH; ++i) {
            const int ks = ch * DT_DEPTH + i;
            const int kw = ks < nk16 ? ks : nk16 - 1;
            f.w[i] = wt[(size_t)kw * 64];
            const unsigned uni = ks < k16_0 ? off0 + (unsigned)ks * step_bytes
                               : (ks < nk16 ? off1 + (unsigned)(ks - k16_0) * step_bytes : offz);
            f.x[i] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rs, lane_bytes, uni, 0));
        }
    };
    auto epilogue = [&](int ntile, f32x4 v) {
        const int n = ntile * 16 + 4 * lq;                     // first of this lane's 4 output rows
        const size_t e = (size_t)(n >> 2) * Bpad + lb;         // float4 index in a Q-layout buffer
        if (st.bias) v += *reinterpret_cast<const f32x4*>(st.bias + n);
        switch (st.epi) {
            case DE_LINEAR:
            case DE_OUT:
                reinterpret_cast<f32x4*>(outp)[e] = v;
                break;
            case DE_RELU:
                v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
                reinterpret_cast<f32x4*>(outp)[e] = v;
                break;
            case DE_GATES: {
                f32x4 g;
                g.x = dsigmoid(v.x); g.y = dsigmoid(v.y); g.z = dsigmoid(v.z); g.w = dsigmoid(v.w);
                const int H = st.N >> 1;
                if (n < H) reinterpret_cast<f32x4*>(outp)[e] = g * reinterpret_cast<const f32x4*>(buf[st.h])[e];   // r * h
                else reinterpret_cast<f32x4*>(buf[DTB_U])[e - (size_t)(H >> 2) * Bpad] = g;                          // u
                break;
            }
            case DE_CAND: {
                f32x4 c;
                c.x = dtanh(v.x); c.y = dtanh(v.y); c.z = dtanh(v.z); c.w = dtanh(v.w);
                const f32x4 ph = reinterpret_cast<const f32x4*>(buf[st.h])[e];
                const f32x4 pu = reinterpret_cast<const f32x4*>(buf[DTB_U])[e];
                const f32x4 hn = pu * ph + (1.f - pu) * c;
                reinterpret_cast<f32x4*>(buf[st.h])[e] = hn;
                if (st.res_out >= 0)
                    reinterpret_cast<f32x4*>(buf[st.res_out])[e] = hn + reinterpret_cast<const f32x4*>(buf[st.res])[e];
                break;
            }
        }
    };
    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
    int c_ch = 0, c_tile = wave;
    auto consume = [&](const DtFrag& f) {
#pragma unroll
        for (int i = 0; i < DT_DEPTH; ++i) {
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(f.w[i].x, f.x[i].x, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(f.w[i].y, f.x[i].y, acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(f.w[i].z, f.x[i].z, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(f.w[i].w, f.x[i].w, acc1, 0, 0, 0);
        }
        if (++c_ch == nch) {
            epilogue(c_tile, acc0 + acc1);
            acc0 = f32x4{0.f, 0.f, 0.f, 0.f};
            acc1 = acc0;
            c_ch = 0;
            c_tile += DT_WAVES;
        }
    };

    DtFrag fa, fb;
    issue(fa, 0);
    for (int c = 0; c < total; c += 2) {
        issue(fb, c + 1);
        consume(fa);
        issue(fa, c + 2);
        if (c + 1 < total) consume(fb);
    }
}

// Bahdanau attention of one utterance by ONE wave (decoder.py:20; TF _bahdanau_score, softmax over ALL T_in: no
// memory_sequence_length, A.8): lane owns the d-quad `lane` of the 256-wide keys / values; rows in ascending order.
__device__ __forceinline__ void dt_attention(const DecTileArgs& a, float* const* buf, int b, float* e_s, float* arow,
                                             int lane) {
    const int T_in = a.T_in, Bpad = a.Bpad;
    const float4 q = reinterpret_cast<const float4*>(buf[DTB_Q])[(size_t)lane * Bpad + b];
    const float4 vv = reinterpret_cast<const float4*>(a.attn_v)[lane];
    const float4* kb = reinterpret_cast<const float4*>(a.keys) + (size_t)b * T_in * 64 + lane;
    for (int tb = 0; tb < T_in; tb += DT_ATT_DEPTH) {
        float4 k[DT_ATT_DEPTH];
#pragma unroll
        for (int j = 0; j < DT_ATT_DEPTH; ++j)
            k[j] = tb + j < T_in ? kb[(size_t)(tb + j) * 64] : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int j = 0; j < DT_ATT_DEPTH; ++j) {
            float s = vv.x * fast_tanh(k[j].x + q.x) + vv.y * fast_tanh(k[j].y + q.y) +
                      vv.z * fast_tanh(k[j].z + q.z) + vv.w * fast_tanh(k[j].w + q.w);
            s = wave_sum(s);
            if (lane == 0 && tb + j < T_in) e_s[tb + j] = s;
        }
    }
    // softmax (max-subtracted, as tf.nn.softmax)
    float mx = -INFINITY;
    for (int t = lane; t < T_in; t += 64) mx = fmaxf(mx, e_s[t]);
    mx = wave_max(mx);
    float sum = 0.f;
    for (int t = lane; t < T_in; t += 64) {
        const float p = expf(e_s[t] - mx);
        e_s[t] = p;
        sum += p;
    }
    sum = wave_sum(sum);
    const float inv = 1.f / sum;
    for (int t = lane; t < T_in; t += 64) {
        const float p = e_s[t] * inv;
        e_s[t] = p;
        arow[t] = p;                                  // alignment row of this step (tacotron.py:51 transposes at the end)
    }
    // context
    const float4* vb = reinterpret_cast<const float4*>(a.values) + (size_t)b * T_in * 64 + lane;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int tb = 0; tb < T_in; tb += DT_ATT_DEPTH) {
        float4 x[DT_ATT_DEPTH];
#pragma unroll
        for (int j = 0; j < DT_ATT_DEPTH; ++j)
            x[j] = tb + j < T_in ? vb[(size_t)(tb + j) * 64] : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int j = 0; j < DT_ATT_DEPTH; ++j) {
            const float p = tb + j < T_in ? e_s[tb + j] : 0.f;
            acc.x = fmaf(p, x[j].x, acc.x); acc.y = fmaf(p, x[j].y, acc.y);
            acc.z = fmaf(p, x[j].z, acc.z); acc.w = fmaf(p, x[j].w, acc.w);
        }
    }
    reinterpret_cast<f32x4*>(buf[DTB_CTX])[(size_t)lane * Bpad + b] = acc;
}

__global__ __launch_bounds__(DT_THREADS, 2) void dec_tile_kernel(DecTileArgs a) {
    extern __shared__ __attribute__((aligned(16))) float dt_smem[];   // [DT_WAVES][T_in rounded to 4] scores
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // Placement: the launch has 8 workgroups per tile, dealt round-robin to the 8 XCDs; the one whose slot matches the
    // tile's XCD of this launch (a.xcd0 + tile, rotating from launch to launch) does the work, the others leave at once:
    // the few long-lived workgroups of the decodes in flight spread over all XCDs instead of piling up on the first ones.
    // (Speed only: any placement computes the same thing.)
    const int tile = blockIdx.x >> 3;
    if ((int)(blockIdx.x & 7) != ((a.xcd0 + tile) & 7)) return;
    const int b0 = tile * 16;
    float* const* buf = a.buf;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(a.region, 0, a.region_bytes, 0x00020000);
    float* e_s = dt_smem + (size_t)wave * ((a.T_in + 3) & ~3);
    const int NRq = a.NR >> 2;
    for (int s = 0; s < a.S; ++s) {
        // previous frame: <GO> (zeros), then the last of the r frames just produced (helpers.py:77)
        const float* prev = s == 0 ? buf[DTB_GO] : a.outQ + ((size_t)(s - 1) * a.NR + (a.NR - a.M)) * a.Bpad;
        float* outq = a.outQ + (size_t)s * a.NR * a.Bpad;
        (void)NRq;
#pragma unroll 1
        for (int i = 0; i < DT_STAGES; ++i) {
            const DecTileStage& st = a.st[i];
            const float* src0 = i == 0 ? prev : buf[st.src0];
            float* outp = i == DT_STAGES - 1 ? outq : (st.out >= 0 ? buf[st.out] : nullptr);
            dt_stage(st, (unsigned)((src0 - a.region) * 4), outp, buf, rs, a.region, b0, a.Bpad, wave, lane);
            __syncthreads();
            if (i == DT_QUERY_STAGE) {
                // Bahdanau attention: one wave per utterance of the tile
                for (int u = wave; u < 16; u += DT_WAVES) {
                    const int b = b0 + u;
                    if (b < a.B) dt_attention(a, buf, b, e_s, a.alignS + ((size_t)s * a.B + b) * a.T_in, lane);
                }
                __syncthreads();
            }
        }
    }
}

int launch_dec_tile(const DecTileArgs& a, hipStream_t s, Timing* tm) {
    const size_t lds = (size_t)DT_WAVES * ((a.T_in + 3) & ~3) * sizeof(float);
    if (lds > 64 * 1024) return -1;
    if (tm) tm->begin(KK_DEC_TILE, s);
    static std::atomic<unsigned> launches{0};
    DecTileArgs b = a;
    b.xcd0 = (int)(launches.fetch_add((unsigned)(a.Bpad / 16)) & 7u);
    hipLaunchKernelGGL(dec_tile_kernel, dim3(8 * (a.Bpad / 16)), dim3(DT_THREADS), lds, s, b);
    if (tm) tm->end(KK_DEC_TILE, s);
    return 0;
}

}  // namespace taco
