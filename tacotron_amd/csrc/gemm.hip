// conv_gemm: dense layers and SAME conv1d as an implicit GEMM on the fp32 matrix cores
// (v_mfma_f32_32x32x2_f32: exact f32, a k-ordered fmaf chain per output element).
//
// Replaces tf.layers.dense / tf.layers.conv1d + activation + tf.layers.batch_normalization
// (inference affine) + the CBHG residual add of the reference (model/modules.py:26,74-84,
// 120-121,138-140; model/decoder.py:57) and BahdanauAttention's memory_layer
// (model/decoder.py:20).  The max-pool of modules.py:70-71 is folded into the A-operand
// loader of the following projection conv.
//
// Tiling: BMxBN block, 256 threads = 2x2 waves, each wave (BM/2)x(BN/2) as 32x32 MFMA
// tiles; BK = 16 staged through LDS k-major so that the one-VGPR A/B fragments
// (lane l: A[m=l&31][k=l>>5], B[k=l>>5][n=l&31]) are conflict-free ds_read_b32.
// Global->register prefetch of tile k+1 overlaps the MFMAs of tile k (one barrier per tile).
#include "kernels.h"

namespace taco {

typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ float4 max4(float4 a, float4 b) {
    return make_float4(fmaxf(a.x, b.x), fmaxf(a.y, b.y), fmaxf(a.z, b.z), fmaxf(a.w, b.w));
}

__device__ __forceinline__ float apply_act(float v, int act, bool second_half) {
    switch (act) {
        case ACT_RELU: return fmaxf(v, 0.f);
        case ACT_SIGMOID: return 1.f / (1.f + expf(-v));
        case ACT_TANH: return tanhf(v);
        case ACT_HIGHWAY_SPLIT: return second_half ? 1.f / (1.f + expf(-v)) : fmaxf(v, 0.f);
        default: return v;
    }
}

template <int BM, int BN>
__global__ __launch_bounds__(256) void conv_gemm_kernel(GemmArgs a) {
    constexpr int BK = 16, PAD = 4;
    constexpr int WM = BM / 2, WN = BN / 2;
    constexpr int TM = WM / 32, TN = WN / 32;
    constexpr int NA = BM * 4 / 256;          // float4 A loads per thread per tile
    constexpr int NB = BK * BN / 4 / 256;     // float4 B loads per thread per tile
    __shared__ __attribute__((aligned(16))) float As[2][BK][BM + PAD];
    __shared__ __attribute__((aligned(16))) float Bs[2][BK][BN + PAD];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int m0 = blockIdx.y * BM;
    const int n0 = blockIdx.x * BN;
    const int wm = (wave >> 1) * WM;
    const int wn = (wave & 1) * WN;

    // per-thread A rows (fixed over the K loop)
    const float* arow[NA];   // start of this row's sequence (b*T rows in)
    int at[NA];              // t within the sequence, or -1 for rows past M
    int arow_l[NA], akq[NA];
#pragma unroll
    for (int i = 0; i < NA; ++i) {
        const int f = tid + 256 * i;
        const int row = f >> 2;
        arow_l[i] = row;
        akq[i] = f & 3;
        const int m = m0 + row;
        if (m < a.M) {
            const int t = m % a.T;
            at[i] = t;
            arow[i] = a.X + (size_t)(m - t) * a.ldx;
        } else {
            at[i] = -1;
            arow[i] = a.X;
        }
    }
    const int nk = a.taps * a.Cin / BK;

    float4 ra[NA], rb[NB];
    int tap = 0, c0 = 0;     // position of the NEXT tile to fetch
    auto fetch = [&]() {
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            const int t2 = at[i] + tap - a.padl;
            if (at[i] >= 0 && t2 >= 0 && t2 < a.T) {
                const float* p = arow[i] + (size_t)t2 * a.ldx + c0 + 4 * akq[i];
                v = *reinterpret_cast<const float4*>(p);
                if (a.pool && t2 + 1 < a.T) v = max4(v, *reinterpret_cast<const float4*>(p + a.ldx));
            }
            ra[i] = v;
        }
        const int k0 = tap * a.Cin + c0;
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const int f = tid + 256 * i;
            const int k = f / (BN / 4);
            const int nq = f % (BN / 4);
            rb[i] = *reinterpret_cast<const float4*>(a.W + (size_t)(k0 + k) * a.ldw + n0 + 4 * nq);
        }
        c0 += BK;
        if (c0 == a.Cin) { c0 = 0; ++tap; }
    };
    auto stash = [&](int buf) {
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const int k = 4 * akq[i];
            As[buf][k + 0][arow_l[i]] = ra[i].x;
            As[buf][k + 1][arow_l[i]] = ra[i].y;
            As[buf][k + 2][arow_l[i]] = ra[i].z;
            As[buf][k + 3][arow_l[i]] = ra[i].w;
        }
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const int f = tid + 256 * i;
            const int k = f / (BN / 4);
            const int nq = f % (BN / 4);
            *reinterpret_cast<float4*>(&Bs[buf][k][4 * nq]) = rb[i];
        }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    fetch();
    stash(0);
    __syncthreads();
    const int lk = lane >> 5, li = lane & 31;
    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < nk) fetch();
#pragma unroll
        for (int kk = 0; kk < BK / 2; ++kk) {
            float af[TM], bf[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) af[i] = As[buf][2 * kk + lk][wm + 32 * i + li];
#pragma unroll
            for (int j = 0; j < TN; ++j) bf[j] = Bs[buf][2 * kk + lk][wn + 32 * j + li];
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i], bf[j], acc[i][j], 0, 0, 0);
        }
        if (kt + 1 < nk) stash(buf ^ 1);
        __syncthreads();
    }

    // epilogue: lane holds column n = ..+(lane&31); rows (r&3) + 8*(r>>2) + 4*(lane>>5)
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int n = n0 + wn + 32 * j + li;
        if (n >= a.N) continue;
        const float bias = a.bias ? a.bias[n] : 0.f;
        const float sc = a.scale ? a.scale[n] : 1.f;
        const float sh = a.shift ? a.shift[n] : 0.f;
        const bool second = n >= (a.N >> 1);
#pragma unroll
        for (int i = 0; i < TM; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * lk;
                if (m >= a.M) continue;
                float v = apply_act(acc[i][j][r] + bias, a.act, second);
                if (a.scale) v = v * sc + sh;
                if (a.R) v += a.R[(size_t)m * a.ldr + n];
                a.C[(size_t)m * a.ldc + n] = v;
            }
        }
    }
}

void launch_conv_gemm(const GemmArgs& a, hipStream_t s, Timing* tm) {
    if (tm) tm->begin(KK_CONV_GEMM, s);
    // tile choice: big tiles when they still give >= ~2 blocks per CU, else 64x64
    const long blocks128 = (long)((a.M + 127) / 128) * ((a.N + 127) / 128);
    if (blocks128 >= 384 && a.N >= 128) {
        dim3 grid((a.N + 127) / 128, (a.M + 127) / 128);
        hipLaunchKernelGGL((conv_gemm_kernel<128, 128>), grid, dim3(256), 0, s, a);
    } else {
        dim3 grid((a.N + 63) / 64, (a.M + 63) / 64);
        hipLaunchKernelGGL((conv_gemm_kernel<64, 64>), grid, dim3(256), 0, s, a);
    }
    if (tm) tm->end(KK_CONV_GEMM, s);
}

// ---------------------------------------------------------------------------------------
__global__ void highway_combine_kernel(const float4* __restrict__ HT, const float4* __restrict__ x,
                                       float4* __restrict__ y, long n4, int dq) {
    // HT row = [H (dq float4) | T (dq float4)], x/y rows = dq float4
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
        const long m = i / dq;
        const int c = (int)(i - m * dq);
        const float4 H = HT[m * 2 * dq + c];
        const float4 T = HT[m * 2 * dq + dq + c];
        const float4 xv = x[i];
        float4 o;
        o.x = H.x * T.x + xv.x * (1.f - T.x);
        o.y = H.y * T.y + xv.y * (1.f - T.y);
        o.z = H.z * T.z + xv.z * (1.f - T.z);
        o.w = H.w * T.w + xv.w * (1.f - T.w);
        y[i] = o;
    }
}

void launch_highway_combine(const float* HT, const float* x, float* y, int M, int depth, hipStream_t s, Timing* tm) {
    if (tm) tm->begin(KK_HIGHWAY, s);
    const long n4 = (long)M * depth / 4;
    int blocks = (int)((n4 + 255) / 256);
    if (blocks > 2048) blocks = 2048;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(highway_combine_kernel, dim3(blocks), dim3(256), 0, s, (const float4*)HT, (const float4*)x,
                       (float4*)y, n4, depth / 4);
    if (tm) tm->end(KK_HIGHWAY, s);
}

__global__ void embed_gather_kernel(const int32_t* __restrict__ ids, const float4* __restrict__ lut,
                                    float4* __restrict__ out, long n4, int wq, int num_rows) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
        const long m = i / wq;
        const int c = (int)(i - m * wq);
        int id = ids[m];
        id = id < 0 ? 0 : (id >= num_rows ? num_rows - 1 : id);
        out[i] = lut[(long)id * wq + c];
    }
}

void launch_embed_gather(const int32_t* ids, const float* lut, float* out, int M, int width, int num_rows,
                         hipStream_t s, Timing* tm) {
    if (tm) tm->begin(KK_GATHER, s);
    const long n4 = (long)M * width / 4;
    int blocks = (int)((n4 + 255) / 256);
    if (blocks > 2048) blocks = 2048;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(embed_gather_kernel, dim3(blocks), dim3(256), 0, s, ids, (const float4*)lut, (float4*)out, n4,
                       width / 4, num_rows);
    if (tm) tm->end(KK_GATHER, s);
}

}  // namespace taco
