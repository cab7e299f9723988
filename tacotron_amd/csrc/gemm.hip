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
// tiles; BK = 16 staged through LDS: A row-major as it is loaded (row stride 20 floats), a lane's
// eight k values of a tile in two ds_read_b128 (lane half lk feeds k = 8 lk + kk to MFMA kk); B k-major,
// one conflict-free ds_read_b32 per fragment (lane l: B[k][n = l & 31]).
// Global->register prefetch of tile k+1 overlaps the MFMAs of tile k (one barrier per tile).
#include "kernels.h"
#include <cstring>

namespace taco {

typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ float4 max4(float4 a, float4 b) {
    return make_float4(fmaxf(a.x, b.x), fmaxf(a.y, b.y), fmaxf(a.z, b.z), fmaxf(a.w, b.w));
}

__device__ __forceinline__ float apply_act(float v, int act) {
    switch (act) {
        case ACT_RELU: return fmaxf(v, 0.f);
        case ACT_SIGMOID: return 1.f / (1.f + expf(-v));
        case ACT_TANH: return tanhf(v);
        default: return v;
    }
}

// POOL: the A operand is max(x[t], x[t+1]) (the max-pool folded into proj_1's loader); a template parameter so that the
// k loop is one basic block, whose address arithmetic and loads the scheduler can spread between the MFMAs
// NW: waves per workgroup, 4 (2 x 2 over the tile) or 8 (2 x 4: a wave owns BM/2 x BN/4)
template <int BM, int BN, bool POOL, int NW = 4>
__global__ __launch_bounds__(64 * NW) void conv_gemm_kernel(GemmArgs a) {
    constexpr int BK = 16, PAD = 4, AST = BK + 4;   // A row stride 20: the 16 lanes of a ds_read_b128 group hit 16 bank quads
    constexpr int NTH = 64 * NW, WCOLS = NW / 2;
    constexpr int WM = BM / 2, WN = BN / WCOLS;
    constexpr int TM = WM / 32, TN = WN / 32;
    constexpr int NA = BM * 4 / NTH;          // float4 A loads per thread per tile
    constexpr int NB = BK * BN / 4 / NTH;     // float4 B loads per thread per tile
    static_assert(NA >= 1 && NB >= 1 && TM >= 1 && TN >= 1, "tile too small for this many waves");
    __shared__ __attribute__((aligned(16))) float As[2][BM][AST];     // row-major: stashed as it is loaded (16 bytes along k)
    __shared__ __attribute__((aligned(16))) float Bs[2][BK][BN + PAD];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    // Tile of this workgroup. The hardware deals workgroups to the 8 XCDs round-robin by linear id: id = 8 l + xcd.
    // All column tiles of a row block get consecutive l on ONE XCD (they read the same A rows: the second to last of
    // them find the rows in that XCD's L2 instead of fetching them again from memory), row blocks go round the XCDs.
    const int nx = (a.N + BN - 1) / BN;
    int m0, n0;
    {
        const int id = blockIdx.x, xcd = id & 7, l = id >> 3;
        const int yy = l / nx;
        n0 = (l - yy * nx) * BN;
        m0 = (yy * 8 + xcd) * BM;
        if (m0 >= a.M) return;                    // (grid rounded up to whole rounds of 8 row blocks)
    }
    const int wm = (wave / WCOLS) * WM;
    const int wn = (wave % WCOLS) * WN;

    // a grouped launch runs several convolutions over the same input (the CBHG conv bank) as grid.z slices
    const float* W = a.W;
    const float* bias = a.bias;
    const float* scale = a.scale;
    const float* shift = a.shift;
    float* C = a.C;
    int taps = a.taps, padl = a.padl;
    if (a.groups) {
        const GemmGroup g = a.groups[blockIdx.z];
        W = g.W; bias = g.bias; scale = g.scale; shift = g.shift;
        taps = g.taps; padl = g.padl;
        C += g.coff;
    }
    // k-tile order: channel block outer, tap inner, so that the `taps` shifted reads of the same A rows follow each
    // other (L1/L2 hits) instead of coming Cin/16 tiles apart, which for the projection convolutions (Cin = 1024,
    // 2048) is past any cache: k-tile kt covers channels 16 (kt / taps) .. +15 of tap kt % taps.
    // split-K: grid.z slices of the k-tile range, raw partial sums out (splitk_epilogue_kernel finishes)
    const int nk_all = taps * a.Cin / BK;
    int kt0 = 0, kt1 = nk_all;
    if (a.splitk > 1) {
        kt0 = (int)((long)nk_all * blockIdx.z / a.splitk);
        kt1 = (int)((long)nk_all * (blockIdx.z + 1) / a.splitk);
    }

    // A operand through a buffer resource: rows outside [0, T) of their utterance (SAME padding) and rows past M
    // are fetched at an out-of-range offset, which returns zeros: no branch around a load
    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(a.X), 0, (int)(((size_t)(a.M - 1) * a.ldx + a.Cin) * 4), 0x00020000);
    constexpr unsigned OOB = 0x80000000u;
    unsigned rowoff[NA];     // byte offset of (sequence start row, this thread's k quad)
    int at[NA];              // t within the sequence, or a value that fails every range test for rows past M
#pragma unroll
    for (int i = 0; i < NA; ++i) {
        const int f = tid + NTH * i;
        const int row = f >> 2, kq = f & 3;
        const int m = m0 + row;
        const int t = m % a.T;
        at[i] = m < a.M ? t : -(1 << 20);
        rowoff[i] = (unsigned)(m - t) * (unsigned)a.ldx * 4u + 16u * kq;
    }
    const float* wp[NB];
#pragma unroll
    for (int i = 0; i < NB; ++i) {
        const int f = tid + NTH * i;
        const int k = f / (BN / 4), nq = f % (BN / 4);
        wp[i] = W + (size_t)k * a.ldw + n0 + 4 * nq;
    }
    int tap = kt0 % taps, c0 = (kt0 / taps) * BK;              // position of the NEXT tile to fetch
    const unsigned ldx4 = (unsigned)a.ldx * 4u;

    typedef float f32x4v __attribute__((ext_vector_type(4)));
    f32x4v ra[NA], rp[NA];
    f32x4v rb[NB];
#define CG_FETCH()                                                                                       \
    do {                                                                                                 \
        _Pragma("unroll") for (int i = 0; i < NA; ++i) {                                                 \
            const int t2 = at[i] + tap - padl;                                                           \
            const bool ok = (unsigned)t2 < (unsigned)a.T;                                                \
            const unsigned off = rowoff[i] + (unsigned)t2 * ldx4 + (unsigned)c0 * 4u;                    \
            const unsigned vo = ok ? off : OOB;                                                          \
            ra[i] = __builtin_bit_cast(f32x4v, __builtin_amdgcn_raw_buffer_load_b128(xr, vo, 0, 0));     \
            if (POOL) {                                                                                  \
                const unsigned vo2 = (ok && t2 + 1 < a.T) ? off + ldx4 : vo;                             \
                rp[i] = __builtin_bit_cast(f32x4v, __builtin_amdgcn_raw_buffer_load_b128(xr, vo2, 0, 0)); \
            }                                                                                            \
        }                                                                                                \
        const size_t wrow = (size_t)(tap * a.Cin + c0) * a.ldw;                                          \
        _Pragma("unroll") for (int i = 0; i < NB; ++i)                                                   \
            rb[i] = *reinterpret_cast<const f32x4v*>(wp[i] + wrow);                                      \
        ++tap;                                                                                           \
        if (tap == taps) { tap = 0; c0 += BK; }                                                          \
    } while (0)
#define CG_STASH(buf)                                                                                    \
    do {                                                                                                 \
        _Pragma("unroll") for (int i = 0; i < NA; ++i) {                                                 \
            const int f = tid + NTH * i;                                                                 \
            const int row = f >> 2, k = 4 * (f & 3);                                                     \
            f32x4v v = ra[i];                                                                            \
            if (POOL) v = __builtin_elementwise_max(v, rp[i]);                                           \
            *reinterpret_cast<f32x4v*>(&As[buf][row][k]) = v;                                            \
        }                                                                                                \
        _Pragma("unroll") for (int i = 0; i < NB; ++i) {                                                 \
            const int f = tid + NTH * i;                                                                 \
            const int k = f / (BN / 4), nq = f % (BN / 4);                                               \
            *reinterpret_cast<f32x4v*>(&Bs[buf][k][4 * nq]) = rb[i];                                     \
        }                                                                                                \
    } while (0)

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int lk = lane >> 5, li = lane & 31;
    // The k index a lane half feeds to MFMA kk of a tile is 8 lk + kk (A and B alike; any fixed pairing of the 16 k's
    // is a valid order of the sum): a lane's eight A values are then contiguous in its row, two ds_read_b128.
    // B fragments of k-pair kk+2 are read from LDS while the MFMAs of k-pair kk run.
#define CG_COMPUTE(buf)                                                                                  \
    do {                                                                                                 \
        float af[BK / 2][TM], bf[BK / 2][TN];                                                            \
        _Pragma("unroll") for (int i = 0; i < TM; ++i) {                                                 \
            const f32x4v a0 = *reinterpret_cast<const f32x4v*>(&As[buf][wm + 32 * i + li][8 * lk]);     \
            const f32x4v a1 = *reinterpret_cast<const f32x4v*>(&As[buf][wm + 32 * i + li][8 * lk + 4]); \
            af[0][i] = a0.x; af[1][i] = a0.y; af[2][i] = a0.z; af[3][i] = a0.w;                          \
            af[4][i] = a1.x; af[5][i] = a1.y; af[6][i] = a1.z; af[7][i] = a1.w;                          \
        }                                                                                                \
        _Pragma("unroll") for (int kk = 0; kk < BK / 2; ++kk)                                            \
            _Pragma("unroll") for (int j = 0; j < TN; ++j) bf[kk][j] = Bs[buf][8 * lk + kk][wn + 32 * j + li]; \
        _Pragma("unroll") for (int kk = 0; kk < BK / 2; ++kk)                                            \
            _Pragma("unroll") for (int i = 0; i < TM; ++i)                                               \
                _Pragma("unroll") for (int j = 0; j < TN; ++j)                                           \
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[kk][i], bf[kk][j], acc[i][j], 0, 0, 0); \
        /* issue order: all A reads and the B reads of two k-pairs first, then B of k-pair kk+2 before the MFMAs of kk */ \
        __builtin_amdgcn_sched_group_barrier(0x100, 2 * TM + 2 * TN, 0);                                 \
        _Pragma("unroll") for (int kk = 0; kk < BK / 2; ++kk) {                                          \
            __builtin_amdgcn_sched_group_barrier(0x008, TM * TN, 0);                                     \
            if (kk + 2 < BK / 2) __builtin_amdgcn_sched_group_barrier(0x100, TN, 0);                     \
        }                                                                                                \
    } while (0)

    if (kt1 > kt0) {
        CG_FETCH();
        CG_STASH(0);
        __syncthreads();
        int buf = 0;
        for (int kt = kt0; kt < kt1 - 1; ++kt) {
            CG_FETCH();              // tile kt+1 in flight under the MFMAs of tile kt
            __builtin_amdgcn_sched_barrier(0);   // (the loads stay up here: sunk among the MFMAs they are waited for at once)
            CG_COMPUTE(buf);
            __builtin_amdgcn_sched_barrier(0);   // (and what waits for them stays behind the MFMAs)
            CG_STASH(buf ^ 1);
            __syncthreads();
            buf ^= 1;
        }
        CG_COMPUTE(buf);
    }
#undef CG_FETCH
#undef CG_STASH
#undef CG_COMPUTE

    // epilogue: lane holds column n = ..+(lane&31); rows (r&3) + 8*(r>>2) + 4*(lane>>5)
    if (a.splitk > 1) {
        const int ldp = nx * BN;
        float* P = a.partial + (size_t)blockIdx.z * a.M * ldp;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int n = n0 + wn + 32 * j + li;
#pragma unroll
            for (int i = 0; i < TM; ++i) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int m = m0 + wm + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * lk;
                    if (m < a.M) P[(size_t)m * ldp + n] = acc[i][j][r];
                }
            }
        }
        return;
    }
    if constexpr (TN == 2) {
        if (a.act == ACT_HIGHWAY_FUSED) {
            // this wave's 64 columns = [32 H | 32 T] of units u0 .. u0+31; the lane holds both for unit u0 + li
            const int nh = n0 + wn + li, nt = nh + 32;
            const int unit = (n0 + wn) / 2 + li;
            if (nt < a.N) {
                const float bh = bias ? bias[nh] : 0.f, bt = bias ? bias[nt] : 0.f;
#pragma unroll
                for (int i = 0; i < TM; ++i) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int m = m0 + wm + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * lk;
                        if (m >= a.M) continue;
                        const float hh = fmaxf(acc[i][0][r] + bh, 0.f);
                        const float tt = 1.f / (1.f + expf(-(acc[i][1][r] + bt)));
                        const float x = a.R[(size_t)m * a.ldr + unit];
                        C[(size_t)m * a.ldc + unit] = hh * tt + x * (1.f - tt);
                    }
                }
            }
            return;
        }
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int n = n0 + wn + 32 * j + li;
        if (n >= a.N) continue;
        const float bv = bias ? bias[n] : 0.f;
        const float sc = scale ? scale[n] : 1.f;
        const float sh = shift ? shift[n] : 0.f;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * lk;
                if (m >= a.M) continue;
                float v = apply_act(acc[i][j][r] + bv, a.act);
                if (a.act == ACT_DB_NORM)
                    v = fminf(fmaxf((20.f * log10f(fmaxf(1e-5f, v)) - a.p0 - a.p1) / -a.p1, 0.f), 1.f);
                if (scale) v = v * sc + sh;
                if (a.R) v += a.R[(size_t)m * a.ldr + n];
                C[(size_t)m * a.ldc + n] = v;
            }
        }
    }
}

// finishes a split-K launch: sums the slices in a fixed order, then the same epilogue
__global__ void splitk_epilogue_kernel(GemmArgs a, int ldp) {
    const long n4 = (long)a.M * (ldp / 4);
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
        const int m = (int)(i / (ldp / 4));
        const int nb = 4 * (int)(i % (ldp / 4));
        if (nb >= a.N) continue;
        float4 sum = *reinterpret_cast<const float4*>(a.partial + (size_t)m * ldp + nb);
        for (int z = 1; z < a.splitk; ++z) {
            const float4 p = *reinterpret_cast<const float4*>(a.partial + ((size_t)z * a.M + m) * ldp + nb);
            sum.x += p.x; sum.y += p.y; sum.z += p.z; sum.w += p.w;
        }
        const float vs[4] = {sum.x, sum.y, sum.z, sum.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int n = nb + e;
            if (n >= a.N) break;
            float v = apply_act(vs[e] + (a.bias ? a.bias[n] : 0.f), a.act);
            if (a.act == ACT_DB_NORM)
                v = fminf(fmaxf((20.f * log10f(fmaxf(1e-5f, v)) - a.p0 - a.p1) / -a.p1, 0.f), 1.f);
            if (a.scale) v = v * a.scale[n] + a.shift[n];
            if (a.R) v += a.R[(size_t)m * a.ldr + n];
            a.C[(size_t)m * a.ldc + n] = v;
        }
    }
}

// The split is a function of K alone: an utterance's sums must not depend on how many other utterances share the
// launch (shard invariance), so M may not influence the reduction order. Only very deep reductions are split (the
// encoder's proj_1, K = 3*2048), whose M = B*T_in is small in every configuration of interest.
int conv_gemm_pick_splitk(int M, int N, int nk) {
    (void)M; (void)N;
    if (nk < 256) return 1;
    return std::min(8, nk / 64);
}

// 1-D grid over (row blocks rounded up to a multiple of 8) x column tiles: the kernel derives its tile from the id
static dim3 cg_grid(const GemmArgs& a, int bm, int bn, int z) {
    const int nx = (a.N + bn - 1) / bn, ny = (a.M + bm - 1) / bm;
    return dim3((unsigned)(nx * 8 * ((ny + 7) / 8)), 1, (unsigned)z);
}

void launch_conv_gemm(const GemmArgs& a_in, hipStream_t s, Timing* tm) {
    GemmArgs a = a_in;
    if (tm) tm->begin(KK_CONV_GEMM, s);
    const int groups = a.groups ? a.ngroups : 1;
    if (a.groups || !a.partial) a.splitk = 1;
    const int z = a.splitk > 1 ? a.splitk : groups;
    // tile choice: big tiles when one slice alone still gives ~a block per CU, else 64x64
    const long blocks128 = (long)((a.M + 127) / 128) * ((a.N + 127) / 128);
    static const char* tile_env = getenv("TACO_GEMM_TILE");     // (experiments: 64x128 or 128x64 for the big-tile cases)
    const bool big = (blocks128 >= 192 && a.N >= 128 && a.splitk == 1) || a.act == ACT_HIGHWAY_FUSED;
    if (big && tile_env && !strcmp(tile_env, "64x128")) {
        if (a.pool) hipLaunchKernelGGL((conv_gemm_kernel<64, 128, true>), cg_grid(a, 64, 128, z), dim3(256), 0, s, a);
        else hipLaunchKernelGGL((conv_gemm_kernel<64, 128, false>), cg_grid(a, 64, 128, z), dim3(256), 0, s, a);
    } else if (big && a.act != ACT_HIGHWAY_FUSED && (tile_env ? !strcmp(tile_env, "128x64")
                                                              : (a.N <= 128 || (a.N % 128 != 0 && a.N % 128 <= 64)))) {
        // half-width tiles where a 128-wide one would be the only one of its row block (N = 128: twice the workgroups
        // per CU, measured 60 against 71 us at M = 32000, K = 640) or mostly padding (N = 1025: 180 against 196 us);
        // the k order of an element's sum is the same for every tile shape, so the choice does not change a bit
        if (a.pool) hipLaunchKernelGGL((conv_gemm_kernel<128, 64, true>), cg_grid(a, 128, 64, z), dim3(256), 0, s, a);
        else hipLaunchKernelGGL((conv_gemm_kernel<128, 64, false>), cg_grid(a, 128, 64, z), dim3(256), 0, s, a);
    } else if (big && getenv("TACO_GEMM_W8")) {     // (experiment) eight waves per 128 x 128 tile
        if (a.pool) hipLaunchKernelGGL((conv_gemm_kernel<128, 128, true, 8>), cg_grid(a, 128, 128, z), dim3(512), 0, s, a);
        else hipLaunchKernelGGL((conv_gemm_kernel<128, 128, false, 8>), cg_grid(a, 128, 128, z), dim3(512), 0, s, a);
    } else if (big) {
        if (a.pool) hipLaunchKernelGGL((conv_gemm_kernel<128, 128, true>), cg_grid(a, 128, 128, z), dim3(256), 0, s, a);
        else hipLaunchKernelGGL((conv_gemm_kernel<128, 128, false>), cg_grid(a, 128, 128, z), dim3(256), 0, s, a);
    } else {
        if (a.pool) hipLaunchKernelGGL((conv_gemm_kernel<64, 64, true>), cg_grid(a, 64, 64, z), dim3(256), 0, s, a);
        else hipLaunchKernelGGL((conv_gemm_kernel<64, 64, false>), cg_grid(a, 64, 64, z), dim3(256), 0, s, a);
        if (a.splitk > 1) {
            const int ldp = (a.N + 63) / 64 * 64;
            long n4 = (long)a.M * (ldp / 4);
            int blocks = (int)std::min<long>((n4 + 255) / 256, 2048);
            hipLaunchKernelGGL(splitk_epilogue_kernel, dim3(blocks), dim3(256), 0, s, a, ldp);
        }
    }
    if (tm) tm->end(KK_CONV_GEMM, s);
}

// ---------------------------------------------------------------------------------------
// cbhg_tail: highway x n_hw + BiGRU input projections in one launch (kernels.h). BM rows per workgroup; the four waves
// split the 256 output columns of a layer (64 each = one [32 H | 32 T] block, so a lane holds H and T of its unit), every
// wave takes all BM rows. The activations [BM][128] stay in LDS (row stride 132: the 16 rows of a ds_read_b128 group
// fall on 16 different bank quads) and are the A operand of every layer; the weights stream through a double-buffered
// [16][256] LDS tile exactly as conv_gemm's B operand does, one continuous pipeline over all layers (the first chunk of
// a layer is in flight under the last MFMAs of the layer before). k order within a chunk as in conv_gemm (lane half lk
// feeds k = 8 lk + kk to MFMA kk), so a row's sums do not depend on BM: any tile choice gives the same bits.
template <int BM>
__global__ __launch_bounds__(256, 2) void cbhg_tail_kernel(CbhgTailArgs a) {
    constexpr int D = 128, BK = 16, XS = D + 4, BN = 256, PAD = 4, TM = BM / 32, NKT = D / BK, NB = BK * BN / 4 / 256;
    extern __shared__ __attribute__((aligned(16))) float tail_smem[];
    float (*xs)[XS] = reinterpret_cast<float (*)[XS]>(tail_smem);
    float (*Bs)[BK][BN + PAD] = reinterpret_cast<float (*)[BK][BN + PAD]>(tail_smem + BM * XS);
    typedef float f32x4v __attribute__((ext_vector_type(4)));
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lk = lane >> 5, li = lane & 31;
    const int m0 = blockIdx.x * BM, wn = 64 * wave;

    // "layers": n_hw highway layers, then Nr / 256 column chunks of the input projection
    const int nrc = (a.Nr + BN - 1) / BN;
    const int nl = a.n_hw + nrc;
    // Base address of every weight chunk (16 k-rows of a layer's 256-column block), built once in LDS: the chunk loop
    // then has no branches (a single basic block lets the compiler count its waits: the chunk two ahead stays in flight
    // while the chunk one ahead is written to LDS) and does not index the kernel-argument pointer arrays with a
    // run-time layer number, which costs a select chain over all eight entries per fetch. The loads go through a
    // buffer resource made from the table entry, so they are plain vector-memory loads whatever the compiler knows
    // about the pointer's origin.
    const int nchunks = nl * NKT;
    unsigned long long* ctab_s = reinterpret_cast<unsigned long long*>(tail_smem + BM * XS + 2 * BK * (BN + PAD));   // [12 * NKT]
    for (int c = tid; c < nchunks; c += 256) {
        const int L = c / NKT, kt = c % NKT;
        const float* W;
        int ldw;
        if (L < a.n_hw) { ldw = BN; W = a.Whw[L]; }
        else { ldw = a.ldwr; W = a.Wr + (L - a.n_hw) * BN; }
        ctab_s[c] = reinterpret_cast<unsigned long long>(W + (size_t)(kt * BK) * ldw);
    }
    __syncthreads();
    // weight chunks are fetched TWO chunks ahead (two register sets, even / odd chunks): with one workgroup's 16 MFMAs
    // per chunk (BM = 32) a single chunk of run-ahead does not cover an L2 round trip
    f32x4v rbA[NB], rbB[NB];
    const int hw_chunks = a.n_hw * NKT;
    auto fetch = [&](f32x4v (&rb)[NB], int c) {
        const int cc = min(c, nchunks - 1);                 // (past the end: the last chunk again, never used)
        const unsigned long long base = ctab_s[cc];
        const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)base);
        const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(base >> 32));
        const int ldw = cc < hw_chunks ? BN : a.ldwr;
        const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc(
            reinterpret_cast<void*>(((unsigned long long)hi << 32) | lo), 0, ((BK - 1) * ldw + BN) * 4, 0x00020000);
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const int f = tid + 256 * i;
            const int k = f / (BN / 4), nq = f % (BN / 4);
            rb[i] = __builtin_bit_cast(f32x4v, __builtin_amdgcn_raw_buffer_load_b128(wr, (unsigned)(k * ldw + 4 * nq) * 4u, 0, 0));
        }
    };
    auto stash = [&](const f32x4v (&rb)[NB], int buf) {
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const int f = tid + 256 * i;
            const int k = f / (BN / 4), nq = f % (BN / 4);
            *reinterpret_cast<f32x4v*>(&Bs[buf][k][4 * nq]) = rb[i];
        }
    };
    fetch(rbA, 0);
    fetch(rbB, 1);
    // the rows' activations (zero rows past M)
    for (int f = tid; f < BM * (D / 4); f += 256) {
        const int row = f / (D / 4), c4 = f % (D / 4);
        const int m = m0 + row;
        f32x4v v = {0.f, 0.f, 0.f, 0.f};
        if (m < a.M) v = *reinterpret_cast<const f32x4v*>(a.X + (size_t)m * a.ldx + 4 * c4);
        *reinterpret_cast<f32x4v*>(&xs[row][4 * c4]) = v;
    }
    stash(rbA, 0);
    __syncthreads();
    fetch(rbA, 2);
    for (int L = 0; L < nl; ++L) {
        f32x16 acc[TM][2];
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
        auto compute = [&](int kt, int buf) {
            float af[BK / 2][TM], bf[BK / 2][2];
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const f32x4v a0 = *reinterpret_cast<const f32x4v*>(&xs[32 * i + li][BK * kt + 8 * lk]);
                const f32x4v a1 = *reinterpret_cast<const f32x4v*>(&xs[32 * i + li][BK * kt + 8 * lk + 4]);
                af[0][i] = a0.x; af[1][i] = a0.y; af[2][i] = a0.z; af[3][i] = a0.w;
                af[4][i] = a1.x; af[5][i] = a1.y; af[6][i] = a1.z; af[7][i] = a1.w;
            }
#pragma unroll
            for (int kk = 0; kk < BK / 2; ++kk)
#pragma unroll
                for (int j = 0; j < 2; ++j) bf[kk][j] = Bs[buf][8 * lk + kk][wn + 32 * j + li];
#pragma unroll
            for (int kk = 0; kk < BK / 2; ++kk)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[kk][i], bf[kk][j], acc[i][j], 0, 0, 0);
        };
        static_assert(NKT % 2 == 0, "the chunk loop is unrolled by two");
        for (int kt = 0; kt < NKT; kt += 2) {
            const int c = L * NKT + kt;          // even chunk: lives in Bs[0], the next (odd) one in rbB
            compute(kt, 0);
            stash(rbB, 1);
            __syncthreads();
            fetch(rbB, c + 3);
            compute(kt + 1, 1);
            stash(rbA, 0);
            __syncthreads();
            fetch(rbA, c + 4);
        }
        if (L < a.n_hw) {
            // highway (modules.py:100-123): y = relu(H) * sigmoid(T) + x * (1 - sigmoid(T)), back into the LDS tile.
            // Every wave is past the barrier that follows the layer's last MFMAs: nobody reads the old x any more.
            const float* bias = a.bhw[L];
            const float bh = bias[wn + li], bt = bias[wn + 32 + li];
            const int unit = wn / 2 + li;
#pragma unroll
            for (int i = 0; i < TM; ++i) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = 32 * i + (r & 3) + 8 * (r >> 2) + 4 * lk;
                    const float hh = fmaxf(acc[i][0][r] + bh, 0.f);
                    const float tt = 1.f / (1.f + expf(-(acc[i][1][r] + bt)));
                    const float x = xs[row][unit];
                    xs[row][unit] = hh * tt + x * (1.f - tt);
                }
            }
            __syncthreads();
        } else {
            const int nc = (L - a.n_hw) * BN;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int n = nc + wn + 32 * j + li;
                if (n >= a.Nr) continue;
                const float bv = a.br ? a.br[n] : 0.f;
#pragma unroll
                for (int i = 0; i < TM; ++i) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int m = m0 + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * lk;
                        if (m < a.M) a.out[(size_t)m * a.ldo + n] = acc[i][j][r] + bv;
                    }
                }
            }
        }
    }
}

void launch_cbhg_tail(const CbhgTailArgs& a, hipStream_t s, Timing* tm) {
    if (tm) tm->begin(KK_CONV_GEMM, s);
    static bool attr_set[64] = {};   // more than the default 64 KB of dynamic LDS: the attribute is per device
    int dev = 0;
    if (hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < 64 && !attr_set[dev]) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(cbhg_tail_kernel<32>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(cbhg_tail_kernel<64>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_set[dev] = true;
    }
    // rows per workgroup: 64 when that still gives a workgroup per CU and more, else 32 (any choice gives the same bits)
    const bool big = (a.M + 63) / 64 >= 256;
    const int bm = big ? 64 : 32;
    const size_t lds = ((size_t)bm * 132 + 2 * 16 * 260) * sizeof(float) + 12 * 8 * sizeof(unsigned long long);
    const dim3 grid((a.M + bm - 1) / bm);
    if (big) hipLaunchKernelGGL((cbhg_tail_kernel<64>), grid, dim3(256), lds, s, a);
    else hipLaunchKernelGGL((cbhg_tail_kernel<32>), grid, dim3(256), lds, s, a);
    if (tm) tm->end(KK_CONV_GEMM, s);
}

// ---------------------------------------------------------------------------------------
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

// Max-pool over time, stride 1, 'same' padding, any window (the reference's width 2 never comes here: it is folded into
// proj_1's operand loader). One thread per four channels of a row; the padding does not take part in the max.
__global__ void maxpool_same_kernel(const float* __restrict__ x, float* __restrict__ out, int M, int T, int ld4,
                                    int width) {
    const long total = (long)M * ld4;
    const int pl = (width - 1) / 2, pr = width - 1 - pl;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int m = (int)(i / ld4), c4 = (int)(i - (long)m * ld4);
        const int t = m % T;
        const int lo = max(t - pl, 0), hi = min(t + pr, T - 1);
        const float4* row = reinterpret_cast<const float4*>(x) + (long)(m - t) * ld4 + c4;
        float4 v = row[(long)lo * ld4];
        for (int u = lo + 1; u <= hi; ++u) {
            const float4 w = row[(long)u * ld4];
            v.x = fmaxf(v.x, w.x); v.y = fmaxf(v.y, w.y); v.z = fmaxf(v.z, w.z); v.w = fmaxf(v.w, w.w);
        }
        reinterpret_cast<float4*>(out)[i] = v;
    }
}

void launch_maxpool_same(const float* x, float* out, int M, int T, int ld, int width, hipStream_t s, Timing* tm) {
    if (tm) tm->begin(KK_CONV_GEMM, s);
    const long total = (long)M * (ld / 4);
    const int blocks = (int)std::max<long>(1, std::min<long>(8192, (total + 255) / 256));
    hipLaunchKernelGGL(maxpool_same_kernel, dim3(blocks), dim3(256), 0, s, x, out, M, T, ld / 4, width);
    if (tm) tm->end(KK_CONV_GEMM, s);
}

}  // namespace taco
