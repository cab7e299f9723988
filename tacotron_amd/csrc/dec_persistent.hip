// dec_persistent: the whole dynamic_decode loop (model/decoder.py:18-40, model/rnn_wrappers.py,
// model/helpers.py:66-78; step maths SURVEY.md A.7) as ONE launch for the default layer sizes
// (prenet 256/128, attention/decoder depth 256, 80 mels x r=5, 256-wide encoder memory).
//
// Why: a decoder step is a chain of dependent all-to-all stages with ~0.5 us of arithmetic each; as separate
// kernels every stage pays a dispatch boundary and re-streams its weights.  Here
//   * each b-tile of 16 utterances is owned by a group of 32 workgroups (one per CU, 256 threads = one wave per
//     SIMD with the whole 512-entry register file); a workgroup computes the same 16-row output tile of its
//     stages for all S steps, so its slices of the decoder weights (58 quads per lane) stay in registers for the
//     whole decode: per step no weight byte moves;
//   * the GRU state / residual tiles a workgroup owns stay in registers as well;
//   * the encoder memory of an utterance is staged once into the LDS of the two workgroups that share its
//     attention (keys: each half of T_in; values: each half of the 256 channels), when it fits (T_in <= 150 at the
//     default sizes; otherwise it is read from L2);
//   * stages hand activations over as 8-byte {value, tag} granules (below): placement-independent, no reliance
//     on dispatch order or XCD sharing;
//   * two linear layers are folded into their consumers' weights at load time (api.hip Loader::dec_fold) and so
//     leave the dependency chain: the r-frame output projection (next step's prenet-1 reads o2 directly; the
//     frames themselves are computed one stage later, off the chain) and the concat projection (GRU-1 reads
//     [h_att ; ctx] directly; d0, needed only for the residual, is computed locally by the workgroup that owns the
//     matching GRU-1 tile). 11 hand-offs per step instead of 13;
//   * every spin is bounded: on timeout the launch's abort word is set, all waits fall through and the host
//     reports the error (taco_decode_status).
// Reductions have a fixed order of their own (K split over 4 waves, two workgroups per utterance in the
// attention): results are reproducible and batch-independent, and differ from the per-stage kernels
// (decoder.hip, K over 8 waves, unfolded weights) only by fp32 rounding.
#include "kernels.h"
#include "dec_common.h"

namespace taco {

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

#ifndef PD_AUX_LD        // cache policy of the hand-off granules (experiments): 16 = sc1, 1 = sc0, 17 = both, 0 = none
#define PD_AUX_LD 16
#endif
#ifndef PD_AUX_ST
#define PD_AUX_ST 16
#endif
#ifdef PD_INV_L1         // (experiment) polls = invalidate the CU's vector L1, then plain loads: L2 hits within an XCD
#ifndef PD_INV_SCOPE
#define PD_INV_SCOPE sc0
#endif
#define PD_STR2(x) #x
#define PD_STR(x) PD_STR2(x)
#define PD_POLL_INV() asm volatile("buffer_inv " PD_STR(PD_INV_SCOPE) ::: "memory")
#else
#define PD_POLL_INV() do {} while (0)
#endif
#ifndef PD_SLEEP         // s_sleep argument after a failed poll (64 clocks each)
#define PD_SLEEP 1
#endif
constexpr int PD_THREADS = 256;   // one wave per SIMD: each wave may use the whole 512-entry register file
constexpr int PD_WAVES = 4;
constexpr unsigned long long PD_WAIT_LIMIT_TICKS = 1000000000ull;   // 10 s of s_memrealtime (100 MHz) per wait
constexpr int PD_BATCH = 4;                    // k16-steps fetched (and tag-checked) together
constexpr int PD_STAGES = 13;
#ifndef PD_STAMP_WG
#define PD_STAMP_WG 0
#endif                  // hand-off points per decoder step

enum { ST_PRE1 = 0, ST_PRE2, ST_ATTG, ST_ATTC, ST_QUERY, ST_PROJ, ST_G1G, ST_G1C, ST_G2G, ST_G2C, ST_OUT, ST_COUNT };

// ---- granule transport ---------------------------------------------------------------------------------
// Every value a workgroup hands to others travels as an 8-byte {value, tag} granule written by a write-through
// (sc1) store and read by sc1 loads (L1-bypassing): the data is its own flag, so a hand-off costs one
// store -> load hop and no separate counter, fence or barrier. tag = 13*step + producing stage + 1 (never 0);
// buffers are zero-filled before the launch, so initial state carries tag 0. A Q-layout quad {x[k..k+3]} of
// batch column b is the 32 bytes {x0,tag,x1,tag,x2,tag,x3,tag} at float offset ((k/4)*Bpad + b)*8, moved as two
// 16-byte accesses whose 8-byte halves are each one granule.
struct GQuad { u32x4 lo, hi; };
__device__ __forceinline__ GQuad pd_ldg(__amdgpu_buffer_rsrc_t rs, unsigned float_off) {
    GQuad g;
    g.lo = __builtin_amdgcn_raw_buffer_load_b128(rs, float_off * 4u, 0, PD_AUX_LD /* sc1 */);
    g.hi = __builtin_amdgcn_raw_buffer_load_b128(rs, float_off * 4u + 16u, 0, PD_AUX_LD);
    return g;
}
// same with the address split into a per-lane byte offset (one VGPR for the whole kernel) and a wave-uniform
// byte offset (SGPR): the K loop of a stage then needs no address registers at all
__device__ __forceinline__ GQuad pd_ldg2(__amdgpu_buffer_rsrc_t rs, unsigned lane_bytes, unsigned uni_bytes) {
    GQuad g;
    g.lo = __builtin_amdgcn_raw_buffer_load_b128(rs, lane_bytes, uni_bytes, PD_AUX_LD /* sc1 */);
    g.hi = __builtin_amdgcn_raw_buffer_load_b128(rs, lane_bytes + 16u, uni_bytes, PD_AUX_LD);
    return g;
}
__device__ __forceinline__ bool gq_ok(const GQuad& g, unsigned tag) {
    return g.lo.y == tag && g.lo.w == tag && g.hi.y == tag && g.hi.w == tag;
}
__device__ __forceinline__ f32x4 gq_val(const GQuad& g) {
    // (element-wise __builtin_bit_cast on ext_vector members miscompiles with this hipcc: use the intrinsics)
    const unsigned a = g.lo.x, b = g.lo.z, c = g.hi.x, d = g.hi.z;
    return f32x4{__uint_as_float(a), __uint_as_float(b), __uint_as_float(c), __uint_as_float(d)};
}
// (`on` = this lane's batch column is a real utterance: the columns that pad the last tile are neither published nor
// polled, so a small batch moves proportionally fewer granules per hand-off)
__device__ __forceinline__ void pd_stg(__amdgpu_buffer_rsrc_t rs, unsigned float_off, f32x4 v, unsigned tag,
                                       bool on = true) {
    if (!on) return;
    const float a = v.x, b = v.y, c = v.z, d = v.w;
    const u32x4 lo = {__float_as_uint(a), tag, __float_as_uint(b), tag};
    const u32x4 hi = {__float_as_uint(c), tag, __float_as_uint(d), tag};
    __builtin_amdgcn_raw_buffer_store_b128(lo, rs, float_off * 4u, 0, PD_AUX_ST /* sc1 */);
    __builtin_amdgcn_raw_buffer_store_b128(hi, rs, float_off * 4u + 16u, 0, PD_AUX_ST);
}

// Bounded wait bookkeeping: a wave whose single wait has lasted PD_WAIT_LIMIT_TICKS of the 100 MHz real-time clock
// sets the launch's abort word and goes on with what it has; every wave looks at the abort word (and the clock) every
// 1024 failed polls and stops waiting once it is set. The bound is on wall time per wait, not on the number of polls:
// a launch that shares the GPU with other queues can be time-sliced and run orders of magnitude slower for a while
// (seen on the first launch of a new stream beside a busy one) without anything being wrong.
struct Spin {
    unsigned* abort_word;      // [0] abort flag, [1] workgroup and [2] tag of the first wait that ran out (diagnostics)
    unsigned n;                // failed polls of the current wait
    bool dead;
    unsigned where;            // tag the current wait is for
    unsigned long long t0;
#ifdef PD_STAMPS
    unsigned long long* st_lds; int stamp_i; bool stamp_on;    // diagnostic builds: stamps collected in LDS
    __device__ __forceinline__ void stamp(unsigned long long flag) {
        if (stamp_on && stamp_i < 250) st_lds[stamp_i++] = __builtin_amdgcn_s_memrealtime() | (flag << 60);
    }
#endif
    __device__ __forceinline__ void begin(unsigned tag) { where = tag; n = 0u; }
    __device__ __forceinline__ bool give_up() {
        if (dead) return true;
        ++n;
        if ((n & 1023u) == 0u) {
            const unsigned long long now = __builtin_amdgcn_s_memrealtime();
            if (n == 1024u) t0 = now;
            if (__hip_atomic_load(abort_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) dead = true;
            else if (now - t0 > PD_WAIT_LIMIT_TICKS) {
                if (__hip_atomic_exchange(abort_word, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) {
                    __hip_atomic_store(abort_word + 1, (unsigned)blockIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_store(abort_word + 2, where, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    // sticky word of the handle (PD_SYNC_STICKY, not cleared per launch): launches of this handle that
                    // are already queued behind this one leave at once instead of running into the same wait
                    __hip_atomic_store(abort_word + (PD_SYNC_STICKY - PD_SYNC_ABORT), 1u, __ATOMIC_RELAXED,
                                       __HIP_MEMORY_SCOPE_AGENT);
                }
                dead = true;
            }
        }
        __builtin_amdgcn_s_sleep(PD_SLEEP);
        return dead;
    }
};

// One stage's matrix product for this workgroup's 16x16 tile: wave w owns k16-steps w, w+4, w+8, ... (CH of them), so
// every wave has an equal share of each K segment; they are fetched (and re-fetched until every granule carries the
// expected tag) PD_BATCH steps at a time, the segments in K order: the stage's late input is its last segment. Wave 0 can fetch up to
// two more quads of its epilogue (update gate, residual) in the same poll loop of its first batch.
// Returns the reduced tile fragment in wave 0 (lane: column b = lane&15, rows 4*(lane>>4)..+3).
struct Extra { unsigned off, tag; f32x4 val; };
// the K dimension of a stage is up to three consecutive segments of k16-steps, each its own granule buffer and tag
struct Segs { unsigned off0, tag0; int e0; unsigned off1, tag1; int e1; unsigned off2, tag2; };
template <int CH, int NX, int E0 = PD_BATCH>   // E0: size of the first batch (so that later batches line up with a segment)
__device__ __forceinline__ f32x4 pd_mma(const float4 (&w)[CH], __amdgpu_buffer_rsrc_t rs, const Segs sg, int nk16,
                                        int Bpad, int lb, int lq, int wave_v, int lane, f32x4* red, Spin& spin,
                                        Extra* ex, bool col_on) {
    const int wave = __builtin_amdgcn_readfirstlane(wave_v);              // provably uniform -> scalar addressing
    const unsigned lane_bytes = (unsigned)(lq * Bpad + lb) * 32u;
    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
    spin.begin(sg.e1 < nk16 ? sg.tag2 : (sg.e0 < nk16 ? sg.tag1 : sg.tag0));   // (the stage's last, late segment)
#pragma unroll
    for (int hb = 0; hb < CH + PD_BATCH; hb += PD_BATCH) {
        // batches [0, E0), [E0, E0 + PD_BATCH), ...
        const int h = hb == 0 ? 0 : hb - PD_BATCH + E0;
        const int hn = hb == 0 ? E0 : PD_BATCH;
        if (h >= CH) break;
        const bool last = h + hn >= CH;      // the epilogue operands (update gate, residual) ride with the last batch:
                                             // a late one must not hold back K segments that are already there
        GQuad g[PD_BATCH] = {};              // (padding columns stay zero: never loaded, never waited for)
        GQuad gx[NX > 0 ? NX : 1] = {};
        // a k16-step that has arrived is not fetched again: a retry only re-reads what is still missing
        bool have[PD_BATCH], havex[NX > 0 ? NX : 1];
#pragma unroll
        for (int i = 0; i < PD_BATCH; ++i) have[i] = false;
#pragma unroll
        for (int i = 0; i < (NX > 0 ? NX : 1); ++i) havex[i] = false;
        while (true) {
            bool ok = true;
            PD_POLL_INV();
#pragma unroll
            for (int i = 0; i < PD_BATCH; ++i) {
                const int ks = (h + i) * PD_WAVES + wave;
                if (i < hn && h + i < CH && ks < nk16 && !have[i]) {
                    const int s = ks < sg.e0 ? 0 : (ks < sg.e1 ? 1 : 2);
                    const unsigned off = s == 0 ? sg.off0 : (s == 1 ? sg.off1 : sg.off2);
                    const int kb = s == 0 ? 0 : (s == 1 ? sg.e0 : sg.e1);
                    const unsigned uni = (off + (unsigned)(4 * (ks - kb) * Bpad) * 8u) * 4u;
                    if (col_on) g[i] = pd_ldg2(rs, lane_bytes, uni);
                }
            }
            if (NX > 0 && last && wave == 0) {
#pragma unroll
                for (int i = 0; i < NX; ++i)
                    if (!havex[i] && col_on) gx[i] = pd_ldg(rs, ex[i].off);
            }
#pragma unroll
            for (int i = 0; i < PD_BATCH; ++i) {
                const int ks = (h + i) * PD_WAVES + wave;
                if (i < hn && h + i < CH && ks < nk16 && !have[i]) {
                    const int s = ks < sg.e0 ? 0 : (ks < sg.e1 ? 1 : 2);
                    const unsigned tag = s == 0 ? sg.tag0 : (s == 1 ? sg.tag1 : sg.tag2);
                    have[i] = __all(!col_on || gq_ok(g[i], tag));
                    ok = ok && have[i];
                }
            }
            if (NX > 0 && last && wave == 0) {
#pragma unroll
                for (int i = 0; i < NX; ++i)
                    if (!havex[i]) {
                        havex[i] = __all(!col_on || gq_ok(gx[i], ex[i].tag));
                        ok = ok && havex[i];
                    }
            }
            if (ok || spin.give_up()) break;
        }
#ifdef PD_STAMPS
        if (wave_v == 0 && lane == 0) spin.stamp(1);
#endif
        if (NX > 0 && last && wave == 0) {
#pragma unroll
            for (int i = 0; i < NX; ++i) ex[i].val = gq_val(gx[i]);
        }
#pragma unroll
        for (int i = 0; i < PD_BATCH; ++i) {
            const int ks = (h + i) * PD_WAVES + wave;
            if (i < hn && h + i < CH && ks < nk16) {
                const f32x4 x = gq_val(g[i]);
                acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(w[h + i].x, x.x, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(w[h + i].y, x.y, acc1, 0, 0, 0);
                acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(w[h + i].z, x.z, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(w[h + i].w, x.w, acc1, 0, 0, 0);
            }
        }
    }
#ifdef PD_STAMPS
    if (wave_v == 0 && lane == 0) spin.stamp(2);
#endif
    __syncthreads();                       // previous readers of `red` are done
    red[wave * 64 + lane] = acc0 + acc1;
    __syncthreads();
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (wave == 0) {
        v = red[lane];
#pragma unroll
        for (int ww = 1; ww < PD_WAVES; ++ww) v += red[ww * 64 + lane];
    }
#ifdef PD_STAMPS
    if (wave_v == 0 && lane == 0) spin.stamp(3);
#endif
    return v;
}

// one quad with a tag wait
__device__ __forceinline__ f32x4 pd_wait_quad(__amdgpu_buffer_rsrc_t rs, unsigned float_off, unsigned tag, Spin& spin) {
    GQuad g;
    spin.begin(tag);
    while (true) {
        PD_POLL_INV();
        g = pd_ldg(rs, float_off);
        if (__all(gq_ok(g, tag)) || spin.give_up()) break;
    }
    return gq_val(g);
}

template <int CH>
__device__ __forceinline__ void pd_load_w(float4 (&w)[CH], const float* Wp, int tile, int ntiles, int nk16, int wave,
                                          int lane) {
#pragma unroll
    for (int i = 0; i < CH; ++i) {
        const int ks = i * PD_WAVES + wave;
        w[i] = (tile < ntiles && ks < nk16)
                   ? reinterpret_cast<const float4*>(Wp)[((size_t)tile * nk16 + ks) * 64 + lane]
                   : make_float4(0.f, 0.f, 0.f, 0.f);
    }
}

// gate non-linearities on the hardware exp2 / rcp (absolute error ~2e-7, like fast_tanh): the epilogue sits on the
// step's dependency chain, the library expf + IEEE divide are ~4x the instructions
__device__ __forceinline__ float fast_sigmoid(float x) {
    return __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(x * -1.4426950408889634f));
}
__device__ __forceinline__ f32x4 sig4(f32x4 v) {
    return f32x4{fast_sigmoid(v.x), fast_sigmoid(v.y), fast_sigmoid(v.z), fast_sigmoid(v.w)};
}
__device__ __forceinline__ f32x4 tanh4(f32x4 v) { return f32x4{fast_tanh(v.x), fast_tanh(v.y), fast_tanh(v.z), fast_tanh(v.w)}; }
__device__ __forceinline__ f32x4 relu4(f32x4 v) {
    return f32x4{fmaxf(v.x, 0.f), fmaxf(v.y, 0.f), fmaxf(v.z, 0.f), fmaxf(v.w, 0.f)};
}

// workgroups per utterance for the attention of a tile that holds n (1..16) utterances: 32 / (n rounded up to a power
// of two), at least 2
__host__ __device__ inline int pd_att_parts(int n) {
    return n <= 1 ? 32 : (n <= 2 ? 16 : (n <= 4 ? 8 : (n <= 8 ? 4 : 2)));
}

template <int N, int CH>
__device__ __forceinline__ float4 (&pd_sub(float4 (&w)[CH], int first))[N] {
    return *reinterpret_cast<float4(*)[N]>(&w[first]);
}

__global__ __launch_bounds__(PD_THREADS) void dec_persistent_kernel(DecPersistArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    // LDS carve (all offsets multiples of 16)
    f32x4* red = reinterpret_cast<f32x4*>(smem);                         // [4][64] f32x4 (also ctx partials), 8 KB reserved
    float* bias_s = reinterpret_cast<float*>(smem + 8192);               // [ST_COUNT][16] reserved, SL_COUNT slots used
    float* e_s = bias_s + ST_COUNT * 16;                                 // [T_in rounded to 4] scores
    const int T4 = (a.T_in + 3) & ~3;
    float* p_s = e_s + T4;                                               // [T_in rounded to 4] exp(score - max)
    float* wsum_s = p_s + T4;                                            // [4] per-wave partial sums, [4..7] key range flags
    float* keys_s = wsum_s + 8;                                          // [Tp][256] when staged

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // (placement experiment, TACO_PD_XCD_PACK: a one-tile launch of 256 workgroups of which only those dealt to one XCD,
    // every eighth, take a role; the hand-offs are placement-independent either way)
    if (a.xcd_pack && (blockIdx.x & 7)) return;
    const int wid = a.xcd_pack ? (int)(blockIdx.x >> 3) : (int)blockIdx.x;
    const int bt = wid >> 5, j = wid & 31;
#ifdef PD_PHASE_STAMPS   // diagnostic build: when did workgroup 0 start, finish step 0 (everybody resident), finish
    if (blockIdx.x == 0 && tid == 0) reinterpret_cast<unsigned long long*>(a.sync)[8] = __builtin_amdgcn_s_memrealtime();
#endif
    {
        // census of the launch (how many workgroups ever started: names an over-subscribed device in
        // taco_decode_status) and the handle's sticky abort word: set by a launch whose bounded wait ran out, it
        // makes every later persistent launch of the handle return before it waits for anything
        unsigned* sw = reinterpret_cast<unsigned*>(a.sync);
        unsigned* flag_s = reinterpret_cast<unsigned*>(smem);       // (one decision per workgroup)
        if (tid == 0) {
            __hip_atomic_fetch_add(sw + PD_SYNC_ARRIVED, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned sticky = __hip_atomic_load(sw + PD_SYNC_STICKY, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (sticky != 0u &&
                __hip_atomic_exchange(sw + PD_SYNC_ABORT, 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) {
                __hip_atomic_store(sw + PD_SYNC_WG, (unsigned)blockIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(sw + PD_SYNC_TAG, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            *flag_s = sticky;
        }
        __syncthreads();
        const unsigned leave = *flag_s;
        __syncthreads();                                            // smem is `red` from here on
        if (leave != 0u) return;
    }
    const int b0 = bt * 16, Bpad = a.Bpad;
    const int lb = b0 + (lane & 15), lq = lane >> 4;
    const bool col_on = lb < a.B;            // this lane's batch column is a real utterance
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(a.region, 0, a.region_bytes, 0x00020000);
    Spin spin{reinterpret_cast<unsigned*>(a.sync) + PD_SYNC_ABORT, 0u, false, 0u, 0ull};
#ifdef PD_STAMPS
    spin.st_lds = reinterpret_cast<unsigned long long*>(smem + 4096); spin.stamp_i = 0; spin.stamp_on = false;
#endif

    // ---- roles. The 32 workgroups of a group split the stages between them so that each holds 58 weight quads per
    // lane (232 registers) and the rest of the register file keeps a stage's operands in flight:
    //   role A (j < 16, k = j): prenet-1 (folded: reads o2) tile k, query tile k, GRU-1 owner (folded reset-gate tile k,
    //        folded candidate tile k, owns h1 / o1), update-gate tiles of the attention GRU and of GRU-2, output tiles
    //        16..24 (k < 9);
    //   role B (j >= 16, k = j-16): prenet-2 tile k (k < 8), attention-GRU owner (reset tile k, candidate tile k, owns
    //        h_att), GRU-2 owner (reset tile k, candidate tile k, owns h2), update-gate tiles of GRU-1 (folded), the
    //        concat projection d0 tile k (only the residual o1 = h1 + d0 needs it: off the chain), output tiles 0..15.
    const bool roleA = j < 16;
    const int k = roleA ? j : j - 16;
    enum { SL_A6 = 0, SL_G12, SL_G8, SL_X8A, SL_X8B0, SL_X8B1, SL_X12, SL_O4, SL_GO, SL_COUNT };
    float4 a6[6], g12[12], g8[8], x8a[8], x8b[8], x12[12], o4[4];
    const int t_a6 = roleA ? 16 + k : k, t_g12 = roleA ? k : 16 + k, t_g8 = roleA ? 16 + k : k;
    const int t_out = roleA ? 16 + k : k;
    pd_load_w<6>(a6, a.Wp[ST_ATTG], t_a6, 32, 24, wave, lane);
    pd_load_w<12>(g12, a.Wp[ST_G1G], t_g12, 32, 48, wave, lane);
    pd_load_w<8>(g8, a.Wp[ST_G2G], t_g8, 32, 32, wave, lane);
    pd_load_w<8>(x8a, a.Wp[roleA ? ST_PRE1 : ST_G2C], k, 16, 32, wave, lane);
    if (roleA) {
        pd_load_w<4>(pd_sub<4>(x8b, 0), a.Wp[ST_QUERY], k, 16, 16, wave, lane);
#pragma unroll
        for (int i = 4; i < 8; ++i) x8b[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        pd_load_w<12>(x12, a.Wp[ST_G1C], k, 16, 48, wave, lane);
    } else {
        pd_load_w<8>(x8b, a.Wp[ST_PROJ], k, 16, 32, wave, lane);
        pd_load_w<6>(pd_sub<6>(x12, 0), a.Wp[ST_ATTC], k, 16, 24, wave, lane);
        pd_load_w<4>(pd_sub<4>(x12, 6), a.Wp[ST_PRE2], k, 8, 16, wave, lane);
#pragma unroll
        for (int i = 10; i < 12; ++i) x12[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    pd_load_w<4>(o4, a.Wp[ST_OUT], t_out, 25, 16, wave, lane);
    for (int i = tid; i < SL_COUNT * 16; i += PD_THREADS) {
        const int sl = i >> 4, r = i & 15;
        const float* bp = nullptr;
        int tile = k, nt = 16;
        switch (sl) {
            case SL_A6: bp = a.bias[ST_ATTG]; tile = t_a6; nt = 32; break;
            case SL_G12: bp = a.bias[ST_G1G]; tile = t_g12; nt = 32; break;
            case SL_G8: bp = a.bias[ST_G2G]; tile = t_g8; nt = 32; break;
            case SL_X8A: bp = a.bias[roleA ? ST_PRE1 : ST_G2C]; break;
            case SL_X8B0: bp = roleA ? nullptr : a.bias[ST_PROJ]; break;
            case SL_X8B1: bp = roleA ? nullptr : a.bias[ST_PRE2]; nt = 8; break;
            case SL_X12: bp = a.bias[roleA ? ST_G1C : ST_ATTC]; break;
            case SL_O4: bp = a.bias[ST_OUT]; tile = t_out; nt = 25; break;
            default: bp = roleA ? a.bias_pre1_go : nullptr; break;
        }
        bias_s[i] = (bp && tile < nt) ? bp[tile * 16 + r] : 0.f;
    }
    // ---- attention role: the 32 workgroups of a tile share its utterances' attention, P = 2 .. 32 workgroups per
    // utterance (2 for a full tile of 16, all 32 for a single utterance): part `part` of utterance u scores the key
    // rows [part Tp, (part+1) Tp) and builds 256/P channels of the context. What a workgroup stages in LDS shrinks
    // with P, so a small batch keeps long inputs' keys and values on chip too. The arithmetic per score row and per
    // context channel does not depend on P (see the context loop), so an utterance's result does not either.
    const int P = pd_att_parts(min(16, a.B - b0));
    const int u_loc = j / P, part = j - u_loc * P;
    const int b_att = b0 + u_loc;
    const bool att_on = b_att < a.B;
    const int Tp = (a.T_in + P - 1) / P;
    const int t_lo = min(part * Tp, a.T_in), t_n = min(Tp, a.T_in - t_lo);   // this part's score rows
    const int nq = 64 / P;                                                   // context channel quads of this part
    const int kl = min(t_n, a.lds_keys);                                     // key rows of this part held in LDS
    float* vals_s = keys_s + min(Tp, a.lds_keys) * 256;                      // [T_in][nq] float4 when staged
    const float* keys_g = a.keys + ((size_t)(att_on ? b_att : 0) * a.T_in + t_lo) * 256;
    const float* vals_g = a.values + (size_t)(att_on ? b_att : 0) * a.T_in * 256 + part * nq * 4;
    // Fast score path: tanh(k + q) = 1 - 2 / (E_k E_q + 1) with E_x = 2^(2 log2(e) x). E_k is stored instead of k when
    // every |k| of the UTTERANCE is below 40 (E_k finite and normal; a product that overflows or underflows still gives
    // the right +-1; the test is on the whole utterance, not on this part, so the choice does not depend on how the
    // attention is split). A step whose |q| reaches 40 reads the plain keys from global memory instead. As many of
    // the part's rows as fit are staged (E_k or k); the others are read from L2 each step, E_k computed on the fly.
    bool keys_exp = false;
    if (att_on) {
        {
            const float4* ku = reinterpret_cast<const float4*>(a.keys + (size_t)b_att * a.T_in * 256);
            float kmax = 0.f;
            for (int i = tid; i < a.T_in * 64; i += PD_THREADS) {
                const float4 kv = ku[i];
                kmax = fmaxf(fmaxf(kmax, fmaxf(fabsf(kv.x), fabsf(kv.y))), fmaxf(fabsf(kv.z), fabsf(kv.w)));
            }
            kmax = wave_max(kmax);
            if (lane == 0) wsum_s[4 + wave] = kmax;
            __syncthreads();
            kmax = fmaxf(fmaxf(wsum_s[4], wsum_s[5]), fmaxf(wsum_s[6], wsum_s[7]));
            keys_exp = kmax < 40.f;      // (NaN keys fail the test and keep the plain path)
            constexpr float c2 = 2.8853900817779268f;
            for (int i = tid; i < kl * 64; i += PD_THREADS) {
                float4 kv = reinterpret_cast<const float4*>(keys_g)[i];
                if (keys_exp)
                    kv = make_float4(__builtin_amdgcn_exp2f(kv.x * c2), __builtin_amdgcn_exp2f(kv.y * c2),
                                     __builtin_amdgcn_exp2f(kv.z * c2), __builtin_amdgcn_exp2f(kv.w * c2));
                reinterpret_cast<float4*>(keys_s)[i] = kv;
            }
        }
        if (a.lds_vals)
            for (int i = tid; i < a.T_in * nq; i += PD_THREADS)
                reinterpret_cast<float4*>(vals_s)[i] =
                    reinterpret_cast<const float4*>(vals_g + (size_t)(i / nq) * 256)[i % nq];
    }
    const float4 v_att = reinterpret_cast<const float4*>(a.attn_v)[lane];
    __syncthreads();

    // tiles this workgroup owns across steps (wave 0 only uses them): role A h1, d0, o1; role B h_att, h2
    f32x4 hA_t = {0.f, 0.f, 0.f, 0.f}, hB_t = hA_t, o1_t = hA_t;   // A: hA=h1 ; B: hA=h_att, hB=h2
    const unsigned te = (unsigned)((k * 4 + lq) * Bpad + lb) * 8u;          // float offset of this lane's quad in tile k
    const unsigned te16 = (unsigned)(((16 + k) * 4 + lq) * Bpad + lb) * 8u; // ... in tile 16+k (output frames)
    const unsigned st_stride = (unsigned)(256 * Bpad * 2);                   // one 256-wide granule buffer
#ifdef PD_STAMPS
#define PD_STAMP() do { spin.stamp_on = (blockIdx.x == PD_STAMP_WG && s == 5); if (tid == 0) spin.stamp(0); } while (0)
#else
#define PD_STAMP() do {} while (0)
#endif
#define PD_BIAS(sl) (*reinterpret_cast<const f32x4*>(bias_s + (sl) * 16 + 4 * lq))
#define PD_MMA(CH, NX, W, SG, nk, EX) pd_mma<CH, NX>(W, rs, SG, nk, Bpad, lb, lq, wave, lane, red, spin, EX, col_on)
    // the r output frames of step `so` (decoder.py:35) from o2, which must carry tag `to2`
#define PD_OUT(so, to2)                                                                                              \
    do {                                                                                                             \
        if (!roleA || k < 9) {                                                                                       \
            const Segs sgo{a.o_o2, (to2), 16, 0u, 0u, 16, 0u, 0u};                                                   \
            f32x4 v = PD_MMA(4, 0, o4, sgo, 16, ex);                                                                 \
            if (wave == 0)                                                                                           \
                pd_stg(rs, a.o_outQ + (unsigned)((so) * 400) * Bpad * 2u + (roleA ? te16 : te), v + PD_BIAS(SL_O4),  \
                       (to2) + 1u, col_on);                                                                          \
        }                                                                                                            \
    } while (0)

    for (int s = 0; s < a.S; ++s) {
        const unsigned T = (unsigned)(PD_STAGES * s);                 // tag of stage i of this step = T + i + 1
        const unsigned Tp = s ? T - PD_STAGES : 0u;                   // previous step (tags below are 0 at s == 0)
        const unsigned cur = (unsigned)(s & 1) * st_stride, prv = (unsigned)((s & 1) ^ 1) * st_stride;
        // [h_att ; ctx] of a step are one 512-wide buffer, double-buffered by step parity
        const unsigned hc = a.o_hc + (unsigned)(s & 1) * 2u * st_stride, hcp = a.o_hc + (unsigned)((s & 1) ^ 1) * 2u * st_stride;
        const unsigned o_ctx = hc + st_stride, o_ctxp = hcp + st_stride;
        Extra ex[2];
        PD_STAMP();
        // ---- stage 0/1: decoder prenet (rnn_wrappers.py:25): relu(W1 [prev frame; ctx] + b1) with the output projection
        // folded in (prev frame = last of the r frames of o2), relu(W2 . + b2); step 0 feeds the <GO> frame: o2 = 0, own bias
        if (roleA) {
            const Segs sg{o_ctxp, s ? Tp + 7u : 0u, 16, a.o_o2, s ? Tp + 12u : 0u, 32, 0u, 0u};
            f32x4 v = PD_MMA(8, 0, x8a, sg, 32, ex);
            if (wave == 0) pd_stg(rs, a.o_p1 + te, relu4(v + PD_BIAS(s ? SL_X8A : SL_GO)), T + 1u, col_on);
        }
        if (s) PD_OUT(s - 1, Tp + 12u);       // previous step's frames: off the dependency chain
        if (!roleA && k < 8) {
            const Segs sg{a.o_p1, T + 1u, 16, 0u, 0u, 16, 0u, 0u};
            f32x4 v = PD_MMA(4, 0, pd_sub<4>(x12, 6), sg, 16, ex);
            if (wave == 0) pd_stg(rs, a.o_p2 + te, relu4(v + PD_BIAS(SL_X8B1)), T + 2u, col_on);
        }
        PD_STAMP();
        // ---- stage 2/3: attention GRU (rnn_wrappers.py:26): gates (A: update tile, B: reset tile), candidate (B)
        {
            const Segs sg{hcp, s ? Tp + 4u : 0u, 16, a.o_p2, T + 2u, 24, 0u, 0u};
            f32x4 v = PD_MMA(6, 0, a6, sg, 24, ex);
            if (wave == 0) {
                const f32x4 g = sig4(v + PD_BIAS(SL_A6));
                if (roleA) pd_stg(rs, a.o_u + te, g, T + 3u, col_on);
                else pd_stg(rs, a.o_rh + te, g * hA_t, T + 3u, col_on);
            }
        }
        PD_STAMP();
        if (!roleA) {
            ex[0].off = a.o_u + te; ex[0].tag = T + 3u;
            const Segs sg{a.o_p2, T + 2u, 8, a.o_rh, T + 3u, 24, 0u, 0u};
            f32x4 v = pd_mma<6, 1, 2>(pd_sub<6>(x12, 0), rs, sg, 24, Bpad, lb, lq, wave, lane, red, spin, ex, col_on);
            if (wave == 0) {
                const f32x4 ug = ex[0].val;
                const f32x4 c = tanh4(v + PD_BIAS(SL_X12));
                hA_t = ug * hA_t + (1.f - ug) * c;
                pd_stg(rs, hc + te, hA_t, T + 4u, col_on);
            }
        }
        PD_STAMP();
        if (roleA) {
            // ---- stage 4: attention query (decoder.py:20)
            const Segs sq{hc, T + 4u, 16, 0u, 0u, 16, 0u, 0u};
            f32x4 v = PD_MMA(4, 0, pd_sub<4>(x8b, 0), sq, 16, ex);
            if (wave == 0) pd_stg(rs, a.o_q + te, v, T + 5u, col_on);
        }
        PD_STAMP();
        // ---- stage 5/6: Bahdanau scores, softmax, context (all workgroups)
        // (workgroups whose attention utterance is a padding column skip the stage: nobody reads that column)
        if (att_on) {
            // scores of this half: e_t = sum_d v_d tanh(keys[t,d] + q_d); lane owns dims 4*lane..+3, wave owns t
            const f32x4 qq = pd_wait_quad(rs, a.o_q + (unsigned)(lane * Bpad + b_att) * 8u, T + 5u, spin);
#ifdef PD_STAMPS
            if (tid == 0) spin.stamp(1);
#endif
            // (separate LDS / global loops: one loop over a selected pointer becomes flat loads, whose waits also
            // cover the sc1 stores in flight)
            auto score2 = [&](const float4 k0, const float4 k1, int t, int t1) {
                float s0 = v_att.x * fast_tanh(k0.x + qq.x) + v_att.y * fast_tanh(k0.y + qq.y) +
                           v_att.z * fast_tanh(k0.z + qq.z) + v_att.w * fast_tanh(k0.w + qq.w);
                float s1 = v_att.x * fast_tanh(k1.x + qq.x) + v_att.y * fast_tanh(k1.y + qq.y) +
                           v_att.z * fast_tanh(k1.z + qq.z) + v_att.w * fast_tanh(k1.w + qq.w);
                s0 = wave_sum(s0);
                s1 = wave_sum(s1);
                if (lane == 0) {
                    e_s[t_lo + t] = s0;
                    if (t1 < t_n) e_s[t_lo + t1] = s1;
                }
            };
            __syncthreads();                       // e_s free (previous step's context done)
            const bool q_small = !__any(!(fabsf(qq.x) < 40.f && fabsf(qq.y) < 40.f && fabsf(qq.z) < 40.f && fabsf(qq.w) < 40.f));
            if (keys_exp && q_small) {
                // e_t = sum_d v_d (1 - 2 / (E_k E_q + 1)) = sum_d v_d - 2 sum_d v_d / (E_k E_q + 1)
                constexpr float c2 = 2.8853900817779268f;
                const f32x4 eq = {__builtin_amdgcn_exp2f(qq.x * c2), __builtin_amdgcn_exp2f(qq.y * c2),
                                  __builtin_amdgcn_exp2f(qq.z * c2), __builtin_amdgcn_exp2f(qq.w * c2)};
                const float sv = (v_att.x + v_att.y) + (v_att.z + v_att.w);
                auto escore2 = [&](const float4 k0, const float4 k1, int t, int t1) {     // k0, k1: E_k of rows t, t1
                    float s0 = v_att.x * __builtin_amdgcn_rcpf(fmaf(k0.x, eq.x, 1.f)) + v_att.y * __builtin_amdgcn_rcpf(fmaf(k0.y, eq.y, 1.f)) +
                               v_att.z * __builtin_amdgcn_rcpf(fmaf(k0.z, eq.z, 1.f)) + v_att.w * __builtin_amdgcn_rcpf(fmaf(k0.w, eq.w, 1.f));
                    float s1 = v_att.x * __builtin_amdgcn_rcpf(fmaf(k1.x, eq.x, 1.f)) + v_att.y * __builtin_amdgcn_rcpf(fmaf(k1.y, eq.y, 1.f)) +
                               v_att.z * __builtin_amdgcn_rcpf(fmaf(k1.z, eq.z, 1.f)) + v_att.w * __builtin_amdgcn_rcpf(fmaf(k1.w, eq.w, 1.f));
                    s0 = wave_sum(fmaf(-2.f, s0, sv));
                    s1 = wave_sum(fmaf(-2.f, s1, sv));
                    if (lane == 0) {
                        e_s[t_lo + t] = s0;
                        if (t1 < t_n) e_s[t_lo + t1] = s1;
                    }
                };
                if (kl > 0) {
                    const float4* kp = reinterpret_cast<const float4*>(keys_s);
                    const int tw0 = wave < kl ? wave : 0;                  // (a part may have fewer rows than waves)
                    float4 k0 = kp[tw0 * 64 + lane];
                    float4 k1 = kp[(wave + PD_WAVES < kl ? wave + PD_WAVES : tw0) * 64 + lane];
                    for (int t = wave; t < kl; t += 2 * PD_WAVES) {
                        const int t1 = t + PD_WAVES, tn = t + 2 * PD_WAVES, tn1 = tn + PD_WAVES;
                        // next pair's keys in flight under this pair's arithmetic
                        const float4 n0 = kp[(tn < kl ? tn : t) * 64 + lane];
                        const float4 n1 = kp[(tn1 < kl ? tn1 : t) * 64 + lane];
                        escore2(k0, t1 < kl ? k1 : k0, t, t1 < kl ? t1 : t_n);
                        k0 = n0;
                        k1 = n1;
                    }
                }
                {
                    // rows beyond the staged ones: same arithmetic, E_k from the plain key in L2
                    const float4* kp = reinterpret_cast<const float4*>(keys_g);
                    auto ek = [&](int t) {
                        const float4 kv = kp[t * 64 + lane];
                        return make_float4(__builtin_amdgcn_exp2f(kv.x * c2), __builtin_amdgcn_exp2f(kv.y * c2),
                                           __builtin_amdgcn_exp2f(kv.z * c2), __builtin_amdgcn_exp2f(kv.w * c2));
                    };
                    for (int t = kl + wave; t < t_n; t += 2 * PD_WAVES) {
                        const int t1 = t + PD_WAVES;
                        escore2(ek(t), ek(t1 < t_n ? t1 : t), t, t1);
                    }
                }
            } else {
                // plain path: staged rows when LDS holds the plain keys, global memory otherwise
                const int kls = keys_exp ? 0 : kl;
                {
                    const float4* kp = reinterpret_cast<const float4*>(keys_s);
                    for (int t = wave; t < kls; t += 2 * PD_WAVES) {
                        const int t1 = t + PD_WAVES;
                        score2(kp[t * 64 + lane], kp[(t1 < kls ? t1 : t) * 64 + lane], t, t1 < kls ? t1 : t_n);
                    }
                }
                {
                    const float4* kp = reinterpret_cast<const float4*>(keys_g);
                    for (int t = kls + wave; t < t_n; t += 2 * PD_WAVES) {
                        const int t1 = t + PD_WAVES;
                        score2(kp[t * 64 + lane], kp[(t1 < t_n ? t1 : t) * 64 + lane], t, t1);
                    }
                }
            }
            __syncthreads();
            // publish this half's scores as {e, tag} granules (one coalesced store), fetch the other half's
            for (int t = tid; t < t_n; t += PD_THREADS)
                __builtin_amdgcn_raw_buffer_store_b64(u32x2{__float_as_uint(e_s[t_lo + t]), T + 6u}, rs,
                                                      (a.o_e + (unsigned)(b_att * a.T_in + t_lo + t) * 2u) * 4u, 0, PD_AUX_ST);
            PD_STAMP();
            {
                const int o_n = a.T_in - t_n;                      // the rows of the other parts: [0, t_lo), [t_lo + t_n, T_in)
                for (int t0 = 0; t0 < o_n; t0 += PD_THREADS) {
                    const int ti = t0 + tid;
                    const int t = ti < t_lo ? ti : ti + t_n;
                    u32x2 g = {0u, T + 6u};
                    spin.begin(T + 6u);
                    while (true) {
                        PD_POLL_INV();
                        if (ti < o_n)
                            g = __builtin_amdgcn_raw_buffer_load_b64(rs, (a.o_e + (unsigned)(b_att * a.T_in + t) * 2u) * 4u, 0, PD_AUX_LD);
                        if (__all(g.y == T + 6u) || spin.give_up()) break;
                    }
                    if (ti < o_n) e_s[t] = __uint_as_float(g.x);
                }
            }
            __syncthreads();
            PD_STAMP();
            // softmax over all T_in (max-subtracted, as tf.nn.softmax); both halves compute it. Every wave finds the max
            // itself, the threads share the exponentials; the 1/sum factor is applied to the context and to the
            // alignment row at the end
            float inv;
            {
                float mx = -INFINITY;
                for (int t = lane; t < a.T_in; t += 64) mx = fmaxf(mx, e_s[t]);
                mx = wave_max(mx);
                float sum = 0.f;
                for (int t = tid; t < a.T_in; t += PD_THREADS) {
                    const float p = __builtin_amdgcn_exp2f((e_s[t] - mx) * 1.4426950408889634f);
                    p_s[t] = p;
                    sum += p;
                }
                sum = wave_sum(sum);
                if (lane == 0) wsum_s[wave] = sum;
                __syncthreads();
                inv = 1.f / ((wsum_s[0] + wsum_s[1]) + (wsum_s[2] + wsum_s[3]));
            }
            PD_STAMP();
            // context, this part's nq channel quads: thread = (class c of t mod 8, quad); a channel is always the sum of
            // the same eight partial sums over t = c, c + 8, ..., combined in the same order, whatever P is (for P = 2
            // this is lane = (t parity, quad), waves striding t by 8)
            const int dq = tid % nq, tc = tid / nq;
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
            if (tc < 8) {
                if (a.lds_vals) {
                    const float4* vp = reinterpret_cast<const float4*>(vals_s);
#pragma unroll 5
                    for (int t = tc; t < a.T_in; t += 8) {
                        const float4 x = vp[t * nq + dq];
                        const float p = p_s[t];
                        acc.x = fmaf(p, x.x, acc.x); acc.y = fmaf(p, x.y, acc.y);
                        acc.z = fmaf(p, x.z, acc.z); acc.w = fmaf(p, x.w, acc.w);
                    }
                } else {
#pragma unroll 5
                    for (int t = tc; t < a.T_in; t += 8) {
                        const float4 x = reinterpret_cast<const float4*>(vals_g + (size_t)t * 256)[dq];
                        const float p = p_s[t];
                        acc.x = fmaf(p, x.x, acc.x); acc.y = fmaf(p, x.y, acc.y);
                        acc.z = fmaf(p, x.z, acc.z); acc.w = fmaf(p, x.w, acc.w);
                    }
                }
                red[tid] = acc;                    // [class][quad]
            }
            __syncthreads();
            if (tid < nq) {
                f32x4 c = red[tid] + red[nq + tid];
#pragma unroll
                for (int ww = 1; ww < 4; ++ww) c += red[(2 * ww) * nq + tid] + red[(2 * ww + 1) * nq + tid];
                pd_stg(rs, o_ctx + (unsigned)((part * nq + tid) * Bpad + b_att) * 8u, c * inv, T + 7u);
            }
            if (part == 0) {    // alignment row (tacotron.py:51 gets it transposed at the end): off the chain
                float* arow = a.alignS + ((size_t)s * a.B + b_att) * a.T_in;
                for (int t = tid; t < a.T_in; t += PD_THREADS) arow[t] = p_s[t] * inv;
            }
            PD_STAMP();
        }
        // ---- stage 8/9: residual GRU 1 (decoder.py:30) with the concat projection (rnn_wrappers.py:53, decoder.py:29)
        // folded in: gates (A: reset tile, B: update tile) and candidate (A) read [h_att ; ctx ; h1] / [h_att ; ctx ; r.h1]
        {
            const Segs sg{hc, T + 4u, 16, a.o_h1 + prv, s ? Tp + 10u : 0u, 32, o_ctx, T + 7u};
            f32x4 v = PD_MMA(12, 0, g12, sg, 48, ex);
            if (wave == 0) {
                const f32x4 g = sig4(v + PD_BIAS(SL_G12));
                if (roleA) pd_stg(rs, a.o_rh + st_stride + te, g * hA_t, T + 9u, col_on);
                else pd_stg(rs, a.o_u + st_stride + te, g, T + 9u, col_on);
            }
        }
        PD_STAMP();
        if (!roleA) {
            // d0 = W_p [h_att ; ctx] + b_p is only needed for the residual o1 = h1 + d0: role B computes it while
            // role A is busy with the candidate, which picks its tile up with its update gate
            const Segs sp{hc, T + 4u, 16, o_ctx, T + 7u, 32, 0u, 0u};
            f32x4 v = PD_MMA(8, 0, x8b, sp, 32, ex);
            if (wave == 0) pd_stg(rs, a.o_d0 + te, v + PD_BIAS(SL_X8B0), T + 8u, col_on);
        } else {
            ex[0].off = a.o_u + st_stride + te; ex[0].tag = T + 9u;
            ex[1].off = a.o_d0 + te; ex[1].tag = T + 8u;
            const Segs sg{hc, T + 4u, 16, o_ctx, T + 7u, 32, a.o_rh + st_stride, T + 9u};
            f32x4 v = PD_MMA(12, 2, x12, sg, 48, ex);
            if (wave == 0) {
                const f32x4 ug = ex[0].val;
                const f32x4 c = tanh4(v + PD_BIAS(SL_X12));
                hA_t = ug * hA_t + (1.f - ug) * c;
                o1_t = hA_t + ex[1].val;
                pd_stg(rs, a.o_h1 + cur + te, hA_t, T + 10u, col_on);
                pd_stg(rs, a.o_o1 + te, o1_t, T + 10u, col_on);
            }
        }
        PD_STAMP();
        // ---- stage 10/11: residual GRU 2 (decoder.py:31): gates (A: update tile, B: reset tile), candidate (B)
        {
            const Segs sg{a.o_h2 + prv, s ? Tp + 12u : 0u, 16, a.o_o1, T + 10u, 32, 0u, 0u};
            f32x4 v = PD_MMA(8, 0, g8, sg, 32, ex);
            if (wave == 0) {
                const f32x4 g = sig4(v + PD_BIAS(SL_G8));
                if (roleA) pd_stg(rs, a.o_u + 2u * st_stride + te, g, T + 11u, col_on);
                else pd_stg(rs, a.o_rh + 2u * st_stride + te, g * hB_t, T + 11u, col_on);
            }
        }
        PD_STAMP();
        if (!roleA) {
            ex[0].off = a.o_u + 2u * st_stride + te; ex[0].tag = T + 11u;
            ex[1].off = a.o_o1 + te; ex[1].tag = T + 10u;
            const Segs sg{a.o_o1, T + 10u, 16, a.o_rh + 2u * st_stride, T + 11u, 32, 0u, 0u};
            f32x4 v = PD_MMA(8, 2, x8a, sg, 32, ex);
            if (wave == 0) {
                const f32x4 ug = ex[0].val;
                const f32x4 c = tanh4(v + PD_BIAS(SL_X8A));
                hB_t = ug * hB_t + (1.f - ug) * c;
                pd_stg(rs, a.o_h2 + cur + te, hB_t, T + 12u, col_on);
                pd_stg(rs, a.o_o2 + te, hB_t + ex[1].val, T + 12u, col_on);
            }
        }
        PD_STAMP();
#ifdef PD_STAMPS
        if (spin.stamp_on && tid == 0)
            for (int i = 0; i < spin.stamp_i; ++i) a.dbg[i] = spin.st_lds[i];
        spin.stamp_on = false;
#endif
#ifdef PD_PHASE_STAMPS
        if (blockIdx.x == 0 && tid == 0 && (s == 0 || s == a.S - 1))
            reinterpret_cast<unsigned long long*>(a.sync)[s == 0 ? 9 : 10] = __builtin_amdgcn_s_memrealtime();
#endif
    }
    {
        Extra ex[2];
        PD_OUT(a.S - 1, (unsigned)(PD_STAGES * (a.S - 1)) + 12u);
    }
#undef PD_OUT
#undef PD_MMA
#undef PD_BIAS
}

// ---------------------------------------------------------------------------------------------------------------
// Hand-off probe (measurement only, taco_debug_handoff_probe): the transport of ONE dependency edge of the decoder
// step, with nothing else in it. 32 workgroups, one per CU like a tile's group; every round each workgroup publishes a
// 1 KB piece of a 256 x 16 activation tile as {value, tag} granules (wave 0, sc1 stores) and then gathers all 32 pieces
// (32 KB, the four waves a quarter each, sc1 loads, retried until every tag matches), reduces across its waves through
// LDS as pd_mma does, and only then publishes the next round's piece, whose value depends on what it gathered: a chain
// of `rounds` dependent all-to-all hand-offs. Its time per round is the floor of one edge of the step on this device
// at this moment: bench.py prices the step's dependency chain with it (11 edges per step).
__global__ __launch_bounds__(PD_THREADS) void pd_handoff_probe_kernel(float* region, unsigned region_bytes, int rounds,
                                                                      unsigned* abort_word, float* sink) {
    __shared__ f32x4 red[PD_WAVES * 64];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), j = blockIdx.x;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(region, 0, region_bytes, 0x00020000);
    Spin spin{abort_word, 0u, false, 0u, 0ull};
#ifdef PD_STAMPS
    spin.st_lds = nullptr; spin.stamp_i = 0; spin.stamp_on = false;
#endif
    f32x4 mine = {1.f, 2.f, 3.f, 4.f};
    constexpr unsigned PIECE = 32u * 8u;                     // floats per piece: 32 lanes x one granule quad
    for (int r = 0; r < rounds; ++r) {
        const unsigned tag = (unsigned)r + 1u;
        const unsigned buf = (unsigned)(r & 1) * 32u * PIECE;
        if (wave == 0) pd_stg(rs, buf + (unsigned)j * PIECE + (unsigned)(lane & 31) * 8u, mine, tag, lane < 32);
        GQuad g[4] = {};
        bool have[4] = {false, false, false, false};
        spin.begin(tag);
        while (true) {
            bool ok = true;
#pragma unroll
            for (int i = 0; i < 4; ++i)
                if (!have[i]) g[i] = pd_ldg(rs, buf + (unsigned)(wave + 4 * (2 * i + (lane >> 5))) * PIECE + (unsigned)(lane & 31) * 8u);
#pragma unroll
            for (int i = 0; i < 4; ++i)
                if (!have[i]) {
                    have[i] = __all(gq_ok(g[i], tag));
                    ok = ok && have[i];
                }
            if (ok || spin.give_up()) break;
        }
        f32x4 acc = gq_val(g[0]) + gq_val(g[1]) + gq_val(g[2]) + gq_val(g[3]);
        __syncthreads();
        red[wave * 64 + lane] = acc;
        __syncthreads();
        if (wave == 0) {
            f32x4 v = red[lane];
#pragma unroll
            for (int ww = 1; ww < PD_WAVES; ++ww) v += red[ww * 64 + lane];
            mine = v * (1.f / 1024.f) + f32x4{1.f, 1.f, 1.f, 1.f};
        }
    }
    if (wave == 0 && lane == 0) sink[j] = mine.x;
}

int launch_handoff_probe(float* region, unsigned region_bytes, int rounds, unsigned* abort_word, float* sink,
                         hipStream_t s) {
    hipLaunchKernelGGL(pd_handoff_probe_kernel, dim3(32), dim3(PD_THREADS), 0, s, region, region_bytes, rounds,
                       abort_word, sink);
    return 0;
}

// LDS of a launch: sized for its fullest tile (the first one), whose workgroups hold the largest parts
size_t dec_persistent_lds_bytes(int T_in, int B, int* lds_keys, int* lds_vals) {
    const size_t cap = 160 * 1024;
    const int P = pd_att_parts(B < 16 ? B : 16);
    const int T4 = (T_in + 3) & ~3, Tp = (T_in + P - 1) / P;
    size_t base = 8192 + (size_t)ST_COUNT * 16 * 4 + (size_t)T4 * 8 + 32;
    base = (base + 15) & ~(size_t)15;
    const size_t vb = (size_t)T_in * (256 / P) * 4;
    const int rows_fit = (int)((cap - base) / 1024);
    *lds_keys = Tp < rows_fit ? Tp : rows_fit;                       // key rows of a part held in LDS
    const size_t kb = (size_t)*lds_keys * 1024;
    *lds_vals = *lds_keys == Tp && base + kb + vb <= cap;
    return base + kb + (*lds_vals ? vb : 0);
}

int launch_dec_persistent(DecPersistArgs a, hipStream_t s, Timing* tm) {
    // more than the default 64 KB of dynamic LDS: the attribute is per device
    static bool attr_set[64] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return -1;
    if (!attr_set[dev]) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(dec_persistent_kernel),
                                hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
            return -1;
        attr_set[dev] = true;
    }
    const size_t lds = dec_persistent_lds_bytes(a.T_in, a.B, &a.lds_keys, &a.lds_vals);
    if (tm) tm->begin(KK_DEC_PERSIST, s);
    static const bool pack_env = getenv("TACO_PD_XCD_PACK") != nullptr;
    a.xcd_pack = pack_env && a.Bpad == 16;
    hipLaunchKernelGGL(dec_persistent_kernel, dim3(a.xcd_pack ? 256 : 32 * (a.Bpad / 16)), dim3(PD_THREADS), lds, s, a);
    if (tm) tm->end(KK_DEC_PERSIST, s);
    return 0;
}

}  // namespace taco
