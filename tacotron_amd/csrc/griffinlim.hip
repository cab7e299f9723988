// Griffin-Lim as TF builds it in the reference (tools/audio.py:57-64,183-205):
//   mag = (10 ** ((clip(S,0,1)*100 - 100 + 20) * 0.05)) ** 1.5
//   y = istft(mag);  60 x { E = stft(y); y = istft(mag * E / max(1e-8, |E|)) }
// with stft = frame(1200, hop 300, no padding) * periodicHann -> rfft(2048, zero-padded at the
// end) and istft = irfft(2048)[:1200] * periodicHann -> overlap-add (no window-sum division).
//
// One kernel per iteration (gl_iter). A 256-thread workgroup owns a run of consecutive frames of one utterance:
// it first rebuilds the time samples those frames cover by overlap-adding the frames of the previous iteration
// (each sample once, fixed order: reproducible; the signal y itself is never materialised in HBM between
// iterations, gl_ola builds it once at the end) into LDS; then each of its four waves takes frames of the run and
// does the whole frame on its own, 16 complex points per lane in registers: window -> 2048-point real FFT as a
// 1024-point complex three-pass (16 x 16 x 4) transform + split (fft_core.h) -> re-phase against the target magnitude
// -> merge -> inverse -> window -> store. The waves never meet at a barrier inside the frame loop; a wave's passes
// exchange data through its private, padded LDS buffer.
// HBM/L2 traffic per frame and iteration: 4.1 KB magnitudes + ~5.7 KB frame reads + 4.8 KB frame writes.
#include <cstdlib>
#include <type_traits>

#include "kernels.h"
#include "fft_core.h"

namespace taco {

// 10^(a p) on the hardware exp2: the exponent a p log2(10) is formed as a high part and a compensation term (the
// rounding errors of the two products, recovered with fma), so the result is within about one ulp of the exact value;
// the reference's (10^a)^p rounds the inner power to fp32 first, which this does not reproduce (nor does the fp64
// oracle). Two library powf per element made this kernel run at 1 TB/s (0.26 ms for a batch of 32); now 4 elements a
// lane per step, one v_exp_f32 each.
__device__ __forceinline__ float gl_mag_one(float lin, float min_db, float ref_db, float power) {
    float s = fminf(fmaxf(lin, 0.f), 1.f);
    s = s * -min_db + min_db;                           // _denormalize_tensorflow (audio.py:276-277)
    const float a = (s + ref_db) * 0.05f;               // exponent of _db_to_amp_tensorflow (audio.py:226-227)
    constexpr float L_hi = 3.321928024291992f, L_lo = 7.0595369e-08f;   // log2(10) = L_hi + L_lo
    const float z = a * power;                          // tf.pow(S, power) (audio.py:64): 10^(a power)
    const float z_lo = __builtin_fmaf(a, power, -z);
    const float x = z * L_hi;
    const float x_lo = __builtin_fmaf(z, L_hi, -x) + __builtin_fmaf(z, L_lo, z_lo * L_hi);
    const float e = __builtin_amdgcn_exp2f(x);
    return __builtin_fmaf(e, x_lo * 0.6931471805599453f, e);
}

__global__ void gl_mag_kernel(const float* __restrict__ lin, float* __restrict__ mag, size_t n, float min_db,
                              float ref_db, float power) {
    // (four at a time only when both pointers are 16-byte aligned: a caller may hand in a row slice of a larger tensor)
    const size_t n4 = ((reinterpret_cast<size_t>(lin) | reinterpret_cast<size_t>(mag)) & 15) ? 0 : n / 4;
    const size_t stride = (size_t)gridDim.x * blockDim.x, i0 = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    for (size_t i = i0; i < n4; i += stride) {
        const float4 v = reinterpret_cast<const float4*>(lin)[i];
        reinterpret_cast<float4*>(mag)[i] =
            make_float4(gl_mag_one(v.x, min_db, ref_db, power), gl_mag_one(v.y, min_db, ref_db, power),
                        gl_mag_one(v.z, min_db, ref_db, power), gl_mag_one(v.w, min_db, ref_db, power));
    }
    for (size_t i = 4 * n4 + i0; i < n; i += stride) mag[i] = gl_mag_one(lin[i], min_db, ref_db, power);
}

void launch_gl_mag(const float* lin, float* mag, size_t n, float min_db, float ref_db, float power, hipStream_t s,
                   Timing* tm) {
    if (tm) tm->begin(KK_GL_MAG, s);
    int blocks = (int)std::min<size_t>(4096, (n / 4 + 255) / 256);
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(gl_mag_kernel, dim3(blocks), dim3(256), 0, s, lin, mag, n, min_db, ref_db, power);
    if (tm) tm->end(KK_GL_MAG, s);
}

constexpr int GL_THREADS = 256;
constexpr int GL_WAVES = 4;
constexpr int GL_RUN = 16;          // consecutive frames per workgroup (upper bound; the launcher may lower it)

// Sample g (frame coordinates: frame m covers [hop m, hop m + win)) of the window-sum-square normalised overlap-add
// of librosa's istft (the oracle's istft_librosa): sum of the covering frames / sum of their squared windows.
__device__ __forceinline__ float gl_ola_norm(const float* __restrict__ frames_u, const float* __restrict__ w, int g,
                                             int T, int hop, int win, int nover) {
    const int q = g / hop;
    const int mhi = min(T - 1, q);
    const int mlo = max(0, q - (nover - 1));
    float s = 0.f, ws = 0.f;
    for (int m2 = mlo; m2 <= mhi; ++m2) {
        const int off = g - hop * m2;
        if (off < win) {
            s += frames_u[(long)m2 * win + off];
            ws += w[off] * w[off];
        }
    }
    return ws > 1.17549435e-38f ? s / ws : s;       // util.tiny(float32)
}

// np.exp(1j * np.angle(X)) * mag (audio.py:179-180): unit phase where X == 0
__device__ __forceinline__ cplx gl_rephase_np(cplx X, float mag) {
    const float a2 = __builtin_fmaf(X.x, X.x, X.y * X.y);
    if (!(a2 > 1e-37f)) {                         // |X|^2 underflows: scale up first (exact zero -> unit phase)
        const cplx Y = {X.x * 0x1p64f, X.y * 0x1p64f};
        const float b2 = __builtin_fmaf(Y.x, Y.x, Y.y * Y.y);
        if (!(b2 > 0.f)) return {mag, 0.f};
        const float t = mag * __builtin_amdgcn_rsqf(b2);
        return {Y.x * t, Y.y * t};
    }
    const float s = mag * __builtin_amdgcn_rsqf(a2);
    return {X.x * s, X.y * s};
}

// LIBROSA = false: the TF graph of audio.py:183-205 (see the head of this file).
// LIBROSA = true: the numpy/librosa loop of audio.py:167-181 on the same frame FFT. librosa's frames are n_fft long
// with the Hann window centred in them, i.e. the TF frame shifted by lp = (n_fft - win)/2 samples: X_lib[k] =
// W^(lp k) X_tf[k], and the inverse takes Y_tf[k] = Y_lib[k] conj(W^(lp k)), so inside the loop the rotation cancels
// and only three things differ: the signal between iterations is the overlap-add divided by the window-sum-square and
// trimmed to [win/2, win/2 + hop (T-1)) in frame coordinates, the frames of the next analysis read it through
// np.pad(mode='reflect'), and the first synthesis starts from caller-supplied random phases exp(2 pi i u) (rotated by
// conj(W^(lp k))) instead of zero phase. first: 0 = iteration, 1 = first synthesis.
// LDS: [y segment: (run-1)*hop + win floats][window: win floats][twiddle tables: FFT_TW + 512 cplx][4 x FFT_LDS cplx]
// WIN_C / HOP_C: compile-time window and hop (the default 1200 / 300) so the range tests, the divisions by the hop and
// the LDS offsets fold; 0 = take the run-time arguments.
template <bool LIBROSA, int WIN_C, int HOP_C>
__global__ __launch_bounds__(GL_THREADS, 2) void gl_iter_kernel(const float* __restrict__ frames_in,
                                                             const float* __restrict__ mag,
                                                             float* __restrict__ frames_out,
                                                             const float2* __restrict__ twiddle,
                                                             const float* __restrict__ window,
                                                             const float* __restrict__ phase_u, int B, int T,
                                                             int hop_rt, int win_rt, int nbins, int first, int run,
                                                             int runs_per_utt) {
    extern __shared__ __attribute__((aligned(16))) unsigned char gl_smem[];
    const int win = WIN_C ? WIN_C : win_rt;
    const int hop = HOP_C ? HOP_C : hop_rt;
    const int seg_cap = (run - 1) * hop + win;
    float* yseg = reinterpret_cast<float*>(gl_smem);
    float* win_s = yseg + ((seg_cap + 3) & ~3);
    cplx* tw_s = reinterpret_cast<cplx*>(win_s + ((win + 3) & ~3));
    const cplx* T2 = tw_s;
    const cplx* T3 = tw_s + FFT_T2;
    cplx* wk_s = tw_s + FFT_TW;            // W[k], k < 512: twiddles of the split / merge step
    cplx* fft_s = wk_s + FFT_N / 2;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    cplx* lds = fft_s + wave * FFT_LDS;
    const cplx* Wt = reinterpret_cast<const cplx*>(twiddle);

    const int b = blockIdx.x / runs_per_utt;
    const int m0 = (blockIdx.x - b * runs_per_utt) * run;
    const int nfr = min(run, T - m0);
    const long fb = (long)b * T;                   // first frame of this utterance
    const int half_win = win >> 1;                 // complex samples carrying data
    const int nover = (win + hop - 1) / hop;       // frames covering one sample (4)

    for (int i = tid; i < win; i += GL_THREADS) win_s[i] = window[i];
    for (int e = tid; e < FFT_TW; e += GL_THREADS) tw_s[e] = fft_tw_entry(e, Wt);
    for (int e = tid; e < FFT_N / 2; e += GL_THREADS) wk_s[e] = Wt[e];
    if (LIBROSA && !first) {
        // ---- the samples the run covers, read through the trim and the reflect padding:
        // padded position hop*m0 + i is sample j = hop*m0 + i - win/2 of y (length hop (T-1)), reflected at its ends
        __syncthreads();                               // win_s
        const int seg = (nfr - 1) * hop + win;
        const int L = hop * (T - 1), ctr = win >> 1;
        for (int i = tid; i < seg; i += GL_THREADS) {
            int j = hop * m0 + i - ctr;
            if (j < 0) j = -j;
            if (j >= L) j = 2 * (L - 1) - j;
            yseg[i] = gl_ola_norm(frames_in + fb * win, win_s, j + ctr, T, hop, win, nover);
        }
    } else if (!first) {
        // ---- the samples the run covers: y[hop*m0 + i] = sum of the covering frames of the previous iteration
        // (ascending frame order, as gl_ola and the oracle add them)
        const int seg = (nfr - 1) * hop + win;
        if constexpr (WIN_C != 0 && HOP_C != 0 && HOP_C % 4 == 0 && WIN_C % HOP_C == 0) {
            // four samples per lane and step, all covering frames' loads issued together (clamped address, the
            // out-of-range frames multiplied away): the phase is a pure stream of independent 16-byte loads
            constexpr int NOVER = WIN_C / HOP_C;
            const float* fu = frames_in + fb * WIN_C;
#pragma unroll 2
            for (int i = 4 * tid; i < seg; i += 4 * GL_THREADS) {
                const int g = HOP_C * m0 + i;
                const int q = g / HOP_C;
                const int r = g - q * HOP_C;
                float4 part[NOVER];
#pragma unroll
                for (int d = 0; d < NOVER; ++d) {
                    const int m2 = q - (NOVER - 1) + d;               // ascending
                    const int mc = min(max(m2, 0), T - 1);
                    part[d] = *reinterpret_cast<const float4*>(fu + (long)mc * WIN_C + r + HOP_C * (NOVER - 1 - d));
                    if (m2 != mc) part[d] = make_float4(0.f, 0.f, 0.f, 0.f);
                }
                float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                for (int d = 0; d < NOVER; ++d) {
                    acc.x += part[d].x; acc.y += part[d].y; acc.z += part[d].z; acc.w += part[d].w;
                }
                *reinterpret_cast<float4*>(yseg + i) = acc;
            }
        } else {
            for (int i = 2 * tid; i < seg; i += 2 * GL_THREADS) {
                const int g = hop * m0 + i;
                const int q = g / hop;
                const int mhi = min(T - 1, q);
                const int mlo = max(0, q - (nover - 1));
                float sx = 0.f, sy = 0.f;
                for (int m2 = mlo; m2 <= mhi; ++m2) {
                    const int off = g - hop * m2;
                    if (off < win) {
                        const float2 v = *reinterpret_cast<const float2*>(frames_in + (fb + m2) * win + off);
                        sx += v.x;
                        sy += v.y;
                    }
                }
                *reinterpret_cast<float2*>(yseg + i) = make_float2(sx, sy);
            }
        }
    }
    __syncthreads();

    for (int fi = wave; fi < nfr; fi += GL_WAVES) {
        const long f = fb + m0 + fi;
        const float* mg = mag + f * nbins;
        // target magnitudes of this lane's bin pairs (k, 1024 - k), k = lane + 64 j: in flight under the forward FFT
        float mk[8], mn[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int k = lane + 64 * j;
            mk[j] = mg[k];
            mn[j] = mg[FFT_N - k];     // (k = 0: bin 1024)
        }
        const float m512 = mg[FFT_N / 2];
        cplx v[16];
        if (!first) {
            // ---- analysis: window, even/odd packing z[c] = (y[2c], y[2c+1]), c = lane + 64 i
            const float* ys = yseg + hop * fi;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int c = lane + 64 * i;
                cplx z = {0.f, 0.f};
                if (c < half_win) {
                    const float2 sv = *reinterpret_cast<const float2*>(ys + 2 * c);
                    const float2 wv = *reinterpret_cast<const float2*>(win_s + 2 * c);
                    z = {sv.x * wv.x, sv.y * wv.y};
                }
                v[i] = z;
            }
            fft_pass1<false>(v, lds, lane);
            __builtin_amdgcn_wave_barrier();
            fft_load(v, lds, lane);
            __builtin_amdgcn_wave_barrier();
            fft_pass2<false>(v, lds, lane, T2);
            __builtin_amdgcn_wave_barrier();
            fft_load(v, lds, lane);
            __builtin_amdgcn_wave_barrier();
            fft_pass3<false>(v, lane, T3);                // v[m] = Z[lane + 64 m]
            // upper half to LDS: the partner bins 1024 - k of the other lanes
#pragma unroll
            for (int i = 8; i < 16; ++i) lds[fft_pad(lane + 64 * i)] = v[i];
            __builtin_amdgcn_wave_barrier();
        }
        // ---- split -> re-phase against mag -> merge on the bin pairs (k, 1024 - k), k = lane + 64 j
        const float* pu = LIBROSA && phase_u ? phase_u + f * nbins : nullptr;
        const int lp = FFT_N - (win >> 1);            // (n_fft - win) / 2
        auto rephase = [](cplx X, float m) -> cplx { return LIBROSA ? gl_rephase_np(X, m) : gl_rephase(X, m); };
        // first synthesis of the librosa loop: mag * exp(2 pi i u[k]) * conj(W^(lp k)), W = exp(-2 pi i / n_fft)
        auto start = [&](float m, int k) -> cplx {
            const float rot = (float)((lp * k) & (2 * FFT_N - 1)) * (1.f / (2 * FFT_N));
            float sn, cs;
            sincospif(2.f * ((pu ? pu[k] : 0.f) + rot), &sn, &cs);
            return {m * cs, m * sn};
        };
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int k = lane + 64 * j;
            if (k == 0) {
                // bins 0 and 1024 are real: X0 = Re Z0 + Im Z0, X1024 = Re Z0 - Im Z0
                float y0, yn;
                if (first) {
                    // (irfft drops the imaginary parts of bins 0 and n_fft/2; their rotation is 1)
                    y0 = LIBROSA ? start(mk[0], 0).x : mk[0];
                    yn = LIBROSA ? start(mn[0], FFT_N).x : mn[0];
                } else {
                    y0 = rephase(cplx{v[0].x + v[0].y, 0.f}, mk[0]).x;
                    yn = rephase(cplx{v[0].x - v[0].y, 0.f}, mn[0]).x;
                }
                v[0] = {0.5f * (y0 + yn), 0.5f * (y0 - yn)};
                // bin 512 pairs with itself: X512 = conj(Z512); Z'512 = conj(Y512)
                cplx ym;
                if (first) ym = LIBROSA ? start(m512, FFT_N / 2) : cplx{m512, 0.f};
                else ym = rephase(cconj(v[8]), m512);
                lds[fft_pad(FFT_N / 2)] = cconj(ym);
            } else {
                const cplx Wk = wk_s[k];
                cplx Yk, Yn;
                if (first) {
                    if (LIBROSA) { Yk = start(mk[j], k); Yn = start(mn[j], FFT_N - k); }
                    else { Yk = {mk[j], 0.f}; Yn = {mn[j], 0.f}; }
                } else {
                    cplx Xk, Xn;
                    rfft_split(v[j], lds[fft_pad(FFT_N - k)], Wk, Xk, Xn);
                    Yk = rephase(Xk, mk[j]);
                    Yn = rephase(Xn, mn[j]);
                }
                cplx Zk, Zn;
                irfft_merge(Yk, Yn, Wk, Zk, Zn);
                v[j] = Zk;
                lds[fft_pad(FFT_N - k)] = Zn;
            }
        }
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int i = 8; i < 16; ++i) v[i] = lds[fft_pad(lane + 64 * i)];
        __builtin_amdgcn_wave_barrier();
        // ---- inverse FFT-1024
        fft_pass1<true>(v, lds, lane);
        __builtin_amdgcn_wave_barrier();
        fft_load(v, lds, lane);
        __builtin_amdgcn_wave_barrier();
        fft_pass2<true>(v, lds, lane, T2);
        __builtin_amdgcn_wave_barrier();
        fft_load(v, lds, lane);
        __builtin_amdgcn_wave_barrier();
        fft_pass3<true>(v, lane, T3);
        // ---- synthesis window, first `win` samples, 1/1024 scaling
        const float sc = 1.f / FFT_N;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int c = lane + 64 * i;
            if (c < half_win) {
                const float2 wv = *reinterpret_cast<const float2*>(win_s + 2 * c);
                *reinterpret_cast<float2*>(frames_out + f * win + 2 * c) = make_float2(v[i].x * sc * wv.x, v[i].y * sc * wv.y);
            }
        }
    }
}


// ---------------------------------------------------------------------------------------------------------------
// gl_iter2: the iteration of the TF flavour at the default geometry (compile-time window / hop), two frames per wave.
// Same arithmetic per frame as gl_iter_kernel<false, WIN, HOP>; what changes is the order of issue. A frame is a chain
// of seven stages separated by exchanges through the wave's LDS buffer (write 16, read 16), and with one frame per wave
// every exchange is an exposed LDS round trip: at two waves per SIMD the counters show the waves parked in s_waitcnt
// for a third of their life and the VALU busy about 40 % of the time. Here a wave carries two independent frames A and
// B through the stages half a stage apart, A.math -> A.write -> A.read | B.math -> B.write -> B.read | A.math ..., so
// the round trip of one frame's exchange lies under the other frame's arithmetic. Both frames share ONE exchange buffer:
// the LDS executes a wave's instructions in order, so B's writes (issued after A's reads) cannot overtake them.
// Every read group is made to land (GL2_LAND) right before the other frame's writes are issued: at that point it is the
// newest group outstanding, so its wait is a plain lgkmcnt(0) on data requested a whole stage earlier, not a counted
// wait entangled with the other frame's fresher operations.
// Zero padding is used: the 1200 windowed samples fill complex points 0..599, so inputs 10..15 of the forward
// transform's first radix-16 pass are zero (dft16_fwd_in10); of the inverse transform only outputs 0..9 are stored and
// the compiler drops the rest of the last pass. Bins 0 / 512 / 1024 take the generic pair formulas (k = 0 pairs with
// itself: Z[1024] = Z[0], W^0 = 1, the partner slot is a dummy entry behind the buffer), so the body has no branch.
// (The plane exchanges read with inline-asm ds_read2_b32, which the compiler's wait-count insertion does not see: the
// wait is written out here. Between such a read and its GL2_LAND nothing touches the registers read into.)
#define GL2_LAND(v)                                                                                               \
    do {                                                                                                          \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                        \
        _Pragma("unroll") for (int i_ = 0; i_ < 16; ++i_) asm volatile("" : "+v"(v[i_]));                         \
    } while (0)

// first-stage DFT-16 of a vector whose elements 10..15 are zero (natural order in and out, as dft16<false>)
__device__ __forceinline__ void dft16_fwd_in10(cplx (&v)[16]) {
    cplx v1[4], d[4], rm[4], rp[4];
#pragma unroll
    for (int a2 = 0; a2 < 2; ++a2) {          // (v[a2], v[4+a2], v[8+a2], 0)
        const cplx v0 = cadd(v[a2], v[8 + a2]);
        v1[a2] = csub(v[a2], v[8 + a2]);
        d[a2] = v[4 + a2];
        v[a2] = cadd(v0, d[a2]);
        v[8 + a2] = csub(v0, d[a2]);
    }
#pragma unroll
    for (int a2 = 2; a2 < 4; ++a2) {          // (v[a2], v[4+a2], 0, 0)
        v1[a2] = v[a2];
        d[a2] = v[4 + a2];
        v[a2] = cadd(v1[a2], d[a2]);
        v[8 + a2] = csub(v1[a2], d[a2]);
    }
    cadd_mi_pi4(v1, d, rm, rp);               // v1 + d * (-i), v1 + d * (+i)
#pragma unroll
    for (int a2 = 0; a2 < 4; ++a2) {
        v[4 + a2] = rm[a2];
        v[12 + a2] = rp[a2];
    }
#pragma unroll
    for (int c1 = 1; c1 < 4; ++c1)
#pragma unroll
        for (int a2 = 1; a2 < 4; ++a2) v[4 * c1 + a2] = cmul(v[4 * c1 + a2], w16<false>(a2 * c1));
    dft4x4<false, 0, 4, 1>(v);
    cplx o[16];
#pragma unroll
    for (int c1 = 0; c1 < 4; ++c1)
#pragma unroll
        for (int c2 = 0; c2 < 4; ++c2) o[c1 + 4 * c2] = v[4 * c1 + c2];
#pragma unroll
    for (int c = 0; c < 16; ++c) v[c] = o[c];
}

#ifdef GL_STAMPS   // diagnostic build: s_memtime stamps of wave 0 of one workgroup (never in the shipped library)
__device__ unsigned long long gl_dbg[64];
#define GL_STAMP(i)                                                                                   \
    do {                                                                                              \
        if (blockIdx.x == GL_STAMP_WG && tid == 0) {                                                  \
            __builtin_amdgcn_sched_barrier(0);                                                        \
            gl_dbg[i] = __builtin_amdgcn_s_memtime();                                                 \
            __builtin_amdgcn_sched_barrier(0);                                                        \
        }                                                                                             \
    } while (0)
#ifndef GL_STAMP_WG
#define GL_STAMP_WG 1000
#endif
#else
#define GL_STAMP(i) do {} while (0)
#endif

// complex elements of a wave's exchange buffer: the padded AoS image (FFT_LDS + the dummy partner slot of bin pair
// (0, 1024) at fft_pad(1024) = 1088) or sixteen register planes of up to 144 dwords (below), whichever is larger
constexpr int GL2_LDS = 1152;
static_assert(GL2_LDS >= FFT_LDS + 8, "the padded image must fit");

template <int WIN, int HOP>
struct GL2Frame {
    cplx v[16];
    cplx z512;
    const float* ys;         // the frame's samples in the workgroup's LDS segment
    float* out;              // its row of frames_out
};

// LDS-only workgroup barrier: waits for the wave's LDS operations, not for its global loads and stores in flight (the
// magnitudes of the next pair, the record stores), which __syncthreads() would drain too
__device__ __forceinline__ void gl_lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// State between iterations. SIN / SOUT = false: the windowed frames of the previous iteration, [B*T][WIN] (every sample
// four times: 4.8 KB per frame written and read). true: RUN RECORDS, the overlap-added signal itself plus what the
// neighbouring runs need to finish their sums in the same order, per run of `run` frames
//   [H0: WIN-HOP | H1: WIN-2 HOP | H2: WIN-3 HOP | SUMS: HOP run]  (2 (WIN-HOP) + HOP run floats)
// H_f = frame f of the run over the run's first WIN-HOP samples (which frames of the previous run also cover), raw;
// SUMS[s - (WIN-HOP)] = sum of the RUN'S OWN frames covering sample s >= WIN-HOP of the run, in ascending frame order
// (complete for s < HOP nfr; for the last WIN-HOP samples, which frames of the next run also cover, the prefix of the
// sample's sum). The next iteration forms y[s] = ((a + b) + c) + d with a = the previous run's prefix (head samples) or
// the own sum, and b, c, d = own / next run's raw head frames in ascending order or 0: the very additions of the
// frame-state gather, in the same order, so both states give the same bits (and so do runs of 8 and of 16 frames).
// (1.3 KB per frame written, 2.3 KB read at run = 16.)
#ifndef GL_EXP_LB      // (experiment: minimum waves per SIMD the register allocation is held to)
#define GL_EXP_LB 2
#endif
template <int WIN, int HOP, bool SIN, bool SOUT>
__global__ __launch_bounds__(GL_THREADS, GL_EXP_LB) void gl_iter2_kernel(const float* __restrict__ frames_in,
                                                                const float* __restrict__ mag,
                                                                float* __restrict__ frames_out,
                                                                const float* __restrict__ lds_image, int B, int T,
                                                                int nbins, int run, int runs_per_utt) {
    extern __shared__ __attribute__((aligned(16))) unsigned char gl_smem[];
    static_assert(WIN % 4 == 0 && HOP % 4 == 0 && WIN % HOP == 0 && WIN / 2 <= 640 && WIN / 2 > 576,
                  "gl_iter2 is written for windows of 1154..1280 samples (ten live inputs per lane)");
    static_assert(WIN == 4 * HOP, "run records are laid out for four covering frames per sample");
    constexpr int HALF = WIN / 2;
    constexpr int HEAD = WIN - HOP;                      // samples at a run's start / end shared with the neighbour run
    constexpr int HOFF1 = HEAD, HOFF2 = 2 * HEAD - HOP;  // H1, H2 in a record; SUMS[s - HEAD] sits at 2 HEAD + s - HEAD
    const int RS = 2 * HEAD + HOP * run;                 // floats per record
    const int seg_cap = (run - 1) * HOP + WIN;
    float* yseg = reinterpret_cast<float*>(gl_smem);
    float* win_s = yseg + ((seg_cap + 3) & ~3);
    cplx* tw_s = reinterpret_cast<cplx*>(win_s + WIN);
    const cplx* T2 = tw_s;
    const cplx* T3 = tw_s + FFT_T2;
    cplx* wk_s = tw_s + FFT_TW;
    cplx* fft_s = wk_s + FFT_N / 2;
    // (the wave number as a scalar: the wave's buffer, its frames and their rows are then scalar values too)
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane_ = lane;
    cplx* lds = fft_s + wave * GL2_LDS;
    // SOUT: partial sums of the samples [8 HOP, 8 HOP + HEAD) of the run, which frames of both rounds of a 16-frame run
    // cover (the second round's inputs still sit in yseg at those places when the first round's outputs arrive)
    float* side_s = reinterpret_cast<float*>(fft_s + GL_WAVES * GL2_LDS);
    float* zero_s = side_s + WIN - HOP;                  // SOUT: HOP zeros (set before the first barrier)
    if (SOUT && tid < HOP / 4) reinterpret_cast<float4*>(zero_s)[tid] = make_float4(0.f, 0.f, 0.f, 0.f);

    // Workgroup i is placed on XCD i % 8: the runs are dealt out so that the workgroups of one XCD hold a contiguous
    // range of runs, and neighbouring runs (which read three frames of each other's range) meet in that XCD's L2.
    int wg = blockIdx.x;
    {
        const int per = gridDim.x >> 3, rem = gridDim.x & 7, x = wg & 7;
        wg = x * per + min(x, rem) + (wg >> 3);
    }
    const int b = wg / runs_per_utt;
    const int ru = wg - b * runs_per_utt;                // the run's index in its utterance
    const int m0 = ru * run;
    const int nfr = min(run, T - m0);
    const long fb = (long)b * T;
    typedef float f4v __attribute__((ext_vector_type(4)));
    (void)RS; (void)side_s; (void)ru; (void)zero_s;

    GL_STAMP(0);
    // target magnitudes of the lane's bin pairs (k, 1024 - k), k = lane + 64 j, for the two frames of the wave's first
    // pair: requested before anything else, so that the loads (from HBM: the magnitudes of a batch do not stay in any
    // cache between iterations) are long back when the first re-phasing needs them; the next pair's are requested when
    // this pair's are dead (after m4)
    float mkA[8], mnA[8], m512A, mkB[8], mnB[8], m512B;
    // The lane's eight bin pairs (k, 1024 - k), in the order the last forward pass leaves them in its registers (see
    // x_pass2 below): k = l, 256 + l, t3, 256 + t3, 64 + l, 320 + l, 192 - l, 448 - l with t3 = 256 - l (128 in lane 0)
    const int t3l = lane ? 256 - lane : 128;
    auto pair_bin = [&](int j) {
        switch (j) {
            case 0: return lane;
            case 1: return 256 + lane;
            case 2: return t3l;
            case 3: return 256 + t3l;
            case 4: return 64 + lane;
            case 5: return 320 + lane;
            case 6: return 192 - lane;
            default: return 448 - lane;
        }
    };
    // (through a buffer resource over the run's rows: a 32-bit lane offset + the frame's row as the scalar offset,
    // instead of sixteen 64-bit lane addresses kept alive from pair to pair)
    const __amdgpu_buffer_rsrc_t mag_rs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(mag + (fb + m0) * nbins), 0, nfr * nbins * 4, 0x00020000);
    auto load_mag = [&](float (&mk)[8], float (&mn)[8], float& m512, int fi) {
        const unsigned row = (unsigned)fi * (unsigned)nbins * 4u;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const unsigned k = (unsigned)pair_bin(j);
            mk[j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(mag_rs, 4u * k, row, 0));
            mn[j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(mag_rs, 4u * (FFT_N - k), row, 0));
        }
        m512 = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(mag_rs, 4u * (FFT_N / 2), row, 0));
    };
    load_mag(mkA, mnA, m512A, min(wave, nfr - 1));
    load_mag(mkB, mnB, m512B, min(wave + GL_WAVES, nfr - 1));
    // window + twiddle tables: a straight copy of the image taco_create built (WIN + 2 (FFT_TW + 512) floats). The
    // loads are issued here, the LDS stores follow the gather's loads below: one memory round trip for both.
    constexpr int N4 = (WIN + 2 * (FFT_TW + FFT_N / 2)) / 4;
    constexpr int TSTEPS = (N4 + GL_THREADS - 1) / GL_THREADS;
    f4v timg[TSTEPS];
#pragma unroll
    for (int k = 0; k < TSTEPS; ++k) {
        const int i = tid + GL_THREADS * k;
        if (i < N4) timg[k] = reinterpret_cast<const f4v*>(lds_image)[i];
    }
    auto store_tables = [&]() {
#pragma unroll
        for (int k = 0; k < TSTEPS; ++k) {
            const int i = tid + GL_THREADS * k;
            if (i < N4) reinterpret_cast<f4v*>(win_s)[i] = timg[k];
        }
    };
    GL_STAMP(1);
    {
        // the samples the run covers: y[HOP*m0 + i] = ((f[q-3] + f[q-2]) + f[q-1]) + f[q], the frames of the previous
        // iteration that cover the sample in ascending order (frames outside the utterance count as zero); four samples
        // per lane and step, the four loads issued together. A run whose samples are covered by frames of the
        // utterance only (all but the first and the last run of an utterance) takes the loads as they are: four
        // addresses a constant apart, no range tests; the others clamp the frame index and replace what was out of
        // range by zero. A sample gets the same sum either way, so the result does not depend on the run length.
        const int seg = (nfr - 1) * HOP + WIN;
        constexpr int NOVER = WIN / HOP;
        static_assert(NOVER == 4, "the gather is written for four covering frames");
        const float* fu = frames_in + fb * WIN;
        if constexpr (SIN) {
            // from run records: a = the previous run's prefix sum (head samples, 0 for the first run) or the own sum,
            // then the raw head frames of the own (head samples) or the next run (tail samples) in ascending order
            const float* rec = frames_in + ((long)b * runs_per_utt + ru) * RS;
            const bool has_prev = ru > 0, has_next = ru + 1 < runs_per_utt;
            // A FULL run (nfr = run = 8 or 16: every run of an utterance but its last) takes the same loads and sums
            // without branches: ONE buffer resource over [previous | own | next] record, whose out-of-range offsets
            // return zero - what a sample does not have (no previous / next run, fewer than four covering head frames)
            // is an offset beyond the resource; which steps of a thread can meet head or tail samples at all is known
            // at compile time, the others are one load and one store (+ 0: the sum of the generic path's absent terms).
            auto gather_full = [&](auto nfrc) {
                constexpr int NFR = decltype(nfrc)::value;
                constexpr int RSF = 2 * HEAD + HOP * NFR;                 // floats per record at this run length
                constexpr int HQ = HOP / 4, NPOS = HQ * (NFR + 3), ST = (NPOS + GL_THREADS - 1) / GL_THREADS;
                constexpr int TAIL0 = HQ * NFR;                          // first group of four of the tail samples
                constexpr unsigned OOB = 0x80000000u;
                const __amdgpu_buffer_rsrc_t rs3 = __builtin_amdgcn_make_buffer_rsrc(
                    const_cast<float*>(rec - RSF), 0, (2 + (has_next ? 1 : 0)) * RSF * 4, 0x00020000);
                auto ld = [&](unsigned off) {
                    return __builtin_bit_cast(f4v, __builtin_amdgcn_raw_buffer_load_b128(rs3, off, 0, 0));
                };
                f4v ga[ST], gb[ST], gc[ST], gd[ST];
                const f4v zero4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int st = 0; st < ST; ++st) {
                    const int lo = GL_THREADS * st;
                    const bool hd = lo < 3 * HQ, tl = lo + GL_THREADS - 1 >= TAIL0;   // (known after unrolling)
                    const int pp = tid + lo;
                    ga[st] = gb[st] = gc[st] = gd[st] = zero4;
                    if (lo + GL_THREADS <= NPOS || pp < NPOS) {
                        unsigned oa = 4u * (RSF + HEAD) + 16u * (unsigned)pp;
                        if (hd) {
                            const unsigned oh = has_prev ? 4u * (HEAD + HOP * NFR) + 16u * (unsigned)pp : OOB;
                            oa = pp < 3 * HQ ? oh : oa;
                        }
                        ga[st] = ld(oa);
                        if (hd || tl) {
                            // r = the group's place among the 3 HQ head (tail) groups; frames 0, 1, 2 of the own (next) run
                            // cover the places from 0, HQ, 2 HQ on
                            const int r = pp - (hd ? 0 : TAIL0);
                            const unsigned base = (hd ? 4u * RSF : 8u * RSF) + 16u * (unsigned)r;
                            gb[st] = ld((unsigned)r < (unsigned)(3 * HQ) ? base : OOB);
                            gc[st] = ld((unsigned)(r - HQ) < (unsigned)(2 * HQ) ? base + 4u * (HOFF1 - HOP) : OOB);
                            gd[st] = ld((unsigned)(r - 2 * HQ) < (unsigned)HQ ? base + 4u * (HOFF2 - 2 * HOP) : OOB);
                        }
                    }
                }
                store_tables();
#pragma unroll
                for (int st = 0; st < ST; ++st) {
                    const int lo = GL_THREADS * st;
                    const bool hd = lo < 3 * HQ, tl = lo + GL_THREADS - 1 >= TAIL0;
                    const int pp = tid + lo;
                    if (lo + GL_THREADS <= NPOS || pp < NPOS) {
                        f4v y = ga[st] + gb[st];
                        if (hd || tl) y = (y + gc[st]) + gd[st];
                        *reinterpret_cast<f4v*>(yseg + 4 * pp) = y;
                    }
                }
            };
            (void)gather_full;
#ifndef GL_EXP_OLD_GATHER
            if (nfr == run && run == 16) gather_full(std::integral_constant<int, 16>());
            else if (nfr == run && run == 8) gather_full(std::integral_constant<int, 8>());
            else
#endif
            {
            constexpr int STEPS = ((GL_RUN - 1) * HOP + WIN + 4 * GL_THREADS - 1) / (4 * GL_THREADS);
            f4v ga[STEPS], gb[STEPS], gc[STEPS], gd[STEPS];
            const f4v zero4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int st = 0; st < STEPS; ++st) {
                const int i = 4 * (tid + GL_THREADS * st);
                ga[st] = gb[st] = gc[st] = gd[st] = zero4;
#ifdef GL_EXP_NO_GATHER
                if (false) {
#else
                if (i < seg) {
#endif
                    const float* hp = nullptr;           // record whose raw head frames cover the sample
                    int j = i;                           // ... and the sample's offset in that record's head
                    if (i < HEAD) {
                        if (has_prev) ga[st] = *reinterpret_cast<const f4v*>(rec - RS + HEAD + HOP * run + i);
                        hp = rec;
                    } else {
                        ga[st] = *reinterpret_cast<const f4v*>(rec + HEAD + i);
                        j = i - HOP * nfr;
                        if (j >= 0 && has_next) hp = rec + RS;
                    }
                    if (hp) {
                        gb[st] = *reinterpret_cast<const f4v*>(hp + j);
                        if (j >= HOP) gc[st] = *reinterpret_cast<const f4v*>(hp + HOFF1 + j - HOP);
                        if (j >= 2 * HOP) gd[st] = *reinterpret_cast<const f4v*>(hp + HOFF2 + j - 2 * HOP);
                    }
                }
            }
            store_tables();
#pragma unroll
            for (int st = 0; st < STEPS; ++st) {
                const int i = 4 * (tid + GL_THREADS * st);
                if (i < seg) *reinterpret_cast<f4v*>(yseg + i) = ((ga[st] + gb[st]) + gc[st]) + gd[st];
            }
            }
        } else if (m0 >= NOVER - 1 && m0 + nfr + NOVER - 2 <= T - 1) {
            const float* base = fu + (long)(m0 - (NOVER - 1)) * WIN + HOP * (NOVER - 1);   // frame q-3 at offset r + 3 HOP
            // every load of the thread is issued before the first sum: one memory round trip for the whole segment
            // (the registers are free here, the transform's working set is not live yet)
            constexpr int STEPS = ((GL_RUN - 1) * HOP + WIN + 4 * GL_THREADS - 1) / (4 * GL_THREADS);
            f4v ga[STEPS], gb[STEPS], gc[STEPS], gd[STEPS];
#pragma unroll
            for (int st = 0; st < STEPS; ++st) {
                const int i = 4 * (tid + GL_THREADS * st);
                if (i < seg) {
                    const int qq = i / HOP;
                    const float* p0 = base + qq * (WIN - HOP) + i;        // = base + qq*WIN + (i - qq*HOP)
#ifdef GL_EXP_NO_GATHER
                    ga[st] = gb[st] = gc[st] = gd[st] = f4v{1e-3f * (float)i, 0.f, 1e-3f, 0.f};
                    (void)p0;
#else
                    ga[st] = *reinterpret_cast<const f4v*>(p0);
                    gb[st] = *reinterpret_cast<const f4v*>(p0 + (WIN - HOP));
                    gc[st] = *reinterpret_cast<const f4v*>(p0 + 2 * (WIN - HOP));
                    gd[st] = *reinterpret_cast<const f4v*>(p0 + 3 * (WIN - HOP));
#endif
                }
            }
            store_tables();
#pragma unroll
            for (int st = 0; st < STEPS; ++st) {
                const int i = 4 * (tid + GL_THREADS * st);
                if (i < seg) *reinterpret_cast<f4v*>(yseg + i) = ((ga[st] + gb[st]) + gc[st]) + gd[st];
            }
        } else {
            store_tables();
#pragma unroll 2
            for (int i = 4 * tid; i < seg; i += 4 * GL_THREADS) {
                const int g = HOP * m0 + i;
                const int q = g / HOP;
                const int r = g - q * HOP;
                float4 part[NOVER];
#pragma unroll
                for (int d = 0; d < NOVER; ++d) {
                    const int m2 = q - (NOVER - 1) + d;
                    const int mc = min(max(m2, 0), T - 1);
                    part[d] = *reinterpret_cast<const float4*>(fu + (long)mc * WIN + r + HOP * (NOVER - 1 - d));
                    if (m2 != mc) part[d] = make_float4(0.f, 0.f, 0.f, 0.f);
                }
                float4 acc = part[0];
#pragma unroll
                for (int d = 1; d < NOVER; ++d) {
                    acc.x += part[d].x; acc.y += part[d].y; acc.z += part[d].z; acc.w += part[d].w;
                }
                *reinterpret_cast<float4*>(yseg + i) = acc;
            }
        }
    }
    GL_STAMP(2);
    __syncthreads();
    GL_STAMP(3);

    typedef GL2Frame<WIN, HOP> Fr;
    // (the twiddles of passes 2 and 3 are read from the LDS tables at each use: kept in registers for the whole kernel,
    // 54 per lane, they push the two frames' working set over the 256-register budget - measured 12.3 ms for 60
    // iterations at B = 32 with the pass-2 set resident and 40 registers spilled, 19.3 ms with both, 9.0 ms this way)
    // ---- the stages of one frame: m* = arithmetic on registers, x* = the exchange (16 LDS writes, then the reads
    // whose results the next m* stage needs)
    auto s0_load = [&](Fr& f, int fi) {
        f.ys = yseg + HOP * fi;
        f.out = frames_out + (fb + m0 + fi) * WIN;
    };
    // Table values: the pass-2 twiddles (they depend on lane & 15 only), the radix-4 pass's twiddles of the lane's four
    // butterflies and the lane's twenty window values are read once per workgroup and stay in registers (room for them
    // since the exchange addresses no longer take forty); only the split twiddles are read from LDS at each use.
    // (Before, tables were read per pair of frames, ahead of the other frame's exchange: LDS returns in order, so a
    // table read issued behind an exchange could only be used after that exchange had landed.)
    cplx tw[12], tw2[15];
#pragma unroll
    for (int i = 0; i < 15; ++i) tw2[i] = T2[i * 16 + (lane & 15)];
    // ... and so do the lane's twenty window values (analysis and synthesis use the same ones)
    float2 wreg[10];
#pragma unroll
    for (int i = 0; i < 10; ++i) {
        const int c = lane + 64 * i;
        wreg[i] = *reinterpret_cast<const float2*>(win_s + 2 * ((i < 9 || c < HALF) ? c : 0));
    }
    // ... and the radix-4 pass's twiddles of the lane's four butterflies
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int i = 0; i < 3; ++i) tw[3 * q + i] = T3[i * 256 + lane + 64 * q];
    // ... and the split twiddles of the lane's eight bin pairs: read inside m3 they would be issued BEHIND the other
    // frame's exchange, and the LDS returns a wave's reads in order - m3 could not start before that exchange had
    // landed (stamps: m3 took 5-7 k cycles beside a transforming neighbour, 2.3 k otherwise)
#ifndef GL_EXP_WK_LDS
    cplx wkr[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) wkr[j] = wk_s[lane + 64 * j];
#endif
    auto a1 = [&](Fr& f, Fr& g) {
        // analysis window, even/odd packing z[c] = (y[2c], y[2c+1]), c = lane + 64 i: i <= 8 always inside the window,
        // i = 9 for the first HALF - 576 lanes, i >= 10 never
#pragma unroll
        for (int i = 0; i < 10; ++i) {
            const int c = lane + 64 * i;
            const int cc = (i < 9 || c < HALF) ? c : 0;
            const float2 wv = wreg[i];
            const float2 sf = *reinterpret_cast<const float2*>(f.ys + 2 * cc);
            const float2 sg = *reinterpret_cast<const float2*>(g.ys + 2 * cc);
            cplx zf = {sf.x * wv.x, sf.y * wv.y}, zg = {sg.x * wv.x, sg.y * wv.y};
            if (i == 9 && c >= HALF) zf = zg = cplx{0.f, 0.f};
            f.v[i] = zf;
            g.v[i] = zg;
        }
    };
    auto d1 = [&](Fr& f) { dft16_fwd_in10(f.v); };
    // The padded places fft_pad(j) = j + (j >> 4) of the exchange patterns, written as ONE lane base + a constant per
    // element (the compiler, given fft_pad(lane + 64 i), forms a register per element instead: ~45 address registers
    // held over the whole kernel and ~250 instructions to set them up):
    //   fft_pad(lane + 64 i) = pr + 68 i, fft_pad(16 lane + c) = 17 lane + c, fft_pad(lane + 256 i) = pr + 272 i,
    //   fft_pad(64 + lane + 256 i) = pr + 68 + 272 i, fft_pad(192 - lane + 256 i) = pm + 272 i, fft_pad(t3l + 256 i)
    //   = pt + 272 i, fft_pad((lane >> 4) 256 + (lane & 15) + 16 c) = pb + 17 c   (checked for every lane, i, c)
    // (GL_RD / GL_RDI / GL_WRI: an exchange's reads and writes; the experiment builds below price them. As plain
    // accesses the compiler pairs the reads into ds_read2_b64 - 8 LDS-array cycles per two elements against 2 x 2 as
    // ds_read_b64; forcing the single form with volatile accesses made the register allocation spill 655 registers.)
#define GL_RD(p) (*(p))
#ifdef GL_EXP_HALFREAD   // (experiment, wrong results: every other exchange read dropped, the value left undefined)
#define GL_RDI(p, i) ((i) & 1 ? ({ cplx r_; asm volatile("" : "=v"(r_)); r_; }) : *(p))
#else
#define GL_RDI(p, i) GL_RD(p)
#endif
    // (GL_PIN: behind an exchange's reads. The scheduler, short of registers, otherwise sinks the reads to where their
    // values are first used - right in front of the wait for them, so that every exchange exposes its read latency
    // instead of spending it under the other frame's arithmetic.)
#ifdef GL_EXP_NO_PIN
#define GL_PIN() do {} while (0)
#else
#define GL_PIN() __builtin_amdgcn_sched_barrier(0)
#endif
#ifdef GL_EXP_ADDTID             // (experiment, wrong results: exchange writes as two ds_write_addtid_b32, lane-linear places)
#define GL_WRI(p, i, val) do { const cplx v__ = (val); asm volatile("ds_write_addtid_b32 %0 offset:%2\n\tds_write_addtid_b32 %1 offset:%3" :: "v"(v__.x), "v"(v__.y), "n"(256 * ((i) & 15)), "n"(4352 + 256 * ((i) & 15)) : "memory"); } while (0)
#elif defined(GL_EXP_HALFWRITE)          // (experiment, wrong results: every other exchange write dropped)
#define GL_WRI(p, i, val) do { if (!((i) & 1)) *(p) = (val); else asm volatile("" :: "v"(val)); } while (0)
#else
#define GL_WRI(p, i, val) (*(p) = (val))
#endif
    cplx* const l_r = lds + lane + (lane >> 4);
    cplx* const l_17 = lds + 17 * lane;
    cplx* const l_b = lds + (lane >> 4) * 272 + (lane & 15);
    cplx* const l_m = lds + (192 - lane) + ((192 - lane) >> 4);
    // ---- Register-plane exchanges (round 3). What an exchange costs is its WRITES: a ds_write moves its address and data
    // registers to the LDS at 2 cycles per dword (6 per complex element as ds_write_b64), and dropping every other
    // write of the four exchanges (wrong results) took 14 % off the kernel, dropping reads 2.5 %. ds_write_addtid_b32
    // takes no address register (address = M0 + offset + 4 lane): 4 cycles per element as a pair. Its places are
    // lane-linear, so the layout is the WRITER's: register r of lane l goes to plane r, [re: 64 lanes | im: 64 lanes],
    // planes S dwords apart; the reader works out which (register, lane) holds the element it wants and fetches
    // (re, im) with one ds_read2_b32 (offsets o, o + 64). S is chosen per exchange so that the reader's 32-lane groups
    // hit 32 different banks (pass 1 -> 2, pass 2 -> radix-4 pass, radix-4 pass -> pass 2 backwards; the reader of the
    // last one, pass 2 -> pass 1 backwards, takes lanes 16 b apart of one plane, 2-way whatever S is). Same values
    // moved: same bits. -DGL_EXP_AOS_EXCHANGE: all four through the padded image (before).
#ifndef GL_EXP_AOS_EXCHANGE
    constexpr int XS1 = 130, XS2 = 144, XS3 = 132;     // plane strides (dwords): 2 / 16 / 4 (mod 32) as the readers need
    static_assert(16 * XS2 <= 2 * GL2_LDS, "planes fit the wave's buffer");
    const float* const pf = reinterpret_cast<const float*>(lds);
    const unsigned lds_m0 = (unsigned)(size_t)lds;      // (the low half of a flat LDS address is the LDS byte address)
#ifdef GL_EXP_NO_EXCHANGE   // (experiment, wrong results: the exchanges move nothing - what the kernel costs without them)
#define GL_AT_BEGIN() do {} while (0)
#define GL_AT(val, S, r) asm volatile("" :: "v"((val).x), "v"((val).y))
#define GL_PRD(dst, addr, o) asm volatile("" : "+v"(dst) : "v"(addr))
#else
#define GL_AT_BEGIN() asm volatile("s_mov_b32 m0, %0\n\ts_nop 0" :: "s"(lds_m0) : "memory")
#define GL_AT(val, S, r)                                                                                          \
    asm volatile("ds_write_addtid_b32 %0 offset:%2\n\tds_write_addtid_b32 %1 offset:%3"                            \
                 :: "v"((val).x), "v"((val).y), "n"(4 * ((r) * (S))), "n"(4 * ((r) * (S) + 64)) : "memory")
    // (re, im) of one element with ONE ds_read2_b32 (offsets o, o + 64 dwords from a lane's byte address). As two plain
    // loads the compiler pairs neighbouring ELEMENTS' real parts instead and rebuilds the pairs with register moves.
#define GL_PRD(dst, addr, o)                                                                                      \
    asm volatile("ds_read2_b32 %0, %1 offset0:%2 offset1:%3" : "=v"(dst) : "v"(addr), "n"(o), "n"((o) + 64) : "memory")
#endif
    (void)pf;
    const unsigned x1b = lds_m0 + 4u * (unsigned)((lane & 15) * XS1 + (lane >> 4));
    auto x_pass1 = [&](Fr& f) {
        // writer: lane t holds y1[16 t + c] in register c; reader l wants y1[l + 64 i] = register l & 15 of lane
        // (l >> 4) + 4 i
        GL_AT_BEGIN();
#pragma unroll
        for (int c = 0; c < 16; ++c) GL_AT(f.v[c], XS1, c);
#pragma unroll
        for (int i = 0; i < 16; ++i) GL_PRD(f.v[i], x1b, 4 * i);
        GL_PIN();
    };
#else
    auto x_pass1 = [&](Fr& f) {
#pragma unroll
        for (int c = 0; c < 16; ++c) GL_WRI(l_17 + c, c, f.v[c]);
#pragma unroll
        for (int i = 0; i < 16; ++i) f.v[i] = GL_RDI(l_r + 68 * i, i);
        GL_PIN();
    };
#endif
    // Pass 2 writes y2[(t - k) 16 + k + 16 c]; the radix-4 pass that follows reads butterfly t's inputs y2[t + 256 i].
    // Which four butterflies a lane takes is free: here t = l, 64 + l, 192 - l, 256 - l (lane 0: 0, 64, 192, 128), so
    // that the outputs X[t + 256 i'] of a lane contain BOTH members of each of its eight bin pairs (k, 1024 - k): the
    // split / re-phase / merge of the real transform then works on registers alone, with no partner exchange through
    // LDS before and after it, and the inverse transform runs the three passes backwards from the same registers.
    cplx* const l_t = lds + t3l + (t3l >> 4);
    (void)l_b; (void)l_m; (void)l_t; (void)l_17; (void)l_r;
#ifndef GL_EXP_AOS_EXCHANGE
    // writer: lane t holds y2[256 (t >> 4) + 16 c + (t & 15)] in register c; the reader wants y2[tq + 256 i] for its
    // butterflies tq = l, 64 + l, 192 - l, t3l: register tq >> 4 of lane 16 i + (tq & 15)
    const unsigned x2b0 = lds_m0 + 4u * (unsigned)((lane >> 4) * XS2 + (lane & 15));
    const unsigned x2b1 = x2b0 + 4u * 4u * XS2;
    const unsigned x2b2 = lds_m0 + 4u * (unsigned)(((192 - lane) >> 4) * XS2 + ((192 - lane) & 15));
    const unsigned x2b3 = lds_m0 + 4u * (unsigned)((t3l >> 4) * XS2 + (t3l & 15));
    auto x_pass2 = [&](Fr& f) {
        GL_AT_BEGIN();
#pragma unroll
        for (int c = 0; c < 16; ++c) GL_AT(f.v[c], XS2, c);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            GL_PRD(f.v[4 * i], x2b0, 16 * i);
            GL_PRD(f.v[4 * i + 1], x2b1, 16 * i);
            GL_PRD(f.v[4 * i + 2], x2b2, 16 * i);
            GL_PRD(f.v[4 * i + 3], x2b3, 16 * i);
        }
        GL_PIN();
    };
#else
    auto x_pass2 = [&](Fr& f) {
#pragma unroll
        for (int c = 0; c < 16; ++c) GL_WRI(l_b + 17 * c, c, f.v[c]);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            f.v[4 * i] = GL_RDI(l_r + 272 * i, 0);
            f.v[4 * i + 1] = GL_RDI(l_r + 68 + 272 * i, 1);
            f.v[4 * i + 2] = GL_RDI(l_m + 272 * i, 0);
            f.v[4 * i + 3] = GL_RDI(l_t + 272 * i, 1);
        }
        GL_PIN();
    };
#endif
    auto m2f = [&](Fr& f) {
        cplx (&v)[16] = f.v;
        cmul_v4(v[1], v[2], v[3], v[4], tw2[0], tw2[1], tw2[2], tw2[3]);
        cmul_v4(v[5], v[6], v[7], v[8], tw2[4], tw2[5], tw2[6], tw2[7]);
        cmul_v4(v[9], v[10], v[11], v[12], tw2[8], tw2[9], tw2[10], tw2[11]);
        cmul_v3(v[13], v[14], v[15], tw2[12], tw2[13], tw2[14]);
    };
    // Radix-4 pass (tw = T3 of the lane's butterflies) -> v[s + 4 i'] = Z[t_s + 256 i']; bin pairs in registers
    // (0,15) (4,11) (3,12) (7,8) (1,14) (5,10) (2,13) (6,9) = pair_bin order; split -> re-phase against the target
    // magnitude -> merge as before (unnormalised, the halvings go into the final scale); then the radix-4 pass backwards.
    // Lane 0 holds the self-paired bins: Z[0] (pairs with itself: DC and Nyquist), Z[512] (conjugate of itself) and the
    // pairs (256, 768), (128, 896), (384, 640), which sit in other registers than the generic pattern expects: four of
    // its registers are rotated into the generic places before and back after.
    auto m3 = [&](Fr& f, const float (&mk)[8], const float (&mn)[8], float m512) {
        // (tw[3 q + i - 1]: twiddle of input i of butterfly q; three statements of four products)
        cmul_v4(f.v[4], f.v[8], f.v[12], f.v[5], tw[0], tw[1], tw[2], tw[3]);
        cmul_v4(f.v[9], f.v[13], f.v[6], f.v[10], tw[4], tw[5], tw[6], tw[7]);
        cmul_v4(f.v[14], f.v[7], f.v[11], f.v[15], tw[8], tw[9], tw[10], tw[11]);
        dft4x4<false, 0, 1, 4>(f.v);
        const bool l0 = lane == 0;
        cplx z512 = f.v[8];
        {
            const cplx a8 = f.v[8], a11 = f.v[11], a12 = f.v[12], a15 = f.v[15];
            f.v[8] = l0 ? a11 : a8;
            f.v[11] = l0 ? a12 : a11;
            f.v[12] = l0 ? a15 : a12;
            f.v[15] = l0 ? f.v[0] : a15;
        }
        constexpr int RK[8] = {0, 4, 3, 7, 1, 5, 2, 6}, RN[8] = {15, 11, 12, 8, 14, 10, 13, 9};
#pragma unroll
        for (int h = 0; h < 8; h += 4) {                 // four pairs at a time: their rotations and twiddle products
            cplx Wk[4], E[4], Tt[4], Xk[4], Un[4];       // are one statement each
#pragma unroll
            for (int j = 0; j < 4; ++j) {
#ifndef GL_EXP_WK_LDS
                Wk[j] = wkr[h + j];
#else
                Wk[j] = GL_RD(wk_s + lane + 64 * (h + j));
#endif
                const cplx zk = f.v[RK[h + j]];
                const cplx cn = cconj(f.v[RN[h + j]]);
                E[j] = cadd(zk, cn);
                Tt[j] = csub(zk, cn);
            }
            cmul_v4(Tt[0], Tt[1], Tt[2], Tt[3], Wk[0], Wk[1], Wk[2], Wk[3]);
            cadd_mi_pi4(E, Tt, Xk, Un);                   // 2 X[k], conj(2 X[1024 - k])
            cplx E2[4], O2[4], zm[4], zp[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float ak = __builtin_fmaf(Xk[j].x, Xk[j].x, Xk[j].y * Xk[j].y);
                const float an = __builtin_fmaf(Un[j].x, Un[j].x, Un[j].y * Un[j].y);
                const cplx Yk = Xk[j] * (mk[h + j] * __builtin_amdgcn_rsqf(ak > 4e-16f ? ak : 4e-16f));   // Y[k]
                const cplx Vn = Un[j] * (mn[h + j] * __builtin_amdgcn_rsqf(an > 4e-16f ? an : 4e-16f));   // conj(Y[1024 - k])
                E2[j] = cadd(Yk, Vn);
                O2[j] = csub(Yk, Vn);
            }
            cmul_vc4(O2[0], O2[1], O2[2], O2[3], Wk[0], Wk[1], Wk[2], Wk[3]);
            cadd_mi_pi4(E2, O2, zm, zp);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                f.v[RK[h + j]] = zp[j];                   // 2 Z'[k]
                f.v[RN[h + j]] = cconj(zm[j]);            // 2 Z'[1024 - k]
            }
        }
        // bin 512 pairs with itself: X512 = conj(Z512), Z'512 = conj(Y512)
        z512 = cconj(gl_rephase(cconj(z512), 2.f * m512));
        {
            const cplx a8 = f.v[8], a11 = f.v[11], a12 = f.v[12], a15 = f.v[15];
            f.v[11] = l0 ? a8 : a11;
            f.v[12] = l0 ? a11 : a12;
            f.v[15] = l0 ? a12 : a15;
            f.v[8] = l0 ? z512 : a8;
        }
        dft4x4<true, 0, 1, 4>(f.v);
        cmul_vc4(f.v[4], f.v[8], f.v[12], f.v[5], tw[0], tw[1], tw[2], tw[3]);
        cmul_vc4(f.v[9], f.v[13], f.v[6], f.v[10], tw[4], tw[5], tw[6], tw[7]);
        cmul_vc4(f.v[14], f.v[7], f.v[11], f.v[15], tw[8], tw[9], tw[10], tw[11]);
    };
    // the exchanges of the inverse are the forward ones backwards: written where the forward pass read, read where it wrote
#ifndef GL_EXP_AOS_EXCHANGE
    // writer: lane l holds y2'[tq_s(l) + 256 i] in register 4 i + s (tq_s = l, 64 + l, 192 - l, t3l); reader t wants
    // y2'[256 (t >> 4) + 16 c + k], k = t & 15, i.e. butterfly tq = 16 c + k, input i = t >> 4: c < 4 -> s = 0, lane
    // 16 c + k; c = 4..7 -> s = 1, lane 16 (c - 4) + k; tq in [129, 192] -> s = 2, lane 192 - tq; tq in [193, 255] -> s = 3,
    // lane 256 - tq; tq = 128 (c = 8, k = 0) is lane 0's fourth butterfly, tq = 192 (c = 12, k = 0) lane 0's third
    const int x3k = lane & 15;
    const unsigned x3a = lds_m0 + 4u * (unsigned)(4 * (lane >> 4) * XS3 + x3k);          // plane 4 i, lane k
    const unsigned x3b = lds_m0 + 4u * (unsigned)((4 * (lane >> 4) + 2) * XS3 - x3k);     // plane 4 i + 2, lane -k
    auto x3 = [&](Fr& f) {
        GL_AT_BEGIN();
#pragma unroll
        for (int r = 0; r < 16; ++r) GL_AT(f.v[r], XS3, r);
        const unsigned o8 = x3k ? x3b + 4u * 64u : x3b + 4u * XS3;      // (k = 0: plane 4 i + 3, lane 0)
        const unsigned o12 = x3k ? x3b + 4u * (XS3 + 64u) : x3b;        // (k = 0: plane 4 i + 2, lane 0)
#pragma unroll
        for (int c = 0; c < 16; ++c) {
            if (c < 4) GL_PRD(f.v[c], x3a, 16 * c);
            else if (c < 8) GL_PRD(f.v[c], x3a, XS3 + 16 * (c - 4));
            else if (c == 8) GL_PRD(f.v[c], o8, 0);
            else if (c < 12) GL_PRD(f.v[c], x3b, 192 - 16 * c);
            else if (c == 12) GL_PRD(f.v[c], o12, 0);
            else GL_PRD(f.v[c], x3b, XS3 + 256 - 16 * c);
        }
        GL_PIN();
    };
#else
    auto x3 = [&](Fr& f) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            GL_WRI(l_r + 272 * i, 0, f.v[4 * i]);
            GL_WRI(l_r + 68 + 272 * i, 1, f.v[4 * i + 1]);
            GL_WRI(l_m + 272 * i, 0, f.v[4 * i + 2]);
            GL_WRI(l_t + 272 * i, 1, f.v[4 * i + 3]);
        }
#pragma unroll
        for (int c = 0; c < 16; ++c) f.v[c] = GL_RDI(l_b + 17 * c, c);
        GL_PIN();
    };
#endif
    // pass 2 backwards: inverse 16-point DFT, then the conjugate twiddles (tw = T2 again, read once per pair)
    auto m2i = [&](Fr& f) {
        dft16<true>(f.v);
        cplx (&v)[16] = f.v;
        cmul_vc4(v[1], v[2], v[3], v[4], tw2[0], tw2[1], tw2[2], tw2[3]);
        cmul_vc4(v[5], v[6], v[7], v[8], tw2[4], tw2[5], tw2[6], tw2[7]);
        cmul_vc4(v[9], v[10], v[11], v[12], tw2[8], tw2[9], tw2[10], tw2[11]);
        cmul_vc3(v[13], v[14], v[15], tw2[12], tw2[13], tw2[14]);
    };
#ifndef GL_EXP_AOS_EXCHANGE
    // writer: lane l holds y1'[l + 64 i] in register i; reader t wants y1'[16 t + c] = register t >> 2 of lane
    // 16 (t & 3) + c. (Lanes 16 b apart of one plane: 2-way on 32 banks whatever the plane stride; measured against the
    // padded image with single ds_read_b64 reads it is still the faster form, 5.83 against 5.93 ms.)
    const unsigned x4p = lds_m0 + 4u * (unsigned)((lane >> 2) * XS1 + 16 * (lane & 3));
    auto x4 = [&](Fr& f) {
        GL_AT_BEGIN();
#pragma unroll
        for (int i = 0; i < 16; ++i) GL_AT(f.v[i], XS1, i);
#pragma unroll
        for (int c = 0; c < 16; ++c) GL_PRD(f.v[c], x4p, c);
        GL_PIN();
    };
#else
    auto x4 = [&](Fr& f) {
#pragma unroll
        for (int i = 0; i < 16; ++i) GL_WRI(l_r + 68 * i, i, f.v[i]);
#pragma unroll
        for (int c = 0; c < 16; ++c) f.v[c] = GL_RDI(l_17 + c, c);
        GL_PIN();
    };
#endif
    // pass 1 backwards: v[i] = 1024 z'[lane + 64 i]; window, store (the two frames of a pair separately: sharing the
    // window read between them bunches their stores and measured slower, 7.45 against 7.34 ms)
    auto m7 = [&](Fr& f, bool store) {
        dft16<true>(f.v);
        const float sc = 0.5f / FFT_N;                   // (1/1024 of the inverse transform, 1/2 of the merge)
        if (store) {
#pragma unroll
            for (int i = 0; i < 10; ++i) {
                const int c = lane + 64 * i;
                if (i < 9 || c < HALF) {
                    const float2 wv = wreg[i];
                    *reinterpret_cast<float2*>(f.out + 2 * c) = make_float2(f.v[i].x * sc * wv.x, f.v[i].y * sc * wv.y);
                }
            }
        }
    };

    // SOUT: the run's record, written through a buffer resource: a store that may go to the record or to LDS depending
    // on the sample becomes a FLAT store on a selected pointer otherwise (measured: the pass below 3x slower)
    const __amdgpu_buffer_rsrc_t rec_rs = __builtin_amdgcn_make_buffer_rsrc(
        frames_out + (SOUT ? ((long)b * runs_per_utt + ru) * RS : 0), 0, SOUT ? RS * 4 : 0, 0x00020000);
    auto rec_store2 = [&](int fo, float2 val) {           // record[fo], record[fo + 1]
        typedef unsigned u2v __attribute__((ext_vector_type(2)));
        __builtin_amdgcn_raw_buffer_store_b64(u2v{__builtin_bit_cast(unsigned, val.x), __builtin_bit_cast(unsigned, val.y)},
                                              rec_rs, 4u * (unsigned)fo, 0, 0);
    };
    auto rec_store4 = [&](int fo, f4v val) {
        typedef unsigned u4v __attribute__((ext_vector_type(4)));
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u4v, val), rec_rs, 4u * (unsigned)fo, 0, 0);
    };
    // SOUT: where the sum of sample s (run coordinates, s >= HEAD) is built up
    auto dest = [&](int s) {
        return (unsigned)(s - 2 * GL_WAVES * HOP) < (unsigned)HEAD ? side_s + (s - 2 * GL_WAVES * HOP) : yseg + s;
    };
    // pass 1 backwards, then the windowed frame goes three ways. (1) Its last HOP samples, to which it is the first
    // contributor of the run: plain store at dest (every input of this round has been read: barrier after a1).
    // (2) Its first HEAD samples: into the wave's exchange buffer (dead after the last exchange; slot = A / B), from
    // where the pass below adds them in frame order. (3) MAYHEAD (frames 0..2 of the run): what falls into the run's
    // first HEAD samples goes to the record raw instead (the sums there start with the previous run's frames).
    auto m7s = [&](Fr& f, int fidx, int slot, bool store, bool mayhead) {
        dft16<true>(f.v);
        if (!store) return;
        const float sc = 0.5f / FFT_N;
        // (the lane's addresses below are formed here, from a copy of the lane number the compiler cannot see through:
        // as loop invariants they would be computed once and held in registers over the whole transform)
        int lane = lane_;
        asm volatile("" : "+v"(lane));
        float* pv = reinterpret_cast<float*>(lds) + slot * HEAD;
        if (mayhead && fidx < 3) {
            const int hraw = fidx == 0 ? 0 : (fidx == 1 ? HOFF1 : HOFF2);
            const int th = HEAD - HOP * fidx;
#pragma unroll
            for (int i = 0; i < 10; ++i) {
                const int c = lane + 64 * i, o = 2 * c;
                if (i < 9 || c < HALF) {
                    const float2 wv = wreg[i];
                    const float2 val = make_float2(f.v[i].x * sc * wv.x, f.v[i].y * sc * wv.y);
                    if (o < th) rec_store2(hraw + o, val);
                    else if (o < HEAD) *reinterpret_cast<float2*>(pv + o) = val;
                    else *reinterpret_cast<float2*>(yseg + HOP * fidx + o) = val;     // (fidx < 3: never the side buffer)
                }
            }
        } else {
            // whether the frame's last HOP samples (hop fidx + 3) lie in the side buffer (hops 8..10 of the run) is the
            // same for all of them: one scalar base instead of a test per lane and element
            static_assert(HEAD == 3 * HOP && 2 * GL_WAVES * HOP % HOP == 0, "dest() is uniform over a frame's last hop");
            const bool sd = (unsigned)(fidx + 3 - 2 * GL_WAVES) < (unsigned)(HEAD / HOP);
            float* const tl = (sd ? side_s - 2 * GL_WAVES * HOP : yseg) + HOP * fidx + 2 * lane;
            float* const pl = pv + 2 * lane;
#pragma unroll
            for (int i = 0; i < 10; ++i) {
                const int c = lane + 64 * i;
                if (i < 9 || c < HALF) {
                    const float2 wv = wreg[i];
                    const float2 val = make_float2(f.v[i].x * sc * wv.x, f.v[i].y * sc * wv.y);
                    if (128 * i + 126 < HEAD) *reinterpret_cast<float2*>(pl + 128 * i) = val;
                    else if (128 * i >= HEAD) *reinterpret_cast<float2*>(tl + 128 * i) = val;
                    else *reinterpret_cast<float2*>((2 * c < HEAD ? pl : tl) + 128 * i) = val;
                }
            }
        }
    };
    // after a round of eight frames F0..F1-1 (all waves' m7s behind a barrier): the samples they reach, four a thread.
    // acc = what the sample's first contributor stored (or the first round left), + the round's other contributions
    // in ascending frame order; complete sums (and the run's last HEAD samples' prefixes) go to the record, the
    // samples the next round adds to go back to LDS.
    auto pass = [&](int rd) {
        const int F0 = 2 * GL_WAVES * rd, F1 = min(F0 + 2 * GL_WAVES, nfr);
        const int qlo = rd ? F0 : HEAD / HOP, qhi = F1 + HEAD / HOP;      // the pass covers hops qlo .. qhi - 1
        const int npos = (qhi - qlo) * (HOP / 4);                          // ... in groups of four samples
        const float* pbase = reinterpret_cast<const float*>(fft_s);
        int lane = lane_;
        asm volatile("" : "+v"(lane));
        // every read of a thread is issued before its first sum (beside a transforming neighbour an LDS round trip
        // takes the better part of a microsecond); a contribution the round does not have reads a zero instead, so
        // the sums need no conditions (x + 0 = x). A pass is about 250 instructions a wave: with one or two waves
        // on a SIMD an instruction issues every four to five cycles, and that, not the data, is what it costs.
        constexpr int NP = ((2 * GL_WAVES + HEAD / HOP) * (HOP / 4) + GL_THREADS - 1) / GL_THREADS;
        f4v a[NP], c[NP][3];
        int sidx[NP];
        bool fin[NP];
#pragma unroll
        for (int u = 0; u < NP; ++u) {
            const int p = min(lane + 64 * wave + GL_THREADS * u, npos - 1);
            const int h = p / (HOP / 4), q = qlo + h;
            const int off = 4 * (p - h * (HOP / 4));
            sidx[u] = HOP * q + off;
            fin[u] = q < F1 || F1 == nfr;
            a[u] = *reinterpret_cast<const f4v*>(dest(sidx[u]));
#pragma unroll
            for (int d = 0; d < 3; ++d) {
                const int fq = q - 2 + d;                                   // (sample HOP q of the run = sample
                const bool ok = fq >= F0 && fq < F1;                        // HOP (2 - d) of frame fq)
                const float* cp = pbase + (fq & 3) * (2 * GL2_LDS) + ((fq >> 2) & 1) * HEAD + HOP * (2 - d) + off;
                c[u][d] = *reinterpret_cast<const f4v*>(ok ? cp : zero_s);
            }
        }
#pragma unroll
        for (int u = 0; u < NP; ++u) {
            if (lane + 64 * wave + GL_THREADS * u < npos) {
                const f4v acc = ((a[u] + c[u][0]) + c[u][1]) + c[u][2];
                if (fin[u]) rec_store4(HEAD + sidx[u], acc);
                else *reinterpret_cast<f4v*>(dest(sidx[u])) = acc;
            }
        }
    };

    // The same pass for a FULL round of eight frames (all rounds of a full run: every run of an utterance but its
    // last), written so that almost nothing of it is per-lane arithmetic: thread t < 225 takes the hops Q0 + 3 u + h0
    // (h0 = t / 75, u = 0..3; Q0 = 2 / 8 in round 0 / 1), four samples at t % 75 of each. With that dealing every unit u
    // is uniform in what the generic pass decides per lane: where the sum is built up (hops 8..10 of the run: the side
    // buffer), whether it is complete (record) or goes on in the next round (LDS), and the place of a contributing
    // frame's buffer is one of three constants chosen by h0. Same sums in the same order, same zeros for the
    // contributions a round does not have: same bits (217 -> ~80 VALU instructions a wave and round).
    auto pass_full = [&](auto rdc) {
        constexpr int RD = decltype(rdc)::value;
        static_assert(HEAD == 3 * HOP && GL_WAVES == 4 && HOP % 4 == 0, "pass_full: eight frames a round, three head hops");
        constexpr int F0 = 8 * RD, Q0 = RD ? 8 : 2, NU = RD ? 4 : 3, HQ = HOP / 4;
        const bool last = F0 + 8 == nfr;                 // (round 1 of a full run: always)
        int t = lane_ + 64 * wave;
        asm volatile("" : "+v"(t));
        const bool act = t < 3 * HQ;
        t = min(t, 3 * HQ - 1);
        const int h0 = (t >= HQ) + (t >= 2 * HQ);
        const int lp = HOP * h0 + 4 * (t - HQ * h0);     // the lane's part of a sample index: hop h0, group t % 75
        const int o4 = 4 * (t - HQ * h0);
        const float* pbase = reinterpret_cast<const float*>(fft_s);
        const int zoff = (int)(zero_s - pbase);
        f4v a[NU], c[NU][3];
#pragma unroll
        for (int u = 0; u < NU; ++u) {
            const int qb = Q0 + 3 * u;                   // hop of the lanes with h0 = 0
            const bool side = qb == 2 * GL_WAVES;        // hops 8, 9, 10
            const float* ab = side ? side_s + (HOP * qb - 2 * GL_WAVES * HOP) : yseg + HOP * qb;
            a[u] = *reinterpret_cast<const f4v*>(ab + lp);
#pragma unroll
            for (int d = 0; d < 3; ++d) {
                // frame fq = qb + h0 - 2 + d of the run, g = fq - F0 of the round: buffer of wave g & 3, slot g >> 2
                const int g0 = qb - 2 + d - F0;
                int off[3];
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    const int g = g0 + k;
                    off[k] = (g >= 0 && g < 8) ? (g & 3) * (2 * GL2_LDS) + ((g >> 2) & 1) * HEAD + HOP * (2 - d) : zoff;
                }
                const int sel = h0 == 0 ? off[0] : (h0 == 1 ? off[1] : off[2]);
                c[u][d] = *reinterpret_cast<const f4v*>(pbase + sel + o4);
            }
        }
#pragma unroll
        for (int u = 0; u < NU; ++u) {
            const int qb = Q0 + 3 * u;
            const bool side = qb == 2 * GL_WAVES;
            const bool ok = act && qb + h0 >= (RD ? F0 : HEAD / HOP) && qb + h0 < F0 + 8 + HEAD / HOP;
            const bool fin = last || qb + 2 < F0 + 8;    // (uniform: the units' hops do not straddle the round's end)
            const f4v acc = ((a[u] + c[u][0]) + c[u][1]) + c[u][2];
            if (ok) {
                if (fin) rec_store4(HEAD + HOP * qb + lp, acc);
                else *reinterpret_cast<f4v*>((side ? side_s + (HOP * qb - 2 * GL_WAVES * HOP) : yseg + HOP * qb) + lp) = acc;
            }
        }
    };

    (void)pass_full;
    // frames of this wave: wave, wave + 4, ...; two at a time, half a stage apart: while one frame's exchange is in
    // flight the other frame computes. A last, single frame runs as a pair with itself (the copy is not stored).
    // SOUT: a round (the workgroup's next eight frames) is bracketed by workgroup barriers, so every wave takes every
    // round, with or without a frame of its own in it.
#ifdef GL_EXP_ADDTID
    asm volatile("s_mov_b32 m0, %0" :: "s"((unsigned)(size_t)lds) : "memory");
#endif
    const int nrounds = SOUT ? (nfr + 2 * GL_WAVES - 1) / (2 * GL_WAVES) : 1;
    for (int rd = 0; rd < nrounds; ++rd) {
        for (int fi = wave + (SOUT ? 2 * GL_WAVES * rd : 0); fi < (SOUT ? min(nfr, 2 * GL_WAVES * (rd + 1)) : nfr);
             fi += 2 * GL_WAVES) {
            Fr A, Bf;
            const bool pair = fi + GL_WAVES < nfr;
            s0_load(A, fi);
            s0_load(Bf, pair ? fi + GL_WAVES : fi);
            const int sb = 4 + 10 * (fi >= 2 * GL_WAVES);
            (void)sb;
            GL_STAMP(sb);
            a1(A, Bf);
            if constexpr (SOUT) gl_lds_barrier();           // (B1) every input of the round is in registers
#ifdef GL_EXP_SINGLE   // (experiment: the two frames one after the other, no interleave: what one frame costs a wave)
            const int fn = fi + 2 * GL_WAVES;
            d1(A); x_pass1(A); GL2_LAND(A.v); m2f(A); dft16<false>(A.v); x_pass2(A); GL2_LAND(A.v); m3(A, mkA, mnA, m512A); x3(A); GL2_LAND(A.v);
            if (fn < nfr) load_mag(mkA, mnA, m512A, fn);
            m2i(A); x4(A);
            if constexpr (SOUT) m7s(A, fi, 0, true, true); else m7(A, true);
            d1(Bf); x_pass1(Bf); GL2_LAND(Bf.v); m2f(Bf); dft16<false>(Bf.v); x_pass2(Bf); GL2_LAND(Bf.v); m3(Bf, mkB, mnB, m512B); x3(Bf); GL2_LAND(Bf.v);
            if (fn < nfr) load_mag(mkB, mnB, m512B, min(fn + GL_WAVES, nfr - 1));
            m2i(Bf); x4(Bf);
            if constexpr (SOUT) m7s(Bf, fi + GL_WAVES, 1, pair, false); else m7(Bf, pair);
#else
            d1(A);   x_pass1(A);
            d1(Bf);  GL2_LAND(A.v);  x_pass1(Bf);
            GL_STAMP(sb + 1);
            m2f(A);  dft16<false>(A.v);  GL2_LAND(Bf.v); x_pass2(A);
            m2f(Bf); dft16<false>(Bf.v); GL2_LAND(A.v);  x_pass2(Bf);
            GL_STAMP(sb + 2);
            const int fn = fi + 2 * GL_WAVES;               // the wave's next pair
            m3(A, mkA, mnA, m512A);   GL2_LAND(Bf.v); x3(A);
            if (fn < nfr) load_mag(mkA, mnA, m512A, fn);
            GL_STAMP(sb + 3);
            m3(Bf, mkB, mnB, m512B);  GL2_LAND(A.v);  x3(Bf);
            if (fn < nfr) load_mag(mkB, mnB, m512B, min(fn + GL_WAVES, nfr - 1));
            GL_STAMP(sb + 4);
            m2i(A);  GL2_LAND(Bf.v); x4(A);
            GL_STAMP(sb + 5);
            m2i(Bf); GL2_LAND(A.v);  x4(Bf);
            GL_STAMP(sb + 6);
            if constexpr (SOUT) {
                // (A's slot of the exchange buffer is written behind B's reads of it: the LDS takes a wave's
                // operations in order)
                m7s(A, fi, 0, true, true);
                GL2_LAND(Bf.v);
                m7s(Bf, fi + GL_WAVES, 1, pair, false);
            } else {
                m7(A, true);
                GL2_LAND(Bf.v);
                m7(Bf, pair);
            }
#endif
            GL_STAMP(sb + 7);
        }
        if constexpr (SOUT) {
            // a wave without a frame in this round (a short last run) still owes the round's barriers
            if (wave + 2 * GL_WAVES * rd >= nfr) gl_lds_barrier();      // (B1)
            gl_lds_barrier();                                           // (B2) the round's contributions are in LDS
            GL_STAMP(24 + 3 * rd);
#ifndef GL_EXP_NO_PASS   // (experiment, wrong results: what the summation pass costs)
#ifndef GL_EXP_OLD_PASS
            if (min(nfr, 2 * GL_WAVES * (rd + 1)) - 2 * GL_WAVES * rd == 2 * GL_WAVES && rd < 2) {
                if (rd == 0) pass_full(std::integral_constant<int, 0>());
                else pass_full(std::integral_constant<int, 1>());
            } else
#endif
                pass(rd);
#endif
            GL_STAMP(25 + 3 * rd);
            if (rd + 1 < nrounds) gl_lds_barrier();                     // (B3) before the buffers are written again
            GL_STAMP(26 + 3 * rd);
        }
    }
}

// More than the default 64 KB of dynamic LDS for gl_iter: the attribute is per device. Called at taco_create (so that
// an entry point that must not do set-up work finds it done) and, idempotently, by every launch.
void gl_prepare_device() {
    static bool attr_set[64] = {};
    int dev = 0;
    if (hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < 64 && !attr_set[dev]) {
        for (auto* k : {gl_iter_kernel<false, 0, 0>, gl_iter_kernel<true, 0, 0>, gl_iter_kernel<false, 1200, 300>,
                        gl_iter_kernel<true, 1200, 300>})
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize,
                                      160 * 1024);
        for (auto* k : {gl_iter2_kernel<1200, 300, false, false>, gl_iter2_kernel<1200, 300, false, true>,
                        gl_iter2_kernel<1200, 300, true, true>})
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize,
                                      160 * 1024);
        attr_set[dev] = true;
    }
}

#ifdef GL_STAMPS
extern "C" int taco_debug_gl_stamps(unsigned long long* out64) {
    return hipMemcpyFromSymbol(out64, HIP_SYMBOL(gl_dbg), sizeof(unsigned long long) * 64) == hipSuccess ? 0 : -1;
}
#endif

// [window | T2 (fft_tw_entry), T3 and split twiddles in gl_iter2_kernel's lane order] as floats, the order it carves its LDS in
std::vector<float> gl_lds_image(const std::vector<float>& tw, const std::vector<float>& win) {
    std::vector<float> img(win);
    auto W = [&](int j) { return std::pair<float, float>(tw[2 * j], tw[2 * j + 1]); };
    for (int e = 0; e < FFT_TW; ++e) {
        int j;
        if (e < FFT_T2) j = 8 * (e % 16) * (e / 16 + 1);
        else {
            // radix-4 pass: entry 64 s + l of a row belongs to butterfly t_s(l) of lane l (gl_iter2_kernel, x_pass2)
            const int e2 = e - FFT_T2, s = (e2 % 256) / 64, l = e2 % 64;
            const int t = s == 0 ? l : (s == 1 ? 64 + l : (s == 2 ? 192 - l : (l ? 256 - l : 128)));
            j = 2 * t * (e2 / 256 + 1);
        }
        img.push_back(W(j).first);
        img.push_back(W(j).second);
    }
    // split / merge twiddles W[k] of the lane's bin pairs, entry l + 64 j = pair j of lane l (pair_bin in the kernel)
    for (int e = 0; e < FFT_N / 2; ++e) {
        const int l = e % 64, t3 = l ? 256 - l : 128;
        const int ks[8] = {l, 256 + l, t3, 256 + t3, 64 + l, 320 + l, 192 - l, 448 - l};
        const int k = ks[e / 64];
        img.push_back(W(k).first);
        img.push_back(W(k).second);
    }
    while (img.size() % 4) img.push_back(0.f);
    return img;
}

// Frames per workgroup of gl_iter2_kernel for a job of B utterances of T frames: whole pairs per wave only (8 or 16
// frames); the shorter run when it does not cost another round of workgroups over the chip's 512 places (two per CU):
// a workgroup lives for about one gather plus one transform time per pair, and a gather takes about as long as a pair.
void gl_read_switches(GLPlan& p) {
    p.one_frame = getenv("TACO_GL_ONE_FRAME") != nullptr;
    const char* e = getenv("TACO_GL_FRAME_STATE");
    p.frame_state = e ? (atoi(e) == 0 ? 0 : 1) : -1;
    e = getenv("TACO_GL_RUN");
    const int r = e ? atoi(e) : 0;
    p.run = (r == 8 || r == 16) ? r : 0;
}

static int gl2_run(const GLPlan& p, int B, int T) {
    if (p.run) return p.run;                             // experiments: 8 or 16
    const long slots = 512;
    auto rounds = [&](int r) { return ((long)B * ((T + r - 1) / r) + slots - 1) / slots; };
    return rounds(8) * 2 <= rounds(16) * 3 ? 8 : GL_RUN;
}
// Run records need every run to hold the three raw head frames: a last run of one or two frames keeps the frame state
// (same bits either way, so the choice may depend on the batch). They pay when the launch is bound by what it moves,
// not when its few workgroups have a CU each and the round barriers only lengthen them: GL2_RECORDS_MIN_GRID.
// TACO_GL_FRAME_STATE=1 keeps the frame state always, =0 takes records whenever the geometry allows (tests compare).
constexpr long GL2_RECORDS_MIN_GRID = 257;
static bool gl2_records_ok(const GLPlan& p, int T, int run, long grid) {
    const int last = T % run;
    if (!(last == 0 || last >= 3)) return false;
    // the records of an utterance must fit where its frames would lie (the caller's buffers are sized for frames):
    // true from T = 4 (runs of 8) / T = 6 (runs of 16) on
    const long runs_per_utt = (T + run - 1) / run;
    if (runs_per_utt * (2 * (1200 - 300) + 300 * run) > (long)T * 1200) return false;
    if (p.frame_state >= 0) return p.frame_state == 0;
    return grid >= GL2_RECORDS_MIN_GRID;
}

int launch_gl_iter(const GLPlan& p, const float* frames_in, const float* mag, float* frames_out, int B, int T,
                   int first, hipStream_t s, Timing* tm, bool librosa, const float* phase_u, int* state) {
    // frames per workgroup: as many as keep the y segment within ~24 KB of LDS (2 workgroups per CU and more)
    int run = GL_RUN;
    while (run > 1 && ((size_t)(run - 1) * p.hop + p.win) * 4 > 24 * 1024) --run;
    const bool dflt = p.win == 1200 && p.hop == 300;
    // the iteration of the TF flavour at the default geometry runs two frames per wave (gl_iter2_kernel); the choice
    // depends on the geometry and the flavour only, never on the batch, so an utterance's result does not depend on
    // what it is batched with
    const bool two = dflt && !librosa && !first && run == GL_RUN && p.lds_image && !p.one_frame;
    // small jobs: shorter runs (down to one frame, or one pair of frames, per wave) until the grid covers the chip
    // twice; the overlap-add order per sample does not depend on the run length, so results are bit-identical
    const int sin = state ? *state : 0;                  // what frames_in holds: 0 = frames, else run records of that run length
    // run records can only be read by the two-frame kernel; within a chain (same B, T, flavour and switches
    // throughout) they never reach the other one
    if (sin && !two) return -1;
    if (tm) tm->begin(KK_GL_ITER, s);
    if (two) {
        run = sin ? sin : gl2_run(p, B, T);
    } else {
        while (run > GL_WAVES && (long)B * ((T + run - 1) / run) < 512) run -= GL_WAVES;
    }
    const int runs_per_utt = (T + run - 1) / run;
    if (!p.prepared) gl_prepare_device();
    const dim3 grid((unsigned)(B * runs_per_utt));
    if (two) {
        // the state written: run records when the caller can take them (it passes `state`) and the geometry allows
        const int sout = sin || (state && gl2_records_ok(p, T, run, (long)B * runs_per_utt)) ? 1 : 0;
        const size_t lds = (size_t)(((run - 1) * p.hop + p.win + 3) & ~3) * 4 + (size_t)p.win * 4 +
                           (size_t)(FFT_TW + FFT_N / 2 + GL_WAVES * GL2_LDS) * sizeof(cplx) +
                           (sout ? (size_t)(p.win - p.hop) * 4 + (size_t)p.hop * 4 : 0)
#ifdef GL_EXP_ONE_WG                         // (experiment: dynamic LDS padded so that only one workgroup fits a CU)
                           + 8192
#endif
            ;
        auto* kern = sin ? gl_iter2_kernel<1200, 300, true, true>
                         : (sout ? gl_iter2_kernel<1200, 300, false, true> : gl_iter2_kernel<1200, 300, false, false>);
        // (records in -> records out always: eligibility depends on T and the run only, which a chain does not change)
        hipLaunchKernelGGL(kern, grid, dim3(GL_THREADS), lds, s, frames_in, mag, frames_out, p.lds_image, B, T,
                           p.nbins, run, runs_per_utt);
        if (state) *state = sout ? run : 0;
    } else {
        if (state) *state = 0;
        const size_t lds = (size_t)((((run - 1) * p.hop + p.win + 3) & ~3) + ((p.win + 3) & ~3)) * 4 +
                           (size_t)(FFT_TW + FFT_N / 2 + GL_WAVES * FFT_LDS) * sizeof(cplx);
        auto* kern = librosa ? (dflt ? gl_iter_kernel<true, 1200, 300> : gl_iter_kernel<true, 0, 0>)
                             : (dflt ? gl_iter_kernel<false, 1200, 300> : gl_iter_kernel<false, 0, 0>);
        hipLaunchKernelGGL(kern, grid, dim3(GL_THREADS), lds, s, frames_in, mag, frames_out, (const float2*)p.twiddle,
                           p.window, phase_u, B, T, p.hop, p.win, p.nbins, first, run, runs_per_utt);
    }
    if (tm) tm->end(KK_GL_ITER, s);
    return 0;
}

// ---------------------------------------------------------------------------------------
// Analysis side (tools/audio.py:23-42,66-79): D = librosa.stft(pre_emphasis(y), n_fft, hop, win, 'hann') with
// center=True: the pre-emphasised signal is reflect-padded by n_fft/2 on both sides, frame t covers padded samples
// [t hop, t hop + n_fft) and is weighted by the periodic Hann window of length win zero-padded (centred) to n_fft.
// One wave per frame, the same register FFT as gl_iter: |D| goes to `mag` (the mel projection's GEMM operand),
// clip((20 log10(max(1e-5, |D|)) - ref - min) / -min, 0, 1) to `lin`.
__global__ __launch_bounds__(GL_THREADS, 2) void stft_center_kernel(const float* __restrict__ wav,
                                                                    const float2* __restrict__ twiddle,
                                                                    const float* __restrict__ window, int B, int n, int T,
                                                                    int hop, int win, int nbins, float preemph,
                                                                    float ref_db, float min_db, float* __restrict__ mag,
                                                                    int ldm, float* __restrict__ lin) {
    extern __shared__ __attribute__((aligned(16))) unsigned char gl_smem[];
    cplx* tw_s = reinterpret_cast<cplx*>(gl_smem);
    const cplx* T2 = tw_s;
    const cplx* T3 = tw_s + FFT_T2;
    cplx* wk_s = tw_s + FFT_TW;
    cplx* fft_s = wk_s + FFT_N / 2;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    cplx* lds = fft_s + wave * FFT_LDS;
    const cplx* Wt = reinterpret_cast<const cplx*>(twiddle);
    for (int e = tid; e < FFT_TW; e += GL_THREADS) tw_s[e] = fft_tw_entry(e, Wt);
    for (int e = tid; e < FFT_N / 2; e += GL_THREADS) wk_s[e] = Wt[e];
    __syncthreads();
    const int n_fft = 2 * FFT_N, lp = (n_fft - win) / 2;
    const long nframes = (long)B * T;
    for (long f = (long)blockIdx.x * GL_WAVES + wave; f < nframes; f += (long)gridDim.x * GL_WAVES) {
        const int b = (int)(f / T), t = (int)(f - (long)b * T);
        const float* x = wav + (size_t)b * n;
        auto sample = [&](int pidx) -> float {          // padded position -> pre-emphasised, windowed sample
            const int wi = pidx - lp;
            if (wi < 0 || wi >= win) return 0.f;
            int j = t * hop + pidx - n_fft / 2;         // position in the unpadded signal
            if (j < 0) j = -j;                          // np.pad(mode='reflect')
            if (j >= n) j = 2 * (n - 1) - j;
            const float v = x[j] - (j > 0 ? preemph * x[j - 1] : 0.f);   // lfilter([1, -a], [1], x), zero state
            return v * window[wi];
        };
        cplx v[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int c = lane + 64 * i;
            v[i] = {sample(2 * c), sample(2 * c + 1)};
        }
        fft_pass1<false>(v, lds, lane);
        __builtin_amdgcn_wave_barrier();
        fft_load(v, lds, lane);
        __builtin_amdgcn_wave_barrier();
        fft_pass2<false>(v, lds, lane, T2);
        __builtin_amdgcn_wave_barrier();
        fft_load(v, lds, lane);
        __builtin_amdgcn_wave_barrier();
        fft_pass3<false>(v, lane, T3);
#pragma unroll
        for (int i = 8; i < 16; ++i) lds[fft_pad(lane + 64 * i)] = v[i];
        __builtin_amdgcn_wave_barrier();
        float* mrow = mag + (size_t)f * ldm;
        float* lrow = lin ? lin + (size_t)f * nbins : nullptr;
        auto emit = [&](int k, float a) {
            mrow[k] = a;
            if (lrow) lrow[k] = fminf(fmaxf((20.f * log10f(fmaxf(1e-5f, a)) - ref_db - min_db) / -min_db, 0.f), 1.f);
        };
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int k = lane + 64 * j;
            if (k == 0) {
                emit(0, fabsf(v[0].x + v[0].y));
                emit(FFT_N, fabsf(v[0].x - v[0].y));
                emit(FFT_N / 2, sqrtf(v[8].x * v[8].x + v[8].y * v[8].y));
            } else {
                cplx Xk, Xn;
                rfft_split(v[j], lds[fft_pad(FFT_N - k)], wk_s[k], Xk, Xn);
                emit(k, sqrtf(Xk.x * Xk.x + Xk.y * Xk.y));
                emit(FFT_N - k, sqrtf(Xn.x * Xn.x + Xn.y * Xn.y));
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
}

void launch_stft_center(const GLPlan& p, const float* wav, int B, int n, int T, float preemph, float ref_db,
                        float min_db, float* mag, int ldm, float* lin, hipStream_t s, Timing* tm) {
    if (tm) tm->begin(KK_GL_ITER, s);
    const long nframes = (long)B * T;
    const int blocks = (int)std::min<long>((nframes + GL_WAVES - 1) / GL_WAVES, 2048);
    const size_t lds = (size_t)(FFT_TW + FFT_N / 2 + GL_WAVES * FFT_LDS) * sizeof(cplx);
    hipLaunchKernelGGL(stft_center_kernel, dim3(blocks), dim3(GL_THREADS), lds, s, wav, (const float2*)p.twiddle,
                       p.window, B, n, T, p.hop, p.win, p.nbins, preemph, ref_db, min_db, mag, ldm, lin);
    if (tm) tm->end(KK_GL_ITER, s);
}

// wav[b][g] = sum of the frames covering sample g (tf overlap_and_add), g < (T-1)*hop + win
__global__ void gl_ola_kernel(const float* __restrict__ frames, float* __restrict__ wav, int B, int T, int hop, int win) {
    const long L = (long)(T - 1) * hop + win;
    const int nover = (win + hop - 1) / hop;
    const long total = (long)B * L;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int b = (int)(i / L);
        const int g = (int)(i - (long)b * L);
        const int mhi = min(T - 1, g / hop);
        const int mlo = max(0, g / hop - (nover - 1));
        float s = 0.f;
        for (int m2 = mlo; m2 <= mhi; ++m2) {
            const int off = g - hop * m2;
            if (off < win) s += frames[((long)b * T + m2) * win + off];
        }
        wav[i] = s;
    }
}

// librosa istft output (the oracle's istft_librosa): wav[b][j] = normalised overlap-add at frame coordinate j + win/2,
// j < hop (T-1)
__global__ void gl_ola_norm_kernel(const float* __restrict__ frames, const float* __restrict__ window,
                                   float* __restrict__ wav, int B, int T, int hop, int win) {
    const long L = (long)(T - 1) * hop;
    const int nover = (win + hop - 1) / hop;
    const long total = (long)B * L;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int b = (int)(i / L);
        const int j = (int)(i - (long)b * L);
        wav[i] = gl_ola_norm(frames + (long)b * T * win, window, j + (win >> 1), T, hop, win, nover);
    }
}

void launch_gl_ola_norm(const GLPlan& p, const float* frames, float* wav, int B, int T, hipStream_t s, Timing* tm) {
    if (tm) tm->begin(KK_GL_OLA, s);
    const long total = (long)B * (long)(T - 1) * p.hop;
    const int blocks = (int)std::min<long>(4096, (total + 255) / 256);
    hipLaunchKernelGGL(gl_ola_norm_kernel, dim3(blocks), dim3(256), 0, s, frames, p.window, wav, B, T, p.hop, p.win);
    if (tm) tm->end(KK_GL_OLA, s);
}

// the same sums (same order per sample) four samples a thread: hop and window multiples of 4, so four consecutive
// samples starting at a multiple of 4 are covered by the same frames and every access is a 16-byte one
__global__ void gl_ola4_kernel(const float* __restrict__ frames, float* __restrict__ wav, int B, int T, int hop, int win) {
    const long L = (long)(T - 1) * hop + win;
    const int nover = (win + hop - 1) / hop;
    const long total4 = (long)B * L / 4;
    for (long i4 = blockIdx.x * (long)blockDim.x + threadIdx.x; i4 < total4; i4 += (long)gridDim.x * blockDim.x) {
        const long i = 4 * i4;
        const int b = (int)(i / L);
        const int g = (int)(i - (long)b * L);
        const int mhi = min(T - 1, g / hop);
        const int mlo = max(0, g / hop - (nover - 1));
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int m2 = mlo; m2 <= mhi; ++m2) {
            const int off = g - hop * m2;
            if (off < win) {
                const float4 v = *reinterpret_cast<const float4*>(frames + ((long)b * T + m2) * win + off);
                acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
            }
        }
        *reinterpret_cast<float4*>(wav + i) = acc;
    }
}

// The signal from run records (gl_iter2_kernel): a sample is the own sum of the run whose SUMS hold it (+ the next
// run's raw head frames where it lies in that run's last WIN-HOP samples), or, for the utterance's first WIN-HOP
// samples, the first run's raw head frames alone: the additions of the iteration's gather, i.e. of gl_ola_kernel.
template <int WIN, int HOP, bool VEC>
__global__ void gl_ola_rec_kernel(const float* __restrict__ recs, float* __restrict__ wav, int B, int T, int run,
                                  int runs_per_utt) {
    constexpr int HEAD = WIN - HOP, HOFF1 = HEAD, HOFF2 = 2 * HEAD - HOP;
    typedef float f4v __attribute__((ext_vector_type(4)));
    const int RS = 2 * HEAD + HOP * run;
    const long L = (long)(T - 1) * HOP + WIN;
    const long total4 = (long)B * L / 4;
    for (long i4 = blockIdx.x * (long)blockDim.x + threadIdx.x; i4 < total4; i4 += (long)gridDim.x * blockDim.x) {
        const long i = 4 * i4;
        const int b = (int)(i / L);
        const int g = (int)(i - (long)b * L);
        const float* ur = recs + (long)b * runs_per_utt * RS;
        f4v acc = {0.f, 0.f, 0.f, 0.f};
        const float* hp = nullptr;
        int j = g;
        if (g < HEAD) {
            hp = ur;
        } else {
            const int ru = min((g - HEAD) / (HOP * run), runs_per_utt - 1);
            const int sidx = g - HOP * run * ru;                      // >= HEAD
            const float* rec = ur + (long)ru * RS;
            acc = *reinterpret_cast<const f4v*>(rec + HEAD + sidx);
            j = sidx - HOP * run;                                     // >= 0: the run is full and the sample in its tail
            if (j >= 0 && ru + 1 < runs_per_utt) hp = rec + RS;
        }
        if (hp) {
            acc += *reinterpret_cast<const f4v*>(hp + j);
            if (j >= HOP) acc += *reinterpret_cast<const f4v*>(hp + HOFF1 + j - HOP);
            if (j >= 2 * HOP) acc += *reinterpret_cast<const f4v*>(hp + HOFF2 + j - 2 * HOP);
        }
        if (VEC) *reinterpret_cast<f4v*>(wav + i) = acc;
        else {
            wav[i] = acc.x; wav[i + 1] = acc.y; wav[i + 2] = acc.z; wav[i + 3] = acc.w;
        }
    }
}

void launch_gl_ola(const GLPlan& p, const float* frames, float* wav, int B, int T, hipStream_t s, Timing* tm,
                   int state) {
    if (tm) tm->begin(KK_GL_OLA, s);
    if (state) {                                       // run records (default geometry only, see launch_gl_iter)
        const int run = state;
        const int runs_per_utt = (T + run - 1) / run;
        const long L4 = (long)B * ((long)(T - 1) * p.hop + p.win) / 4;
        const int blocks = (int)std::max<long>(1, std::min<long>(4096, (L4 + 255) / 256));
        auto* kern = (reinterpret_cast<size_t>(wav) & 15) == 0 ? gl_ola_rec_kernel<1200, 300, true>
                                                               : gl_ola_rec_kernel<1200, 300, false>;
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, s, frames, wav, B, T, run, runs_per_utt);
        if (tm) tm->end(KK_GL_OLA, s);
        return;
    }
    const long L = (long)(T - 1) * p.hop + p.win;
    const long total = (long)B * L;
    const bool vec = p.hop % 4 == 0 && p.win % 4 == 0 && L % 4 == 0 &&
                     ((reinterpret_cast<size_t>(frames) | reinterpret_cast<size_t>(wav)) & 15) == 0;
    if (vec) {
        const int blocks = (int)std::min<long>(4096, (total / 4 + 255) / 256);
        hipLaunchKernelGGL(gl_ola4_kernel, dim3(std::max(blocks, 1)), dim3(256), 0, s, frames, wav, B, T, p.hop, p.win);
    } else {
        const int blocks = (int)std::min<long>(4096, (total + 255) / 256);
        hipLaunchKernelGGL(gl_ola_kernel, dim3(blocks), dim3(256), 0, s, frames, wav, B, T, p.hop, p.win);
    }
    if (tm) tm->end(KK_GL_OLA, s);
}

}  // namespace taco
