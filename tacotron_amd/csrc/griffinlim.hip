// Griffin-Lim as TF builds it in the reference (tools/audio.py:57-64,183-205):
//   mag = (10 ** ((clip(S,0,1)*100 - 100 + 20) * 0.05)) ** 1.5
//   y = istft(mag);  60 x { E = stft(y); y = istft(mag * E / max(1e-8, |E|)) }
// with stft = frame(1200, hop 300, no padding) * periodicHann -> rfft(2048, zero-padded at the
// end) and istft = irfft(2048)[:1200] * periodicHann -> overlap-add (no window-sum division).
//
// One kernel per iteration (gl_iter): a 256-thread workgroup owns one frame at a time -
// it rebuilds the frame's 1200 input samples by overlap-adding the <=4 frames of the previous
// iteration that cover them (fixed order, so results are reproducible), windows, runs the
// 2048-point real FFT in LDS (1024-point complex radix-4 Stockham + split), re-phases against
// the target magnitude, runs the inverse and writes its windowed 1200 output samples.  The
// signal y itself is never materialised between iterations; gl_ola builds it once at the end.
// HBM/L2 traffic per frame and iteration: 4.1 KB magnitudes + <=19.2 KB frame reads + 4.8 KB
// frame writes.
#include "kernels.h"
#include "fft_core.h"

namespace taco {

__global__ void gl_mag_kernel(const float* __restrict__ lin, float* __restrict__ mag, size_t n, float min_db,
                              float ref_db, float power) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        float s = fminf(fmaxf(lin[i], 0.f), 1.f);
        s = s * -min_db + min_db;                       // _denormalize_tensorflow (audio.py:276-277)
        s = powf(10.f, (s + ref_db) * 0.05f);           // _db_to_amp_tensorflow (audio.py:226-227)
        mag[i] = powf(s, power);                        // tf.pow(S, power) (audio.py:64)
    }
}

void launch_gl_mag(const float* lin, float* mag, size_t n, float min_db, float ref_db, float power, hipStream_t s,
                   Timing* tm) {
    if (tm) tm->begin(KK_GL_MAG, s);
    int blocks = (int)std::min<size_t>(4096, (n + 255) / 256);
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(gl_mag_kernel, dim3(blocks), dim3(256), 0, s, lin, mag, n, min_db, ref_db, power);
    if (tm) tm->end(KK_GL_MAG, s);
}

constexpr int GL_THREADS = FFT_T;   // 256

__global__ __launch_bounds__(GL_THREADS) void gl_iter_kernel(const float* __restrict__ frames_in,
                                                             const float* __restrict__ mag,
                                                             float* __restrict__ frames_out,
                                                             const float2* __restrict__ twiddle,
                                                             const float* __restrict__ window, int B, int T, int hop,
                                                             int win, int nbins, int first) {
    __shared__ __attribute__((aligned(16))) cplx W[2 * FFT_N];
    __shared__ __attribute__((aligned(16))) cplx buf[2][FFT_N];
    const int tid = threadIdx.x;
    for (int i = tid; i < 2 * FFT_N; i += GL_THREADS) {
        const float2 w = twiddle[i];
        W[i] = {w.x, w.y};
    }
    const int half_win = win >> 1;                // complex samples carrying data
    const int nover = (win + hop - 1) / hop;      // frames covering one sample (4)
    const long nframes = (long)B * T;

    for (long f = blockIdx.x; f < nframes; f += gridDim.x) {
        const int m = (int)(f % T);
        const long fb = f - m;                    // first frame of this utterance
        __syncthreads();                          // W loaded / previous frame's buffers free
        int cur = 0;
        if (!first) {
            // ---- analysis input: y[hop*m + n] = sum of covering frames, * window, packed even/odd
            for (int c = tid; c < FFT_N; c += GL_THREADS) {
                cplx z = {0.f, 0.f};
                if (c < half_win) {
                    const int n = 2 * c;
                    const int g = hop * m + n;
                    const int mhi = min(T - 1, g / hop);
                    const int mlo = max(0, g / hop - (nover - 1));
                    float sx = 0.f, sy = 0.f;
                    for (int m2 = mlo; m2 <= mhi; ++m2) {
                        const int off = g - hop * m2;
                        if (off + 1 < win + 1 && off < win) {
                            const float2 v = *reinterpret_cast<const float2*>(frames_in + (fb + m2) * win + off);
                            sx += v.x;
                            sy += v.y;
                        }
                    }
                    z = {sx * window[n], sy * window[n + 1]};
                }
                buf[0][c] = z;
            }
            __syncthreads();
            // ---- forward FFT-1024
#pragma unroll
            for (int ps = 0, p = 1; ps < FFT_PASSES; ++ps, p <<= 2) {
                fft1024_radix4<false>(buf[cur], buf[cur ^ 1], tid, p, W);
                cur ^= 1;
                __syncthreads();
            }
        }
        // ---- split -> re-phase against mag -> merge, in place on the bin pairs (k, 1024-k)
        {
            cplx* Z = buf[cur];
            const float* mg = mag + f * nbins;
            for (int k = tid; k < FFT_N / 2; k += GL_THREADS) {
                if (k == 0) {
                    // bins 0 and 1024 are real: X0 = Re Z0 + Im Z0, X1024 = Re Z0 - Im Z0
                    float y0, yn;
                    if (first) { y0 = mg[0]; yn = mg[FFT_N]; }
                    else {
                        const cplx z0 = Z[0];
                        y0 = gl_rephase({z0.x + z0.y, 0.f}, mg[0]).x;
                        yn = gl_rephase({z0.x - z0.y, 0.f}, mg[FFT_N]).x;
                    }
                    Z[0] = {0.5f * (y0 + yn), 0.5f * (y0 - yn)};
                    // bin 512 pairs with itself: X512 = conj(Z512); Z'512 = conj(Y512)
                    cplx ym;
                    if (first) ym = {mg[FFT_N / 2], 0.f};
                    else ym = gl_rephase(cconj(Z[FFT_N / 2]), mg[FFT_N / 2]);
                    Z[FFT_N / 2] = cconj(ym);
                } else {
                    const cplx Wk = W[k];
                    cplx Yk, Yn;
                    if (first) { Yk = {mg[k], 0.f}; Yn = {mg[FFT_N - k], 0.f}; }
                    else {
                        cplx Xk, Xn;
                        rfft_split(Z[k], Z[FFT_N - k], Wk, Xk, Xn);
                        Yk = gl_rephase(Xk, mg[k]);
                        Yn = gl_rephase(Xn, mg[FFT_N - k]);
                    }
                    cplx Zk, Zn;
                    irfft_merge(Yk, Yn, Wk, Zk, Zn);
                    Z[k] = Zk;
                    Z[FFT_N - k] = Zn;
                }
            }
        }
        __syncthreads();
        // ---- inverse FFT-1024
#pragma unroll
        for (int ps = 0, p = 1; ps < FFT_PASSES; ++ps, p <<= 2) {
            fft1024_radix4<true>(buf[cur], buf[cur ^ 1], tid, p, W);
            cur ^= 1;
            __syncthreads();
        }
        // ---- synthesis window, first `win` samples, 1/1024 scaling
        {
            const cplx* z = buf[cur];
            const float sc = 1.f / FFT_N;
            for (int c = tid; c < half_win; c += GL_THREADS) {
                const cplx v = z[c];
                float2 o;
                o.x = v.x * sc * window[2 * c];
                o.y = v.y * sc * window[2 * c + 1];
                *reinterpret_cast<float2*>(frames_out + f * win + 2 * c) = o;
            }
        }
    }
}

void launch_gl_iter(const GLPlan& p, const float* frames_in, const float* mag, float* frames_out, int B, int T,
                    int first, hipStream_t s, Timing* tm) {
    if (tm) tm->begin(KK_GL_ITER, s);
    const long nframes = (long)B * T;
    const int blocks = (int)std::min<long>(nframes, 256 * 5);
    hipLaunchKernelGGL(gl_iter_kernel, dim3(blocks), dim3(GL_THREADS), 0, s, frames_in, mag, frames_out,
                       (const float2*)p.twiddle, p.window, B, T, p.hop, p.win, p.nbins, first);
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

void launch_gl_ola(const GLPlan& p, const float* frames, float* wav, int B, int T, hipStream_t s, Timing* tm) {
    if (tm) tm->begin(KK_GL_OLA, s);
    const long total = (long)B * ((long)(T - 1) * p.hop + p.win);
    const int blocks = (int)std::min<long>(4096, (total + 255) / 256);
    hipLaunchKernelGGL(gl_ola_kernel, dim3(blocks), dim3(256), 0, s, frames, wav, B, T, p.hop, p.win);
    if (tm) tm->end(KK_GL_OLA, s);
}

}  // namespace taco
