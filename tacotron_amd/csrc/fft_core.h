// 2048-point real FFT building blocks shared by the Griffin-Lim kernel (griffinlim.hip) and the
// host emulation test (tests/host/fft_host_test.cpp): plain C++ so the index maths can be
// checked on the CPU where there is no GPU.
//
// A real 2048-point transform is computed as a 1024-point complex Stockham radix-4 FFT of the
// even/odd-packed signal z[m] = x[2m] + i x[2m+1] plus a split step; 256 "threads" each own
// one radix-4 butterfly per pass (5 passes, ping-pong buffers).
#pragma once

#if defined(__HIPCC__)
#define TACO_HD __host__ __device__ __forceinline__
#else
#define TACO_HD inline
#endif

namespace taco {

struct cplx {
    float x, y;
};
TACO_HD cplx cadd(cplx a, cplx b) { return {a.x + b.x, a.y + b.y}; }
TACO_HD cplx csub(cplx a, cplx b) { return {a.x - b.x, a.y - b.y}; }
TACO_HD cplx cmul(cplx a, cplx b) { return {a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x}; }
TACO_HD cplx cconj(cplx a) { return {a.x, -a.y}; }

constexpr int FFT_N = 1024;      // complex length
constexpr int FFT_T = 256;       // butterflies ("threads") per pass
constexpr int FFT_PASSES = 5;    // 4^5 = 1024

// One radix-4 Stockham butterfly. t in [0,256); p = 4^pass = size of the sub-transforms done so
// far. W[j] = exp(-2*pi*i*j/2048), j < 2048. INV uses conjugate twiddles (no 1/N scaling).
template <bool INV>
TACO_HD void fft1024_radix4(const cplx* x, cplx* y, int t, int p, const cplx* W) {
    const int k = t & (p - 1);
    const int step = 512 / p;    // twiddle exp(-+2*pi*i*k*m/(4p)) = W[k*m*(2048/(4p))]
    cplx u0 = x[t];
    cplx u1 = x[t + FFT_T];
    cplx u2 = x[t + 2 * FFT_T];
    cplx u3 = x[t + 3 * FFT_T];
    if (p > 1) {
        cplx w1 = W[k * step], w2 = W[2 * k * step], w3 = W[3 * k * step];
        if (INV) { w1 = cconj(w1); w2 = cconj(w2); w3 = cconj(w3); }
        u1 = cmul(u1, w1);
        u2 = cmul(u2, w2);
        u3 = cmul(u3, w3);
    }
    const cplx v0 = cadd(u0, u2);
    const cplx v1 = csub(u0, u2);
    const cplx v2 = cadd(u1, u3);
    const cplx d = csub(u1, u3);
    const cplx v3 = INV ? cplx{-d.y, d.x} : cplx{d.y, -d.x};   // d * (+i) : d * (-i)
    const int j = ((t - k) << 2) + k;
    y[j] = cadd(v0, v2);
    y[j + p] = cadd(v1, v3);
    y[j + 2 * p] = csub(v0, v2);
    y[j + 3 * p] = csub(v1, v3);
}

// Split step of the real transform for the bin pair (k, 1024-k), 1 <= k <= 511:
// from Z = FFT1024(z) to the real-signal spectrum X[k], X[1024-k].
TACO_HD void rfft_split(cplx Zk, cplx Zn, cplx Wk, cplx& Xk, cplx& Xn) {
    const cplx cn = cconj(Zn);
    const cplx E = {0.5f * (Zk.x + cn.x), 0.5f * (Zk.y + cn.y)};
    const cplx D = {0.5f * (Zk.x - cn.x), 0.5f * (Zk.y - cn.y)};
    const cplx O = {D.y, -D.x};              // D / i
    const cplx P = cmul(Wk, O);
    Xk = cadd(E, P);
    Xn = cconj(csub(E, P));
}

// Inverse of rfft_split: from a Hermitian spectrum pair to Z' with z' = IFFT1024(Z')
// (x[2m] = Re z'[m] / 1024, x[2m+1] = Im z'[m] / 1024).
TACO_HD void irfft_merge(cplx Yk, cplx Yn, cplx Wk, cplx& Zk, cplx& Zn) {
    const cplx cn = cconj(Yn);
    const cplx E = {0.5f * (Yk.x + cn.x), 0.5f * (Yk.y + cn.y)};
    const cplx D = {0.5f * (Yk.x - cn.x), 0.5f * (Yk.y - cn.y)};
    const cplx O = cmul(D, cconj(Wk));
    const cplx iO = {-O.y, O.x};
    Zk = cadd(E, iO);
    // Z'[N-k] = conj(E) + i conj(O) = conj(E - iO)... i*conj(O) = conj(-i*O) => conj(E) - conj(iO)
    Zn = cconj(csub(E, iO));
}

// A = X / max(1e-8, |X|) scaled by the target magnitude (tools/audio.py:194-195).
TACO_HD cplx gl_rephase(cplx X, float mag) {
#if defined(__HIP_DEVICE_COMPILE__)
    const float a = sqrtf(X.x * X.x + X.y * X.y);
#else
    const float a = __builtin_sqrtf(X.x * X.x + X.y * X.y);
#endif
    const float s = mag / (a > 1e-8f ? a : 1e-8f);
    return {X.x * s, X.y * s};
}

}  // namespace taco
