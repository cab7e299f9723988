// Host emulation of the Griffin-Lim kernel's FFT path (tacotron_amd/csrc/fft_core.h): runs the
// same butterflies "thread by thread" and checks them against a direct O(N^2) DFT in double.
// Prints "max_err_fwd max_err_inv max_err_roundtrip"; exit code 0 if all are small.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../../tacotron_amd/csrc/fft_core.h"

using namespace taco;

int main() {
    const int NF = 2048, WIN = 1200, NB = 1025;
    std::vector<cplx> W(NF);
    for (int j = 0; j < NF; ++j) W[j] = {(float)cos(2.0 * M_PI * j / NF), (float)-sin(2.0 * M_PI * j / NF)};
    std::vector<double> x(NF, 0.0);
    srand(1);
    for (int n = 0; n < WIN; ++n) x[n] = rand() / (double)RAND_MAX - 0.5;
    // reference spectrum
    std::vector<double> Xr(NB), Xi(NB);
    for (int k = 0; k < NB; ++k) {
        double sr = 0, si = 0;
        for (int n = 0; n < WIN; ++n) { sr += x[n] * cos(2.0 * M_PI * k * n / NF); si -= x[n] * sin(2.0 * M_PI * k * n / NF); }
        Xr[k] = sr; Xi[k] = si;
    }
    // emulated forward
    std::vector<cplx> buf[2] = {std::vector<cplx>(FFT_N), std::vector<cplx>(FFT_N)};
    for (int c = 0; c < FFT_N; ++c) buf[0][c] = {(float)x[2 * c], (float)x[2 * c + 1]};
    int cur = 0;
    for (int ps = 0, p = 1; ps < FFT_PASSES; ++ps, p <<= 2) {
        for (int t = 0; t < FFT_T; ++t) fft1024_radix4<false>(buf[cur].data(), buf[cur ^ 1].data(), t, p, W.data());
        cur ^= 1;
    }
    std::vector<cplx> X(NB);
    {
        cplx* Z = buf[cur].data();
        X[0] = {Z[0].x + Z[0].y, 0.f};
        X[FFT_N] = {Z[0].x - Z[0].y, 0.f};
        X[FFT_N / 2] = cconj(Z[FFT_N / 2]);
        for (int k = 1; k < FFT_N / 2; ++k) rfft_split(Z[k], Z[FFT_N - k], W[k], X[k], X[FFT_N - k]);
    }
    double efwd = 0, scale = 0;
    for (int k = 0; k < NB; ++k) {
        efwd = fmax(efwd, fmax(fabs(X[k].x - Xr[k]), fabs(X[k].y - Xi[k])));
        scale = fmax(scale, hypot(Xr[k], Xi[k]));
    }
    // emulated inverse of an arbitrary Hermitian spectrum Y (bins 0/1024 real)
    std::vector<cplx> Y(NB);
    for (int k = 0; k < NB; ++k) Y[k] = {(float)(rand() / (double)RAND_MAX - 0.5), (float)(rand() / (double)RAND_MAX - 0.5)};
    Y[0].y = 0; Y[FFT_N].y = 0;
    std::vector<double> yr(NF);
    for (int n = 0; n < NF; ++n) {
        double s = Y[0].x + Y[FFT_N].x * ((n & 1) ? -1.0 : 1.0);
        for (int k = 1; k < FFT_N; ++k) s += 2.0 * (Y[k].x * cos(2.0 * M_PI * k * n / NF) - Y[k].y * sin(2.0 * M_PI * k * n / NF));
        yr[n] = s / NF;
    }
    {
        cplx* Z = buf[0].data();
        cur = 0;
        Z[0] = {0.5f * (Y[0].x + Y[FFT_N].x), 0.5f * (Y[0].x - Y[FFT_N].x)};
        Z[FFT_N / 2] = cconj(Y[FFT_N / 2]);
        for (int k = 1; k < FFT_N / 2; ++k) irfft_merge(Y[k], Y[FFT_N - k], W[k], Z[k], Z[FFT_N - k]);
    }
    for (int ps = 0, p = 1; ps < FFT_PASSES; ++ps, p <<= 2) {
        for (int t = 0; t < FFT_T; ++t) fft1024_radix4<true>(buf[cur].data(), buf[cur ^ 1].data(), t, p, W.data());
        cur ^= 1;
    }
    double einv = 0, yscale = 0;
    for (int c = 0; c < FFT_N; ++c) {
        einv = fmax(einv, fabs(buf[cur][c].x / FFT_N - yr[2 * c]));
        einv = fmax(einv, fabs(buf[cur][c].y / FFT_N - yr[2 * c + 1]));
        yscale = fmax(yscale, fabs(yr[2 * c]));
    }
    printf("%.3e %.3e %.3e %.3e\n", efwd, scale, einv, yscale);
    return (efwd < 1e-4 * scale && einv < 1e-5 * fmax(1.0, yscale)) ? 0 : 1;
}
