// Host emulation of the Griffin-Lim kernel's FFT path (tacotron_amd/csrc/fft_core.h): runs the
// same three passes "lane by lane" (all lanes up to an exchange, then all lanes past it, as the wave
// does) and checks them against a direct O(N^2) DFT in double.
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
    std::vector<cplx> lds(FFT_LDS);
    std::vector<cplx> tw(FFT_TW);
    for (int e = 0; e < FFT_TW; ++e) tw[e] = fft_tw_entry(e, W.data());
    const cplx* T2 = tw.data();
    const cplx* T3 = tw.data() + FFT_T2;
    static cplx v[FFT_LANES][16];
    auto fft = [&](bool inv) {   // in: v[l][i] = x[l + 64 i]; out: v[l][m] = X[l + 64 m]
        for (int l = 0; l < FFT_LANES; ++l) { if (inv) fft_pass1<true>(v[l], lds.data(), l); else fft_pass1<false>(v[l], lds.data(), l); }
        for (int l = 0; l < FFT_LANES; ++l) fft_load(v[l], lds.data(), l);
        for (int l = 0; l < FFT_LANES; ++l) { if (inv) fft_pass2<true>(v[l], lds.data(), l, T2); else fft_pass2<false>(v[l], lds.data(), l, T2); }
        for (int l = 0; l < FFT_LANES; ++l) fft_load(v[l], lds.data(), l);
        for (int l = 0; l < FFT_LANES; ++l) { if (inv) fft_pass3<true>(v[l], l, T3); else fft_pass3<false>(v[l], l, T3); }
    };
    for (int l = 0; l < FFT_LANES; ++l)
        for (int i = 0; i < 16; ++i) v[l][i] = {(float)x[2 * (l + 64 * i)], (float)x[2 * (l + 64 * i) + 1]};
    fft(false);
    std::vector<cplx> buf[2] = {std::vector<cplx>(FFT_N), std::vector<cplx>(FFT_N)};
    int cur = 0;
    for (int l = 0; l < FFT_LANES; ++l)
        for (int m = 0; m < 16; ++m) buf[cur][l + 64 * m] = v[l][m];
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
    for (int l = 0; l < FFT_LANES; ++l)
        for (int i = 0; i < 16; ++i) v[l][i] = buf[0][l + 64 * i];
    fft(true);
    for (int l = 0; l < FFT_LANES; ++l)
        for (int m = 0; m < 16; ++m) buf[cur][l + 64 * m] = v[l][m];
    double einv = 0, yscale = 0;
    for (int c = 0; c < FFT_N; ++c) {
        einv = fmax(einv, fabs(buf[cur][c].x / FFT_N - yr[2 * c]));
        einv = fmax(einv, fabs(buf[cur][c].y / FFT_N - yr[2 * c + 1]));
        yscale = fmax(yscale, fabs(yr[2 * c]));
    }
    printf("%.3e %.3e %.3e %.3e\n", efwd, scale, einv, yscale);
    return (efwd < 1e-4 * scale && einv < 1e-5 * fmax(1.0, yscale)) ? 0 : 1;
}
