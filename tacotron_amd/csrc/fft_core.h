// 2048-point real FFT building blocks shared by the Griffin-Lim kernel (griffinlim.hip) and the
// host emulation test (tests/host/fft_host_test.cpp): plain C++ so the index maths can be
// checked on the CPU where there is no GPU.
//
// A real 2048-point transform is computed as a 1024-point complex FFT of the even/odd-packed
// signal z[m] = x[2m] + i x[2m+1] plus a split step. The complex FFT is done by ONE 64-lane wave
// with 16 points per lane in registers, as a three-pass mixed-radix Stockham transform
// (16 x 16 x 4): lane l is butterfly l of the two radix-16 passes and butterflies l, l+64, l+128,
// l+192 of the radix-4 pass. Every pass reads x[l + 64 i] (contiguous over lanes) and writes
// y[(t - k) R + k + p i'] (k = t mod p, p = product of the radices done so far), exchanged through a
// wave-private LDS buffer that is padded by one element per 16 (conflict-free strides).
#pragma once

#if defined(__HIPCC__)
#define TACO_HD __host__ __device__ __forceinline__
#else
#define TACO_HD inline
#endif

namespace taco {

#if defined(__HIPCC__)
// (re, im) as one two-float vector: a complex add is one packed instruction on the pair, a complex multiply two
// (packed multiply + packed FMA with swizzled operands); left as a struct of two floats the vectoriser pairs
// components of different elements instead and pays for it in register moves
typedef float cplx __attribute__((ext_vector_type(2)));
TACO_HD cplx cadd(cplx a, cplx b) { return a + b; }
TACO_HD cplx csub(cplx a, cplx b) { return a - b; }
TACO_HD cplx cmul(cplx a, cplx b) { return __builtin_elementwise_fma(a.yy, cplx{-b.y, b.x}, a.xx * b); }
TACO_HD cplx cconj(cplx a) { return cplx{a.x, -a.y}; }
// a + (-i) d and a + (+i) d as ONE packed add: the rotation is the instruction's operand selection (second source
// half-swapped) and negation of one half. Left to the compiler, the rotated value is built with a move and a sign flip
// per butterfly (two extra instructions in each of the 40 radix-4 butterflies of a frame). Same arithmetic, bit for bit.
#if defined(__HIP_DEVICE_COMPILE__)
TACO_HD cplx cadd_mi(cplx a, cplx d) {
    cplx r;
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(d));
    return r;
}
TACO_HD cplx cadd_pi(cplx a, cplx d) {
    cplx r;
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]" : "=v"(r) : "v"(a), "v"(d));
    return r;
}
// a * b and a * conj(b) for a twiddle b that comes from memory: the half-swapped, half-negated second factor of the
// FMA is operand selection as well (the compiler builds it with a move and a sign flip per product; for compile-time
// constant twiddles cmul folds them into literals and stays the better form). Bit-identical to cmul / cmul(a, cconj(b)).
// (one asm statement per product, two per butterfly pair below: the compiler pads every asm statement with an s_nop
// against hazards it cannot see into, 170 of them per pair of frames when each instruction was a statement of its own)
TACO_HD cplx cmul_v(cplx a, cplx b) {
    cplx r;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[0,1]\n\t"
        "v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]"
        : "=&v"(r) : "v"(a), "v"(b));
    return r;
}
TACO_HD cplx cmul_vc(cplx a, cplx b) {
    cplx r;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[0,1] neg_hi:[0,1]\n\t"
        "v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,1,0] op_sel_hi:[1,0,1]"
        : "=&v"(r) : "v"(a), "v"(b));
    return r;
}
// 3 products a_i <- a_i * b_i in one asm statement
TACO_HD void cmul_v3(cplx& a0, cplx& a1, cplx& a2, cplx b0, cplx b1, cplx b2) {
    cplx r0, r1, r2;
    asm("v_pk_mul_f32 %0, %3, %6 op_sel:[0,0] op_sel_hi:[0,1]\n\t"
        "v_pk_fma_f32 %0, %3, %6, %0 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]\n\t"
        "v_pk_mul_f32 %1, %4, %7 op_sel:[0,0] op_sel_hi:[0,1]\n\t"
        "v_pk_fma_f32 %1, %4, %7, %1 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]\n\t"
        "v_pk_mul_f32 %2, %5, %8 op_sel:[0,0] op_sel_hi:[0,1]\n\t"
        "v_pk_fma_f32 %2, %5, %8, %2 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]"
        : "=&v"(r0), "=&v"(r1), "=&v"(r2) : "v"(a0), "v"(a1), "v"(a2), "v"(b0), "v"(b1), "v"(b2));
    a0 = r0;
    a1 = r1;
    a2 = r2;
}
// 3 products a_i <- a_i * conj b_i in one asm statement
TACO_HD void cmul_vc3(cplx& a0, cplx& a1, cplx& a2, cplx b0, cplx b1, cplx b2) {
    cplx r0, r1, r2;
    asm("v_pk_mul_f32 %0, %3, %6 op_sel:[0,0] op_sel_hi:[0,1] neg_hi:[0,1]\n\t"
        "v_pk_fma_f32 %0, %3, %6, %0 op_sel:[1,1,0] op_sel_hi:[1,0,1]\n\t"
        "v_pk_mul_f32 %1, %4, %7 op_sel:[0,0] op_sel_hi:[0,1] neg_hi:[0,1]\n\t"
        "v_pk_fma_f32 %1, %4, %7, %1 op_sel:[1,1,0] op_sel_hi:[1,0,1]\n\t"
        "v_pk_mul_f32 %2, %5, %8 op_sel:[0,0] op_sel_hi:[0,1] neg_hi:[0,1]\n\t"
        "v_pk_fma_f32 %2, %5, %8, %2 op_sel:[1,1,0] op_sel_hi:[1,0,1]"
        : "=&v"(r0), "=&v"(r1), "=&v"(r2) : "v"(a0), "v"(a1), "v"(a2), "v"(b0), "v"(b1), "v"(b2));
    a0 = r0;
    a1 = r1;
    a2 = r2;
}
// 4 products a_i <- a_i * b_i in one asm statement
TACO_HD void cmul_v4(cplx& a0, cplx& a1, cplx& a2, cplx& a3, cplx b0, cplx b1, cplx b2, cplx b3) {
    cplx r0, r1, r2, r3;
    asm("v_pk_mul_f32 %0, %4, %8 op_sel:[0,0] op_sel_hi:[0,1]\n\t"
        "v_pk_fma_f32 %0, %4, %8, %0 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]\n\t"
        "v_pk_mul_f32 %1, %5, %9 op_sel:[0,0] op_sel_hi:[0,1]\n\t"
        "v_pk_fma_f32 %1, %5, %9, %1 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]\n\t"
        "v_pk_mul_f32 %2, %6, %10 op_sel:[0,0] op_sel_hi:[0,1]\n\t"
        "v_pk_fma_f32 %2, %6, %10, %2 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]\n\t"
        "v_pk_mul_f32 %3, %7, %11 op_sel:[0,0] op_sel_hi:[0,1]\n\t"
        "v_pk_fma_f32 %3, %7, %11, %3 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]"
        : "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3) : "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(b0), "v"(b1), "v"(b2), "v"(b3));
    a0 = r0;
    a1 = r1;
    a2 = r2;
    a3 = r3;
}
// 4 products a_i <- a_i * conj b_i in one asm statement
TACO_HD void cmul_vc4(cplx& a0, cplx& a1, cplx& a2, cplx& a3, cplx b0, cplx b1, cplx b2, cplx b3) {
    cplx r0, r1, r2, r3;
    asm("v_pk_mul_f32 %0, %4, %8 op_sel:[0,0] op_sel_hi:[0,1] neg_hi:[0,1]\n\t"
        "v_pk_fma_f32 %0, %4, %8, %0 op_sel:[1,1,0] op_sel_hi:[1,0,1]\n\t"
        "v_pk_mul_f32 %1, %5, %9 op_sel:[0,0] op_sel_hi:[0,1] neg_hi:[0,1]\n\t"
        "v_pk_fma_f32 %1, %5, %9, %1 op_sel:[1,1,0] op_sel_hi:[1,0,1]\n\t"
        "v_pk_mul_f32 %2, %6, %10 op_sel:[0,0] op_sel_hi:[0,1] neg_hi:[0,1]\n\t"
        "v_pk_fma_f32 %2, %6, %10, %2 op_sel:[1,1,0] op_sel_hi:[1,0,1]\n\t"
        "v_pk_mul_f32 %3, %7, %11 op_sel:[0,0] op_sel_hi:[0,1] neg_hi:[0,1]\n\t"
        "v_pk_fma_f32 %3, %7, %11, %3 op_sel:[1,1,0] op_sel_hi:[1,0,1]"
        : "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3) : "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(b0), "v"(b1), "v"(b2), "v"(b3));
    a0 = r0;
    a1 = r1;
    a2 = r2;
    a3 = r3;
}
// four (a - i d, a + i d) pairs in one statement
TACO_HD void cadd_mi_pi4(const cplx (&a)[4], const cplx (&d)[4], cplx (&rm)[4], cplx (&rp)[4]) {
    asm("v_pk_add_f32 %0, %8, %12 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]\n\t"
        "v_pk_add_f32 %4, %8, %12 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]\n\t"
        "v_pk_add_f32 %1, %9, %13 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]\n\t"
        "v_pk_add_f32 %5, %9, %13 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]\n\t"
        "v_pk_add_f32 %2, %10, %14 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]\n\t"
        "v_pk_add_f32 %6, %10, %14 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]\n\t"
        "v_pk_add_f32 %3, %11, %15 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]\n\t"
        "v_pk_add_f32 %7, %11, %15 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]"
        : "=&v"(rm[0]), "=&v"(rm[1]), "=&v"(rm[2]), "=&v"(rm[3]), "=&v"(rp[0]), "=&v"(rp[1]), "=&v"(rp[2]), "=&v"(rp[3])
        : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(d[0]), "v"(d[1]), "v"(d[2]), "v"(d[3]));
}
// (a - i d, a + i d) together
TACO_HD void cadd_mi_pi(cplx a, cplx d, cplx& rm, cplx& rp) {
    asm("v_pk_add_f32 %0, %2, %3 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]\n\t"
        "v_pk_add_f32 %1, %2, %3 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]"
        : "=&v"(rm), "=&v"(rp) : "v"(a), "v"(d));
}
#else
TACO_HD cplx cadd_mi(cplx a, cplx d) { return cplx{a.x + d.y, a.y - d.x}; }
TACO_HD cplx cadd_pi(cplx a, cplx d) { return cplx{a.x - d.y, a.y + d.x}; }
TACO_HD cplx cmul_v(cplx a, cplx b) { return cmul(a, b); }
TACO_HD cplx cmul_vc(cplx a, cplx b) { return cmul(a, cconj(b)); }
TACO_HD void cadd_mi_pi(cplx a, cplx d, cplx& rm, cplx& rp) { rm = cadd_mi(a, d); rp = cadd_pi(a, d); }
TACO_HD void cadd_mi_pi4(const cplx (&a)[4], const cplx (&d)[4], cplx (&rm)[4], cplx (&rp)[4]) {
    for (int i = 0; i < 4; ++i) { rm[i] = cadd_mi(a[i], d[i]); rp[i] = cadd_pi(a[i], d[i]); }
}
TACO_HD void cmul_v3(cplx& a0, cplx& a1, cplx& a2, cplx b0, cplx b1, cplx b2) { a0 = cmul_v(a0, b0); a1 = cmul_v(a1, b1); a2 = cmul_v(a2, b2); }
TACO_HD void cmul_vc3(cplx& a0, cplx& a1, cplx& a2, cplx b0, cplx b1, cplx b2) { a0 = cmul_vc(a0, b0); a1 = cmul_vc(a1, b1); a2 = cmul_vc(a2, b2); }
TACO_HD void cmul_v4(cplx& a0, cplx& a1, cplx& a2, cplx& a3, cplx b0, cplx b1, cplx b2, cplx b3) { a0 = cmul_v(a0, b0); a1 = cmul_v(a1, b1); a2 = cmul_v(a2, b2); a3 = cmul_v(a3, b3); }
TACO_HD void cmul_vc4(cplx& a0, cplx& a1, cplx& a2, cplx& a3, cplx b0, cplx b1, cplx b2, cplx b3) { a0 = cmul_vc(a0, b0); a1 = cmul_vc(a1, b1); a2 = cmul_vc(a2, b2); a3 = cmul_vc(a3, b3); }
#endif
#else
struct cplx {
    float x, y;
};
TACO_HD cplx cadd(cplx a, cplx b) { return {a.x + b.x, a.y + b.y}; }
TACO_HD cplx csub(cplx a, cplx b) { return {a.x - b.x, a.y - b.y}; }
TACO_HD cplx cmul(cplx a, cplx b) { return {a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x}; }
TACO_HD cplx cconj(cplx a) { return {a.x, -a.y}; }
TACO_HD cplx cadd_mi(cplx a, cplx d) { return {a.x + d.y, a.y - d.x}; }
TACO_HD cplx cadd_pi(cplx a, cplx d) { return {a.x - d.y, a.y + d.x}; }
TACO_HD cplx cmul_v(cplx a, cplx b) { return cmul(a, b); }
TACO_HD cplx cmul_vc(cplx a, cplx b) { return cmul(a, cconj(b)); }
TACO_HD void cadd_mi_pi(cplx a, cplx d, cplx& rm, cplx& rp) { rm = cadd_mi(a, d); rp = cadd_pi(a, d); }
TACO_HD void cadd_mi_pi4(const cplx (&a)[4], const cplx (&d)[4], cplx (&rm)[4], cplx (&rp)[4]) {
    for (int i = 0; i < 4; ++i) { rm[i] = cadd_mi(a[i], d[i]); rp[i] = cadd_pi(a[i], d[i]); }
}
TACO_HD void cmul_v3(cplx& a0, cplx& a1, cplx& a2, cplx b0, cplx b1, cplx b2) { a0 = cmul_v(a0, b0); a1 = cmul_v(a1, b1); a2 = cmul_v(a2, b2); }
TACO_HD void cmul_vc3(cplx& a0, cplx& a1, cplx& a2, cplx b0, cplx b1, cplx b2) { a0 = cmul_vc(a0, b0); a1 = cmul_vc(a1, b1); a2 = cmul_vc(a2, b2); }
TACO_HD void cmul_v4(cplx& a0, cplx& a1, cplx& a2, cplx& a3, cplx b0, cplx b1, cplx b2, cplx b3) { a0 = cmul_v(a0, b0); a1 = cmul_v(a1, b1); a2 = cmul_v(a2, b2); a3 = cmul_v(a3, b3); }
TACO_HD void cmul_vc4(cplx& a0, cplx& a1, cplx& a2, cplx& a3, cplx b0, cplx b1, cplx b2, cplx b3) { a0 = cmul_vc(a0, b0); a1 = cmul_vc(a1, b1); a2 = cmul_vc(a2, b2); a3 = cmul_vc(a3, b3); }
#endif

constexpr int FFT_N = 1024;                 // complex length
constexpr int FFT_LANES = 64;               // one wave
constexpr int FFT_PER_LANE = 16;
constexpr int FFT_LDS = FFT_N + FFT_N / 16; // padded exchange buffer, complex elements
TACO_HD int fft_pad(int j) { return j + (j >> 4); }

// 4-point DFT in place, natural order. INV: conjugate kernel (no scaling).
template <bool INV>
TACO_HD void dft4(cplx& a0, cplx& a1, cplx& a2, cplx& a3) {
    const cplx v0 = cadd(a0, a2);
    const cplx v1 = csub(a0, a2);
    const cplx v2 = cadd(a1, a3);
    const cplx d = csub(a1, a3);
    a0 = cadd(v0, v2);
    a2 = csub(v0, v2);
    if (INV) cadd_mi_pi(v1, d, a3, a1);           // a1 = v1 + d * (+i), a3 = v1 + d * (-i)
    else cadd_mi_pi(v1, d, a1, a3);               // a1 = v1 + d * (-i), a3 = v1 + d * (+i)
}

// w16^n = exp(-2 pi i n / 16) for the products n = a2 * c1 that occur (0,1,2,3,4,6,9)
template <bool INV>
TACO_HD cplx w16(int n) {
    constexpr float C1 = 0.92387953251128674f, S1 = 0.38268343236508977f, C2 = 0.70710678118654752f;
    cplx w;
    switch (n) {
        case 1: w = {C1, -S1}; break;
        case 2: w = {C2, -C2}; break;
        case 3: w = {S1, -C1}; break;
        case 4: w = {0.f, -1.f}; break;
        case 6: w = {-C2, -C2}; break;
        case 9: w = {-C1, S1}; break;
        default: w = {1.f, 0.f}; break;
    }
    return INV ? cconj(w) : w;
}

// Four independent 4-point DFTs in place: DFT j works on v[o + j sj + i si], i = 0..3; the four pairs of +-i rotations
// are one statement (cadd_mi_pi4).
template <bool INV, int O, int SJ, int SI, int N>
TACO_HD void dft4x4(cplx (&v)[N]) {
    cplx v1[4], d[4], rm[4], rp[4];
#if defined(__HIPCC__)
#pragma unroll
#endif
    for (int j = 0; j < 4; ++j) {
        cplx &a0 = v[O + j * SJ], &a1 = v[O + j * SJ + SI], &a2 = v[O + j * SJ + 2 * SI], &a3 = v[O + j * SJ + 3 * SI];
        const cplx v0 = cadd(a0, a2);
        v1[j] = csub(a0, a2);
        const cplx v2 = cadd(a1, a3);
        d[j] = csub(a1, a3);
        a0 = cadd(v0, v2);
        a2 = csub(v0, v2);
    }
    cadd_mi_pi4(v1, d, rm, rp);
#if defined(__HIPCC__)
#pragma unroll
#endif
    for (int j = 0; j < 4; ++j) {
        v[O + j * SJ + SI] = INV ? rp[j] : rm[j];          // v1 + d * (+i) : v1 + d * (-i)
        v[O + j * SJ + 3 * SI] = INV ? rm[j] : rp[j];
    }
}

// 16-point DFT in place, natural order in and out (index a = 4 a1 + a2, c = c1 + 4 c2).
template <bool INV>
TACO_HD void dft16(cplx (&v)[16]) {
    dft4x4<INV, 0, 1, 4>(v);                 // DFT a2 over (v[a2], v[4 + a2], v[8 + a2], v[12 + a2]) -> v[4 c1 + a2]
#if defined(__HIPCC__)
#pragma unroll
#endif
    for (int c1 = 1; c1 < 4; ++c1)
#if defined(__HIPCC__)
#pragma unroll
#endif
        for (int a2 = 1; a2 < 4; ++a2) v[4 * c1 + a2] = cmul(v[4 * c1 + a2], w16<INV>(a2 * c1));
    dft4x4<INV, 0, 4, 1>(v);                 // DFT c1 over v[4 c1 .. 4 c1 + 3] -> v[4 c1 + c2]
    cplx o[16];
#if defined(__HIPCC__)
#pragma unroll
#endif
    for (int c1 = 0; c1 < 4; ++c1)
#if defined(__HIPCC__)
#pragma unroll
#endif
        for (int c2 = 0; c2 < 4; ++c2) o[c1 + 4 * c2] = v[4 * c1 + c2];
#if defined(__HIPCC__)
#pragma unroll
#endif
    for (int c = 0; c < 16; ++c) v[c] = o[c];
}

// Twiddle tables of passes 2 and 3, built once per workgroup (in LDS on the device) from
// W[j] = exp(-2 pi i j / 2048), j < 2048, laid out so that the 64 lanes read consecutive entries:
//   T2[(i-1)*16 + k]  = w256^(k i),   k < 16, i = 1..15   (pass 2: p = 16, radix 16, k = lane & 15)
//   T3[(i-1)*256 + t] = w1024^(t i),  t < 256, i = 1..3   (pass 3: p = 256, radix 4, t = lane + 64 q)
constexpr int FFT_T2 = 15 * 16, FFT_T3 = 3 * 256, FFT_TW = FFT_T2 + FFT_T3;
TACO_HD cplx fft_tw_entry(int e, const cplx* W) {
    if (e < FFT_T2) { const int i = e / 16 + 1, k = e % 16; return W[8 * k * i]; }
    e -= FFT_T2;
    const int i = e / 256 + 1, t = e % 256;
    return W[2 * t * i];
}

// ---- the passes, split into the part before and after the exchange so that a host emulation can run
// "all lanes load, then all lanes store" like the wave does.
// every pass starts from v[i] = x[lane + 64 i]
TACO_HD void fft_load(cplx (&v)[16], const cplx* lds, int lane) {
#if defined(__HIPCC__)
#pragma unroll
#endif
    for (int i = 0; i < 16; ++i) v[i] = lds[fft_pad(lane + 64 * i)];
}
TACO_HD void fft_store_natural(const cplx (&v)[16], cplx* lds, int lane) {
#if defined(__HIPCC__)
#pragma unroll
#endif
    for (int i = 0; i < 16; ++i) lds[fft_pad(lane + 64 * i)] = v[i];
}
// pass 1 (p = 1): y[16 t + c]
template <bool INV>
TACO_HD void fft_pass1(cplx (&v)[16], cplx* lds, int lane) {
    dft16<INV>(v);
#if defined(__HIPCC__)
#pragma unroll
#endif
    for (int c = 0; c < 16; ++c) lds[fft_pad(16 * lane + c)] = v[c];
}
// pass 2 (p = 16): twiddle, DFT16, y[(t - k) 16 + k + 16 c]; v must hold x[lane + 64 i]
template <bool INV>
TACO_HD void fft_pass2(cplx (&v)[16], cplx* lds, int lane, const cplx* T2) {
#if defined(__HIPCC__)
#pragma unroll
#endif
    for (int i = 1; i < 16; ++i) {
        const cplx w = T2[(i - 1) * 16 + (lane & 15)];
        v[i] = INV ? cmul_vc(v[i], w) : cmul_v(v[i], w);
    }
    dft16<INV>(v);
    const int base = (lane >> 4) * 256 + (lane & 15);
#if defined(__HIPCC__)
#pragma unroll
#endif
    for (int c = 0; c < 16; ++c) lds[fft_pad(base + 16 * c)] = v[c];
}
// pass 3 (p = 256, radix 4): in registers; on return v[m] = X[lane + 64 m] (natural order)
template <bool INV>
TACO_HD void fft_pass3(cplx (&v)[16], int lane, const cplx* T3) {
#if defined(__HIPCC__)
#pragma unroll
#endif
    for (int q = 0; q < 4; ++q) {
#if defined(__HIPCC__)
#pragma unroll
#endif
        for (int i = 1; i < 4; ++i) {
            const cplx w = T3[(i - 1) * 256 + lane + 64 * q];
            v[q + 4 * i] = INV ? cmul_vc(v[q + 4 * i], w) : cmul_v(v[q + 4 * i], w);
        }
        dft4<INV>(v[q], v[q + 4], v[q + 8], v[q + 12]);
    }
}

// Split step of the real transform for the bin pair (k, 1024-k), 1 <= k <= 511:
// from Z = FFT1024(z) to the real-signal spectrum X[k], X[1024-k].
#if defined(__HIPCC__)
TACO_HD cplx chalf_sum(cplx a, cplx b) { return 0.5f * (a + b); }
TACO_HD cplx chalf_diff(cplx a, cplx b) { return 0.5f * (a - b); }
#else
TACO_HD cplx chalf_sum(cplx a, cplx b) { return {0.5f * (a.x + b.x), 0.5f * (a.y + b.y)}; }
TACO_HD cplx chalf_diff(cplx a, cplx b) { return {0.5f * (a.x - b.x), 0.5f * (a.y - b.y)}; }
#endif
TACO_HD void rfft_split(cplx Zk, cplx Zn, cplx Wk, cplx& Xk, cplx& Xn) {
    const cplx cn = cconj(Zn);
    const cplx E = chalf_sum(Zk, cn);
    const cplx D = chalf_diff(Zk, cn);
    const cplx O = {D.y, -D.x};              // D / i
    const cplx P = cmul_v(Wk, O);
    Xk = cadd(E, P);
    Xn = cconj(csub(E, P));
}

// Inverse of rfft_split: from a Hermitian spectrum pair to Z' with z' = IFFT1024(Z')
// (x[2m] = Re z'[m] / 1024, x[2m+1] = Im z'[m] / 1024).
TACO_HD void irfft_merge(cplx Yk, cplx Yn, cplx Wk, cplx& Zk, cplx& Zn) {
    const cplx cn = cconj(Yn);
    const cplx E = chalf_sum(Yk, cn);
    const cplx D = chalf_diff(Yk, cn);
    const cplx O = cmul_vc(D, Wk);
    const cplx iO = {-O.y, O.x};
    Zk = cadd(E, iO);
    Zn = cconj(csub(E, iO));
}

// A = X / max(1e-8, |X|) scaled by the target magnitude (tools/audio.py:194-195).
TACO_HD cplx gl_rephase(cplx X, float mag) {
#if defined(__HIP_DEVICE_COMPILE__)
    // 1 / max(1e-8, |X|) = rsq(max(1e-16, |X|^2)): one hardware reciprocal square root (1 ulp) instead of the IEEE
    // sqrt and divide sequences (~20 instructions each, 17 bins per lane and frame)
    const float a2 = __builtin_fmaf(X.x, X.x, X.y * X.y);
    const float s = mag * __builtin_amdgcn_rsqf(a2 > 1e-16f ? a2 : 1e-16f);
    return X * s;
#else
    const float a = __builtin_sqrtf(X.x * X.x + X.y * X.y);
    const float s = mag / (a > 1e-8f ? a : 1e-8f);
#endif
    return {X.x * s, X.y * s};
}

}  // namespace taco
