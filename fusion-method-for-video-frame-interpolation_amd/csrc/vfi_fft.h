// Workgroup-cooperative complex FFT in LDS for gfx950 -- the transform engine of the steerable pyramid
// (the arithmetic the reference gets from torch.fft inside the third-party `steerable.SCFpyr_PyTorch`, call sites
// src/train/pyramid.py:37,44).  No FFT library is involved.
//
//   * `lines` independent transforms of length n live side by side in one LDS buffer (line l at buf + l*pitch);
//     256 threads work on all of them together.  lines * m <= kMaxElems (m = the length actually transformed).
//   * smooth lengths (factors 2, 3, 5) run a mixed-radix Stockham autosort FFT with radices {16, 8, 4, 2, 3, 5}:
//     natural order in, natural order out.  One stage of radix R (p = product of the previous radices, T = m / R):
//         butterfly i in [0, T):  k = i % p;  u[r] = x[i + r*T] * W_m^(r * k * m/(p*R));  v = DFT_R(u);
//                                 y[(i/p)*p*R + k + r*p] = v[r]
//     done IN PLACE: every thread gathers its butterflies into registers, the workgroup synchronises, the results are
//     scattered back -- one LDS buffer, <= 35 complex values per thread.
//   * any other length (the sqrt(2)-scaled pyramid levels: 1358 = 2*7*97, 764 = 4*191, 679, 382, 191, 43 ...) runs
//     Bluestein's chirp-z form on a power-of-two length M >= 2n-1 through the same stages:
//         X[k] = w[k] * sum_j (x[j] w[j]) * conj(w[k-j]),  w[j] = exp(-i pi j^2 / n)
//     = w[k] * IFFT_M( FFT_M(x w, zero padded) * B ),  B = FFT_M(conj(w) wrapped) / M  (precomputed in double);
//     the inverse transform uses the conjugate tables (w is even in j, so B is too).
//   * twiddles come from per-length tables built in double precision (exp(-2 pi i k / m), k < m); a butterfly loads
//     W^k, W^2k, W^4k, W^8k and forms the other powers with at most two further products.
// Transforms are un-normalised in both directions (torch.fft's 1/n of the inverse is applied by the callers).
//
// Everything below is __host__ __device__ with the thread index passed in, so tests/fft_host_check.cpp can run the
// very same index arithmetic on the CPU by looping over "threads" between the synchronisation points.
#pragma once
#include <hip/hip_runtime.h>

namespace vfi {
namespace fft {

constexpr int kThreads = 256;
constexpr int kMaxElems = 8704;      // complex values of all lines of one workgroup (8 columns of a 1080-row level)
constexpr int kMaxElemsB = 8192;     // ... when the plan is a Bluestein plan (its spectral product needs more registers)
constexpr int kLdsElems = 9600;      // LDS elements incl. padding and the twiddle table (75 KiB: two workgroups per CU)
constexpr int kLdsElemsMax = 11264;  // ... for a single line that does not fit the budget above (Bluestein on 8192 points: 4K frames)
constexpr int kMaxStages = 8;

struct Plan1D {                      // plain data, passed to kernels by value
    int n;                           // transform length
    int m;                           // length run through the stages: n (smooth) or the Bluestein length M
    int nstages, bluestein;
    int radix[kMaxStages];
    int tw_len;                      // leading entries of `tw` a workgroup keeps in LDS
    const float2 *tw;                // m entries  exp(-2 pi i k / m)
    const float2 *chirp;             // n entries  exp(-i pi j^2 / n)            (Bluestein only)
    const float2 *bfilt;             // m entries  FFT_M(conj(chirp) wrapped) / M (Bluestein only)
};

#define VFI_HD __host__ __device__ __forceinline__

// ---- complex arithmetic on PACKED f32 instructions ------------------------------------------------------------------------
// The transforms are vector-issue bound, and v_pk_add_f32 / v_pk_mul_f32 / v_pk_fma_f32 work on a
// (re, im) register pair at the price of one instruction: a complex add is one instruction, a complex product two, an
// add of a value turned by -+i one (the turn is an operand modifier).  hipcc packs plain `a + b` on two-element vectors by
// itself but does not fold a swap-and-negate into op_sel / neg, so those forms are written out (tools/probes/pk_complex.hip
// checks them against scalar arithmetic on the GPU); the host build (tests/native/*) takes the scalar expressions.
typedef float cpk __attribute__((ext_vector_type(2)));
VFI_HD cpk pk(float2 a) { return cpk{a.x, a.y}; }
VFI_HD float2 unpk(cpk a) { return make_float2(a.x, a.y); }
#if defined(__HIP_DEVICE_COMPILE__) && !defined(VFI_FFT_SCALAR_COMPLEX)
#define VFI_FFT_PACKED 1
VFI_HD float2 cadd(float2 a, float2 b) { return unpk(pk(a) + pk(b)); }
VFI_HD float2 csub(float2 a, float2 b) { return unpk(pk(a) - pk(b)); }
VFI_HD float2 cscale(float2 a, float k) { return unpk(pk(a) * k); }
VFI_HD float2 cfma(float2 a, float k, float2 b) { return unpk(pk(a) * k + pk(b)); }      // a * k + b, k real
#else
VFI_HD float2 cadd(float2 a, float2 b) { return make_float2(a.x + b.x, a.y + b.y); }
VFI_HD float2 csub(float2 a, float2 b) { return make_float2(a.x - b.x, a.y - b.y); }
VFI_HD float2 cscale(float2 a, float k) { return make_float2(a.x * k, a.y * k); }
VFI_HD float2 cfma(float2 a, float k, float2 b) { return make_float2(a.x * k + b.x, a.y * k + b.y); }
#endif
VFI_HD float2 cconj(float2 a) { return make_float2(a.x, -a.y); }
// a * b
VFI_HD float2 cmul(float2 a, float2 b) {
#if defined(VFI_FFT_PACKED)
    const cpk A = pk(a), B = pk(b);
    cpk r;      // (-a.y b.y, a.y b.x) + (a.x b.x, a.x b.y); one statement: the compiler pads every asm statement whose result the
                // next instruction reads with a wait state, the pair itself needs none
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[1,0] neg_lo:[0,1]\n\t"
        "v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,0,0] op_sel_hi:[0,1,1]" : "=&v"(r) : "v"(A), "v"(B));
    return unpk(r);
#else
    return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
#endif
}
// a * conj(b)
VFI_HD float2 cmulc(float2 a, float2 b) {
#if defined(VFI_FFT_PACKED)
    const cpk A = pk(a), B = pk(b);
    cpk r;      // (a.y b.y, a.y b.x) + (a.x b.x, -a.x b.y)
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[1,0]\n\t"
        "v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,0,0] op_sel_hi:[0,1,1] neg_hi:[0,1,0]" : "=&v"(r) : "v"(A), "v"(B));
    return unpk(r);
#else
    return make_float2(a.x * b.x + a.y * b.y, a.y * b.x - a.x * b.y);
#endif
}
// conj(a * b)
VFI_HD float2 cmul_cj(float2 a, float2 b) {
#if defined(VFI_FFT_PACKED)
    const cpk A = pk(a), B = pk(b);
    cpk r;      // (-a.y b.y, -a.y b.x) + (a.x b.x, -a.x b.y)
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]\n\t"
        "v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,0,0] op_sel_hi:[0,1,1] neg_hi:[0,1,0]" : "=&v"(r) : "v"(A), "v"(B));
    return unpk(r);
#else
    return make_float2(a.x * b.x - a.y * b.y, -(a.x * b.y + a.y * b.x));
#endif
}
// a * (kr -+ i ki) with compile-time constants (forward: kr - i ki = the conjugate of exp(+i..) tables; inverse: kr + i ki);
// the constant pair sits in two SGPRs
template <bool INV> VFI_HD float2 cmul_k(float2 a, float kr, float ki) {
    const float kim = INV ? ki : -ki;
#if defined(VFI_FFT_PACKED)
    const cpk A = pk(a), K = {kr, kim};
    cpk r;      // (a.x kr, a.y kr) + (-a.y ki, a.x ki)
    asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]\n\t"
        "v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_lo:[1,0,0]" : "=&v"(r) : "v"(A), "s"(K));
    return unpk(r);
#else
    return make_float2(a.x * kr - a.y * kim, a.y * kr + a.x * kim);
#endif
}
// multiply by -i (forward transform) or +i (inverse)
template <bool INV> VFI_HD float2 rot(float2 a) { return INV ? make_float2(-a.y, a.x) : make_float2(a.y, -a.x); }
// t + rot(v), t - rot(v): one instruction each
template <bool INV> VFI_HD float2 add_rot(float2 t, float2 v) {
#if defined(VFI_FFT_PACKED)
    const cpk T = pk(t), V = pk(v);
    cpk r;
    if (INV) asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]" : "=v"(r) : "v"(T), "v"(V));       // (t.x - v.y, t.y + v.x)
    else asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]" : "=v"(r) : "v"(T), "v"(V));           // (t.x + v.y, t.y - v.x)
    return unpk(r);
#else
    return cadd(t, rot<INV>(v));
#endif
}
template <bool INV> VFI_HD float2 sub_rot(float2 t, float2 v) { return add_rot<!INV>(t, v); }
// W8^1 (forward: (1 - i)/sqrt2, inverse: (1 + i)/sqrt2) and W8^3 (forward: (-1 - i)/sqrt2, inverse: (-1 + i)/sqrt2)
template <bool INV> VFI_HD float2 w8_1(float2 a) {      // h * (a + rot(a))
#if defined(VFI_FFT_PACKED)
    const cpk A = pk(a), K = {0.70710678118654752440f, 0.70710678118654752440f};
    cpk r;
    if (INV) asm("v_pk_add_f32 %0, %1, %1 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]\n\tv_pk_mul_f32 %0, %0, %2" : "=&v"(r) : "v"(A), "s"(K));
    else asm("v_pk_add_f32 %0, %1, %1 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]\n\tv_pk_mul_f32 %0, %0, %2" : "=&v"(r) : "v"(A), "s"(K));
    return unpk(r);
#else
    return cscale(add_rot<INV>(a, a), 0.70710678118654752440f);
#endif
}
template <bool INV> VFI_HD float2 w8_3(float2 a) {      // -h * (a - rot(a))
#if defined(VFI_FFT_PACKED)
    const cpk A = pk(a), K = {-0.70710678118654752440f, -0.70710678118654752440f};
    cpk r;
    if (INV) asm("v_pk_add_f32 %0, %1, %1 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]\n\tv_pk_mul_f32 %0, %0, %2" : "=&v"(r) : "v"(A), "s"(K));
    else asm("v_pk_add_f32 %0, %1, %1 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]\n\tv_pk_mul_f32 %0, %0, %2" : "=&v"(r) : "v"(A), "s"(K));
    return unpk(r);
#else
    return cscale(sub_rot<INV>(a, a), -0.70710678118654752440f);
#endif
}

template <int R, bool INV> struct Dft;
template <bool INV> struct Dft<2, INV> {
    static VFI_HD void run(float2 *v) {
        const float2 a = v[0], b = v[1];
        v[0] = cadd(a, b); v[1] = csub(a, b);
    }
};
template <bool INV> struct Dft<3, INV> {
    static VFI_HD void run(float2 *v) {
        const float2 a = v[0], s = cadd(v[1], v[2]), d = csub(v[1], v[2]);
        const float2 m = cfma(s, -0.5f, a);
        const float2 e = cscale(d, 0.86602540378443864676f);
        v[0] = cadd(a, s); v[1] = add_rot<INV>(m, e); v[2] = sub_rot<INV>(m, e);
    }
};
template <bool INV> struct Dft<4, INV> {
    static VFI_HD void run(float2 *v) {
        const float2 t0 = cadd(v[0], v[2]), t1 = csub(v[0], v[2]), t2 = cadd(v[1], v[3]), d = csub(v[1], v[3]);
        v[0] = cadd(t0, t2); v[1] = add_rot<INV>(t1, d); v[2] = csub(t0, t2); v[3] = sub_rot<INV>(t1, d);
    }
};
template <bool INV> struct Dft<5, INV> {
    static VFI_HD void run(float2 *v) {
        constexpr float c1 = 0.30901699437494742410f, c2 = -0.80901699437494742410f;   // cos(2pi/5), cos(4pi/5)
        constexpr float s1 = 0.95105651629515357212f, s2 = 0.58778525229247312917f;    // sin(2pi/5), sin(4pi/5)
        const float2 a = v[0], t1 = cadd(v[1], v[4]), t2 = cadd(v[2], v[3]), t3 = csub(v[1], v[4]), t4 = csub(v[2], v[3]);
        const float2 m1 = cfma(t2, c2, cfma(t1, c1, a));
        const float2 m2 = cfma(t2, c1, cfma(t1, c2, a));
        const float2 e1 = cfma(t4, s2, cscale(t3, s1));
        const float2 e2 = cfma(t4, -s1, cscale(t3, s2));
        v[0] = cadd(cadd(a, t1), t2);
        v[1] = add_rot<INV>(m1, e1); v[4] = sub_rot<INV>(m1, e1);
        v[2] = add_rot<INV>(m2, e2); v[3] = sub_rot<INV>(m2, e2);
    }
};
template <bool INV> struct Dft<8, INV> {
    static VFI_HD void run(float2 *v) {
        float2 e[4] = {v[0], v[2], v[4], v[6]}, o[4] = {v[1], v[3], v[5], v[7]};
        Dft<4, INV>::run(e);
        Dft<4, INV>::run(o);
        o[1] = w8_1<INV>(o[1]); o[3] = w8_3<INV>(o[3]);
        v[0] = cadd(e[0], o[0]); v[4] = csub(e[0], o[0]);
        v[1] = cadd(e[1], o[1]); v[5] = csub(e[1], o[1]);
        v[2] = add_rot<INV>(e[2], o[2]); v[6] = sub_rot<INV>(e[2], o[2]);      // (o[2] turned by -+i)
        v[3] = cadd(e[3], o[3]); v[7] = csub(e[3], o[3]);
    }
};
template <bool INV> struct Dft<16, INV> {
    // k = q + 4s, input index c + 4r:  X[q+4s] = sum_c W4^(cs) * W16^(cq) * (sum_r x[c+4r] W4^(rq))
    static VFI_HD void run(float2 *v) {
        float2 t[4][4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            float2 col[4] = {v[c], v[c + 4], v[c + 8], v[c + 12]};
            Dft<4, INV>::run(col);
#pragma unroll
            for (int q = 0; q < 4; ++q) t[c][q] = col[q];
        }
        // W16^e, e = c*q: cos/sin(2 pi e / 16); a * (cr - i si) forward, a * (cr + i si) inverse
        constexpr float c1 = 0.92387953251128675613f, s1 = 0.38268343236508977173f, h = 0.70710678118654752440f;
        t[1][1] = cmul_k<INV>(t[1][1], c1, s1);            // e = 1
        t[1][2] = w8_1<INV>(t[1][2]);                      // e = 2
        t[1][3] = cmul_k<INV>(t[1][3], s1, c1);            // e = 3
        t[2][1] = w8_1<INV>(t[2][1]);                      // e = 2
        t[2][3] = w8_3<INV>(t[2][3]);                      // e = 6
        t[3][1] = cmul_k<INV>(t[3][1], s1, c1);            // e = 3
        t[3][2] = w8_3<INV>(t[3][2]);                      // e = 6
        t[3][3] = cmul_k<INV>(t[3][3], -c1, -s1);          // e = 9
        (void)h;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            // row q: {t[0][q], t[1][q], t[2][q] (turned by -+i for q = 2: e = 4), t[3][q]} through a radix-4 butterfly
            const float2 x0 = t[0][q], x1 = t[1][q], x2 = t[2][q], x3 = t[3][q];
            float2 t0, t1;
            if (q == 2) { t0 = add_rot<INV>(x0, x2); t1 = sub_rot<INV>(x0, x2); }
            else { t0 = cadd(x0, x2); t1 = csub(x0, x2); }
            const float2 t2 = cadd(x1, x3), d = csub(x1, x3);
            v[q] = cadd(t0, t2); v[q + 4] = add_rot<INV>(t1, d); v[q + 8] = csub(t0, t2); v[q + 12] = sub_rot<INV>(t1, d);
        }
    }
};

// composite radix R = R1 * R2 inside the registers (input index c + R1*r, output index q + R2*s):
//   X[q + R2*s] = sum_c W_R1^(c*s) * W_R^(c*q) * (sum_r x[c + R1*r] * W_R2^(r*q))
template <int R> struct Cis;       // exp(+2 pi i e / R) as {cos, sin}, e < R
template <> struct Cis<6> {
    static VFI_HD float2 at(int e) {
        constexpr float h = 0.86602540378443864676f;
        constexpr float t[6][2] = {{1.0f, 0.0f}, {0.5f, h}, {-0.5f, h}, {-1.0f, 0.0f}, {-0.5f, -h}, {0.5f, -h}};
        return make_float2(t[e][0], t[e][1]);
    }
};
template <> struct Cis<10> {
    static VFI_HD float2 at(int e) {
        constexpr float c1 = 0.80901699437494742410f, s1 = 0.58778525229247312917f;      // cos / sin (pi/5)
        constexpr float c2 = 0.30901699437494742410f, s2 = 0.95105651629515357212f;      // cos / sin (2 pi/5)
        constexpr float t[10][2] = {{1.0f, 0.0f}, {c1, s1}, {c2, s2}, {-c2, s2}, {-c1, s1}, {-1.0f, 0.0f}, {-c1, -s1}, {-c2, -s2}, {c2, -s2}, {c1, -s1}};
        return make_float2(t[e][0], t[e][1]);
    }
};
template <> struct Cis<9> {
    static VFI_HD float2 at(int e) {
        constexpr float t[9][2] = {{1.0f, 0.0f}, {0.76604444311897801345f, 0.64278760968653925190f}, {0.17364817766693041445f, 0.98480775301220802032f},
                                   {-0.5f, 0.86602540378443864676f}, {-0.93969262078590831688f, 0.34202014332566887944f},
                                   {-0.93969262078590831688f, -0.34202014332566887944f}, {-0.5f, -0.86602540378443864676f},
                                   {0.17364817766693041445f, -0.98480775301220802032f}, {0.76604444311897801345f, -0.64278760968653925190f}};
        return make_float2(t[e][0], t[e][1]);
    }
};
template <> struct Cis<12> {
    static VFI_HD float2 at(int e) {
        constexpr float h = 0.86602540378443864676f;
        constexpr float t[12][2] = {{1.0f, 0.0f}, {h, 0.5f}, {0.5f, h}, {0.0f, 1.0f}, {-0.5f, h}, {-h, 0.5f},
                                    {-1.0f, 0.0f}, {-h, -0.5f}, {-0.5f, -h}, {0.0f, -1.0f}, {0.5f, -h}, {h, -0.5f}};
        return make_float2(t[e][0], t[e][1]);
    }
};
template <> struct Cis<15> {
    static VFI_HD float2 at(int e) {
        constexpr float t[15][2] = {{1.0f, 0.0f}, {0.91354545764260086660f, 0.40673664307580015276f}, {0.66913060635885823757f, 0.74314482547739413310f},
                                    {0.30901699437494745126f, 0.95105651629515353118f}, {-0.10452846326765333207f, 0.99452189536827340088f},
                                    {-0.5f, 0.86602540378443864676f}, {-0.80901699437494734024f, 0.58778525229247324813f},
                                    {-0.97814760073380568883f, 0.20791169081775931482f}, {-0.97814760073380568883f, -0.20791169081775931482f},
                                    {-0.80901699437494734024f, -0.58778525229247324813f}, {-0.5f, -0.86602540378443864676f},
                                    {-0.10452846326765333207f, -0.99452189536827340088f}, {0.30901699437494745126f, -0.95105651629515353118f},
                                    {0.66913060635885823757f, -0.74314482547739413310f}, {0.91354545764260086660f, -0.40673664307580015276f}};
        return make_float2(t[e][0], t[e][1]);
    }
};
template <int R1, int R2, bool INV> struct DftComposite {
    static VFI_HD void run(float2 *v) {
        constexpr int R = R1 * R2;
        float2 t[R1][R2];
#pragma unroll
        for (int c = 0; c < R1; ++c) {
            float2 col[R2];
#pragma unroll
            for (int r = 0; r < R2; ++r) col[r] = v[c + R1 * r];
            Dft<R2, INV>::run(col);
#pragma unroll
            for (int q = 0; q < R2; ++q) {
                if (c * q == 0) { t[c][q] = col[q]; continue; }
                const float2 w = Cis<R>::at((c * q) % R);      // forward: exp(-i..) = conj
                t[c][q] = cmul_k<INV>(col[q], w.x, w.y);
            }
        }
#pragma unroll
        for (int q = 0; q < R2; ++q) {
            float2 row[R1];
#pragma unroll
            for (int c = 0; c < R1; ++c) row[c] = t[c][q];
            Dft<R1, INV>::run(row);
#pragma unroll
            for (int s = 0; s < R1; ++s) v[q + R2 * s] = row[s];
        }
    }
};
template <bool INV> struct Dft<6, INV> { static VFI_HD void run(float2 *v) { DftComposite<2, 3, INV>::run(v); } };
template <bool INV> struct Dft<10, INV> { static VFI_HD void run(float2 *v) { DftComposite<2, 5, INV>::run(v); } };
template <bool INV> struct Dft<9, INV> { static VFI_HD void run(float2 *v) { DftComposite<3, 3, INV>::run(v); } };
template <bool INV> struct Dft<12, INV> { static VFI_HD void run(float2 *v) { DftComposite<3, 4, INV>::run(v); } };
template <bool INV> struct Dft<15, INV> { static VFI_HD void run(float2 *v) { DftComposite<3, 5, INV>::run(v); } };

// exact floor(t / d) for 0 <= t < 2^15, 1 <= d < 2^15 through a float reciprocal (the distance of (t + 0.5) / d from
// an integer is >= 0.5 / d, far above the rounding error)
VFI_HD int fast_div(int t, float inv_d) { return (int)(((float)t + 0.5f) * inv_d); }
// products of small non-negative ints (< 2^23): full-rate v_mul_u32_u24 instead of the quarter-rate 32-bit multiply
VFI_HD int mul24(int a, int b) {
#ifdef __HIP_DEVICE_COMPILE__
    return __mul24(a, b);
#else
    return a * b;
#endif
}

// LDS layout of a line: one pad element after every 32 (a stage's scatter with a small stride -- stage 0 writes
// y[i*R + r] -- would otherwise put 32 lanes on one bank pair)
VFI_HD int phys(int a) { return a + (a >> 5); }
VFI_HD int padded_length(int m) { return m + ((m + 31) >> 5); }

// Who transforms which lines.  Cooperative: all 256 threads share all lines (workgroup barriers between the halves of a
// stage).  Wave-private (lines % 4 == 0): wave w owns lines w, w+4, ... and nobody else touches them between fill and
// drain, so its stages need no workgroup barrier at all -- the four waves drift apart and their LDS / VALU / wait phases
// overlap on the SIMDs instead of meeting at 2 x nstages barriers per transform.
struct Team {
    int tid, nthr;            // this thread's index in the team, team size (256 or 64)
    int line0, line_step;     // the team's lines: line0 + line_step * i, i < nlines
    int nlines;
};
VFI_HD Team team_all(int tid, int lines) { return Team{tid, kThreads, 0, 1, lines}; }
VFI_HD Team team_wave(int tid, int lines) {
    const int wave = tid >> 6;
    return Team{tid & 63, 64, wave, kThreads / 64, (lines - wave + kThreads / 64 - 1) / (kThreads / 64)};
}

template <int R> struct StageRegs {
    static constexpr int QB = ((kMaxElems + R - 1) / R + kThreads - 1) / kThreads;   // butterflies per thread
    static constexpr int QBB = ((kMaxElemsB + R - 1) / R + kThreads - 1) / kThreads;  // ... of a Bluestein plan
    static constexpr int NW = R > 8 ? 4 : (R > 4 ? 3 : (R > 2 ? 2 : 1));              // twiddles loaded per butterfly
    float2 v[QB][R];
};

// ---- one Stockham stage (FORWARD transform), in two halves around a workgroup synchronisation -------------------------
// The engine only runs forward stages: an inverse transform is conj(FFT(conj(x))), and the callers conjugate while they
// fill / drain the lines (free).  That halves the code.
// MULB: the inputs are replaced by conj(x * bfilt[index]) on the way in -- Bluestein's spectral product and the
// conjugation that turns the second forward transform into the inverse one, folded into its first stage.
// All loads of all of a thread's butterflies (LDS operands, twiddles, filter) are issued before any arithmetic, so their
// latencies overlap; out-of-range butterflies read butterfly 0 and are dropped by the scatter.
template <int R, bool MULB>
VFI_HD void stage_gather(StageRegs<R> &s, const Team tm, const float2 *buf, int pitch, int m, int p,
                         const float2 *tw, int tw_len, const float2 *__restrict__ bfilt) {
    constexpr int QB = MULB ? StageRegs<R>::QBB : StageRegs<R>::QB, NW = StageRegs<R>::NW;
    const int T = m / R, total = tm.nlines * T, twstep = m / (p * R);
    // how many of W^e, W^2e, W^4e, W^8e (e < T) lie inside the LDS table; the others are formed by squaring (uniform)
    const int nload = T <= tw_len ? (2 * T <= tw_len ? (4 * T <= tw_len ? (8 * T <= tw_len ? 4 : 3) : 2) : 1) : 0;
    const float inv_T = 1.0f / (float)T, inv_p = 1.0f / (float)p;
    float2 w[QB][NW];
    float2 bf[MULB ? QB : 1][MULB ? R : 1];
#pragma unroll
    for (int q = 0; q < QB; ++q) {
        if (tm.nthr * q < total) {                     // (uniform per wave)
            const int t0 = tm.tid + tm.nthr * q, t = t0 < total ? t0 : 0;
            const int ll = fast_div(t, inv_T), i = t - mul24(ll, T), k = i - mul24(fast_div(i, inv_p), p);
            const float2 *x = buf + mul24(tm.line0 + mul24(ll, tm.line_step), pitch);
#pragma unroll
            for (int r = 0; r < R; ++r) s.v[q][r] = x[phys(i + mul24(r, T))];
            if (MULB) {
#pragma unroll
                for (int r = 0; r < R; ++r) bf[MULB ? q : 0][MULB ? r : 0] = bfilt[i + mul24(r, T)];
            }
            if (p > 1) {                               // (first stage: every twiddle is 1)
                const int e = mul24(k, twstep);        // e < T
                w[q][0] = tw[e];
                if (NW > 1 && nload > 1) w[q][1] = tw[2 * e];
                if (NW > 2 && nload > 2) w[q][2] = tw[4 * e];
                if (NW > 3 && nload > 3) w[q][3] = tw[8 * e];
            }
        }
    }
#pragma unroll
    for (int q = 0; q < QB; ++q) {
        if (tm.nthr * q < total) {
            if (MULB) {
#pragma unroll
                for (int r = 0; r < R; ++r) s.v[q][r] = cmul_cj(s.v[q][r], bf[MULB ? q : 0][MULB ? r : 0]);
            }
            if (p > 1) {
                float2 wr[R > 2 ? R : 3];
                wr[1] = w[q][0];
                if (R > 2) wr[2] = nload > 1 ? w[q][NW > 1 ? 1 : 0] : cmul(wr[1], wr[1]);
                if (R > 4) wr[4] = nload > 2 ? w[q][NW > 2 ? 2 : 0] : cmul(wr[2], wr[2]);
                if (R > 8) wr[8] = nload > 3 ? w[q][NW > 3 ? 3 : 0] : cmul(wr[4], wr[4]);
                if (R > 3) wr[3] = cmul(wr[1], wr[2]);
                if (R > 5) wr[5] = cmul(wr[4], wr[1]);
                if (R > 6) { wr[6] = cmul(wr[4], wr[2]); wr[7] = cmul(wr[4], wr[3]); }
                if (R > 9) { wr[9] = cmul(wr[8], wr[1]); wr[10] = cmul(wr[8], wr[2]); wr[11] = cmul(wr[8], wr[3]); }
                if (R > 12) { wr[12] = cmul(wr[8], wr[4]); wr[13] = cmul(wr[8], wr[5]); wr[14] = cmul(wr[8], wr[6]); }
                if (R > 15) wr[15] = cmul(wr[8], wr[7]);
#pragma unroll
                for (int r = 1; r < R; ++r) s.v[q][r] = cmul(s.v[q][r], wr[r]);
            }
            Dft<R, false>::run(s.v[q]);
        }
    }
}

template <int R>
VFI_HD void stage_scatter(const StageRegs<R> &s, const Team tm, float2 *buf, int pitch, int m, int p) {
    const int T = m / R, total = tm.nlines * T;
    const float inv_T = 1.0f / (float)T, inv_p = 1.0f / (float)p;
#pragma unroll
    for (int q = 0; q < StageRegs<R>::QB; ++q) {
        const int t = tm.tid + tm.nthr * q;
        if (t < total) {
            const int ll = fast_div(t, inv_T), i = t - mul24(ll, T), g = fast_div(i, inv_p), k = i - mul24(g, p);
            float2 *y = buf + mul24(tm.line0 + mul24(ll, tm.line_step), pitch);
            const int a0 = mul24(g, mul24(p, R)) + k;
#pragma unroll
            for (int r = 0; r < R; ++r) y[phys(a0 + mul24(r, p))] = s.v[q][r];
        }
    }
}

// ---- what the CALLER does while it fills / drains the lines ------------------------------------------------------------
// fill : element j < n  <-  load_value<INV>(x[j], chirp[j])   (chirp ignored for smooth lengths); elements n <= j < m zero
// drain: X[k]           =   store_value<INV>(line[k], chirp[k])
// (inverse transforms by conjugation; Bluestein's chirp factors; the second transform of Bluestein's form leaves conj(Y).)
template <bool INV> VFI_HD float2 load_value(float2 x, float2 chirp, bool bluestein) {
    if (INV) x = cconj(x);
    return bluestein ? cmul(x, chirp) : x;
}
template <bool INV> VFI_HD float2 store_value(float2 b, float2 chirp, bool bluestein) {
    if (bluestein) b = cmulc(chirp, b);
    return INV ? cconj(b) : b;
}

// ---- filling / draining the lines: `total` elements, element e handled by thread e % 256 -------------------------------
// Chunks of kChunk elements per thread: all global loads of a chunk are issued before the first use (their latencies
// overlap), while the values in flight stay at a bounded number of registers.
struct Slot { float2 z, c; float s; };
constexpr int kChunk = 18;
template <typename LoadF, typename UseF>
__device__ __forceinline__ void for_slots(int total, LoadF load, UseF use) {
    for (int q0 = 0; q0 * kThreads < total; q0 += kChunk) {      // (uniform trip count)
        Slot v[kChunk];
#pragma unroll
        for (int q = 0; q < kChunk; ++q) {
            const int e = (int)threadIdx.x + kThreads * (q0 + q);
            if (kThreads * (q0 + q) < total) v[q] = load(e < total ? e : 0);
        }
#pragma unroll
        for (int q = 0; q < kChunk; ++q) {
            const int e = (int)threadIdx.x + kThreads * (q0 + q);
            if (e < total) use(e, v[q]);
        }
    }
}

// A store whose address is (uniform plane descriptor, uniform byte offset, per-thread byte offset): no 64-bit address
// arithmetic per element, and lanes that must not store pass an offset of 0xffffffff (out of the descriptor's range:
// dropped) instead of branching around the store.
__device__ __forceinline__ __amdgpu_buffer_rsrc_t plane_rsrc(const void *base, size_t bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(base), 0, (int)(unsigned)bytes, 0x00020000);
}
__device__ __forceinline__ void plane_store(__amdgpu_buffer_rsrc_t r, unsigned lane_bytes, int uniform_bytes, float2 v) {
    typedef unsigned u2 __attribute__((ext_vector_type(2)));
    u2 d;
    d.x = __float_as_uint(v.x);
    d.y = __float_as_uint(v.y);
    __builtin_amdgcn_raw_buffer_store_b64(d, r, lane_bytes, __builtin_amdgcn_readfirstlane(uniform_bytes), 0);
}
__device__ __forceinline__ void plane_store(__amdgpu_buffer_rsrc_t r, unsigned lane_bytes, int uniform_bytes, float v) {
    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), r, lane_bytes, __builtin_amdgcn_readfirstlane(uniform_bytes), 0);
}

// Three-phase variants for passes that STORE to global memory.  On gfx950 vmcnt counts loads and stores together, in
// order: a value that depends on a (conditional) load of the chunk -- a Bluestein chirp factor, a mask -- makes the compiler
// put `s_waitcnt vmcnt(0)` in front of every element's arithmetic, and once the first stores of the chunk are out, that
// wait sits out their round trips one by one.  Here the values of a whole chunk are formed first (`make`) and stored
// afterwards (`put`): the loads are consumed before the first store is issued and the chunk's stores go out back to back.
template <typename LoadF, typename MakeF, typename PutF>
__device__ __forceinline__ void for_slots(int total, LoadF load, MakeF make, PutF put) {
    for (int q0 = 0; q0 * kThreads < total; q0 += kChunk) {      // (uniform trip count)
        Slot v[kChunk];
#pragma unroll
        for (int q = 0; q < kChunk; ++q) {
            const int e = (int)threadIdx.x + kThreads * (q0 + q);
            if (kThreads * (q0 + q) < total) v[q] = load(e < total ? e : 0);
        }
        decltype(make(0, v[0])) o[kChunk];
#pragma unroll
        for (int q = 0; q < kChunk; ++q) {
            const int e = (int)threadIdx.x + kThreads * (q0 + q);
            if (e < total) o[q] = make(e, v[q]);
        }
#pragma unroll
        for (int q = 0; q < kChunk; ++q) {
            const int e = (int)threadIdx.x + kThreads * (q0 + q);
            if (e < total) put(e, o[q]);
        }
    }
}

// ---- structured slot walks: almost no per-element index arithmetic ---------------------------------------------------
// Column tile (C = 1 << shift columns, h rows; line = column): thread t owns column cc = t & (C-1) and the rows
// u = u0 + step*q (u0 = t >> shift, step = 256 >> shift).  The LDS index of (cc, u) is cc*pitch + phys(u) and
// phys(u0 + step*q) = phys(u0) + step*q + ((step*q) >> 5) exactly (the low five bits of step*q and u0 never carry), so it is
// a per-thread base plus a UNIFORM term; global offsets are a per-thread base plus uq * row_pitch (uniform) as well.
//   load(u, uq, cc) -> Slot          (uq = step*q, uniform; u = u0 + uq)
//   use(u, uq, cc, lds_index, slot)
template <typename LoadF, typename UseF>
__device__ __forceinline__ void for_tile(int h, int shift, int pitch, LoadF load, UseF use) {
    const int t = threadIdx.x, cc = t & ((1 << shift) - 1), u0 = t >> shift, step = kThreads >> shift;
    const int lbase = mul24(cc, pitch) + phys(u0);
    const int nq = (h - 1) / step + 1;                            // (uniform) slots per thread
    for (int q0 = 0; q0 < nq; q0 += kChunk) {
        Slot v[kChunk];
#pragma unroll
        for (int q = 0; q < kChunk; ++q) {
            if (q0 + q < nq) {
                const int uq = step * (q0 + q), u = u0 + uq;
                v[q] = load(u < h ? u : u0, u < h ? uq : 0, cc);
            }
        }
#pragma unroll
        for (int q = 0; q < kChunk; ++q) {
            if (q0 + q < nq) {
                const int uq = step * (q0 + q), u = u0 + uq;
                if (u < h) use(u, uq, cc, lbase + uq + (uq >> 5), v[q]);
            }
        }
    }
}
//   make(u, uq, cc, lds_index, slot) -> Out ; put(u, uq, cc, out)      (storing passes, see for_slots)
template <typename LoadF, typename MakeF, typename PutF>
__device__ __forceinline__ void for_tile(int h, int shift, int pitch, LoadF load, MakeF make, PutF put) {
    const int t = threadIdx.x, cc = t & ((1 << shift) - 1), u0 = t >> shift, step = kThreads >> shift;
    const int lbase = mul24(cc, pitch) + phys(u0);
    const int nq = (h - 1) / step + 1;                            // (uniform) slots per thread
    for (int q0 = 0; q0 < nq; q0 += kChunk) {
        Slot v[kChunk];
#pragma unroll
        for (int q = 0; q < kChunk; ++q) {
            if (q0 + q < nq) {
                const int uq = step * (q0 + q), u = u0 + uq;
                v[q] = load(u < h ? u : u0, u < h ? uq : 0, cc);
            }
        }
        decltype(make(0, 0, 0, 0, v[0])) o[kChunk];
#pragma unroll
        for (int q = 0; q < kChunk; ++q) {
            if (q0 + q < nq) {
                const int uq = step * (q0 + q), u = u0 + uq;
                if (u < h) o[q] = make(u, uq, cc, lbase + uq + (uq >> 5), v[q]);
            }
        }
#pragma unroll
        for (int q = 0; q < kChunk; ++q) {
            if (q0 + q < nq) {
                const int uq = step * (q0 + q), u = u0 + uq;
                if (u < h) put(u, uq, cc, o[q]);
            }
        }
    }
}
// Rows (`lines` rows of n >= 256 elements; line = row): thread t owns the columns j = t + 256*qq of every row.
// phys(t + 256*qq) = phys(t) + 264*qq, so the LDS index is phys(t) + (l*pitch + 264*qq), the bracket uniform.
//   load(l, j, jq) -> Slot           (jq = 256*qq uniform; j = t + jq)
//   use(l, j, jq, lds_index, slot)
template <typename LoadF, typename UseF>
__device__ __forceinline__ void for_rows(int lines, int n, int pitch, LoadF load, UseF use) {
    const int t = threadIdx.x, pt = phys(t);
    const int qn = (n - 1) / kThreads + 1, ns = lines * qn;       // (uniform)
    for (int s0 = 0; s0 < ns; s0 += kChunk) {
        Slot v[kChunk];
#pragma unroll
        for (int q = 0; q < kChunk; ++q) {
            if (s0 + q < ns) {
                const int l = (s0 + q) / qn, qq = (s0 + q) - l * qn, jq = kThreads * qq, j = t + jq;
                v[q] = load(l, j < n ? j : 0, j < n ? jq : -t);
            }
        }
#pragma unroll
        for (int q = 0; q < kChunk; ++q) {
            if (s0 + q < ns) {
                const int l = (s0 + q) / qn, qq = (s0 + q) - l * qn, jq = kThreads * qq, j = t + jq;
                if (j < n) use(l, j, jq, pt + (l * pitch + (kThreads + kThreads / 32) * qq), v[q]);
            }
        }
    }
}

//   make(l, j, jq, lds_index, slot) -> Out ; put(l, j, jq, out)        (storing passes, see for_slots)
template <typename LoadF, typename MakeF, typename PutF>
__device__ __forceinline__ void for_rows(int lines, int n, int pitch, LoadF load, MakeF make, PutF put) {
    const int t = threadIdx.x, pt = phys(t);
    const int qn = (n - 1) / kThreads + 1, ns = lines * qn;       // (uniform)
    for (int s0 = 0; s0 < ns; s0 += kChunk) {
        Slot v[kChunk];
#pragma unroll
        for (int q = 0; q < kChunk; ++q) {
            if (s0 + q < ns) {
                const int l = (s0 + q) / qn, qq = (s0 + q) - l * qn, jq = kThreads * qq, j = t + jq;
                v[q] = load(l, j < n ? j : 0, j < n ? jq : -t);
            }
        }
        decltype(make(0, 0, 0, 0, v[0])) o[kChunk];
#pragma unroll
        for (int q = 0; q < kChunk; ++q) {
            if (s0 + q < ns) {
                const int l = (s0 + q) / qn, qq = (s0 + q) - l * qn, jq = kThreads * qq, j = t + jq;
                if (j < n) o[q] = make(l, j, jq, pt + (l * pitch + (kThreads + kThreads / 32) * qq), v[q]);
            }
        }
#pragma unroll
        for (int q = 0; q < kChunk; ++q) {
            if (s0 + q < ns) {
                const int l = (s0 + q) / qn, qq = (s0 + q) - l * qn, jq = kThreads * qq, j = t + jq;
                if (j < n) put(l, j, jq, o[q]);
            }
        }
    }
}

// ---- device entry: all 256 threads of the workgroup call it; the lines must be filled and synchronised -----------------
// Stages are real function calls (one body per radix, shared by every kernel and every position in the stage list):
// inlined, the ~10 unrolled bodies of a kernel made the register allocator spill kilobytes per lane.
// Workgroup barrier that orders LDS traffic only.  __syncthreads() also drains every outstanding GLOBAL access
// (s_waitcnt vmcnt(0)): with it, the stores of one band's drain would have to land before the next band could start.
__device__ __forceinline__ void lds_barrier() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

// Within one wave LDS operations execute in issue order, so a wave-private stage only has to stop the COMPILER from
// moving its gather above the previous scatter (and wait for the reads it consumes, which the data dependence does).
__device__ __forceinline__ void wave_lds_order() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront", "local");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront", "local");
}

template <int R, bool MULB>
__device__ __noinline__ void stage(float2 *buf, int lines, int pitch, int m, int p, const float2 *tw, int tw_len,
                                   const float2 *__restrict__ bfilt, int wave_private) {
    StageRegs<R> s;
    const Team tm = wave_private ? team_wave(threadIdx.x, lines) : team_all(threadIdx.x, lines);
    stage_gather<R, MULB>(s, tm, buf, pitch, m, p, tw, tw_len, bfilt);
    if (wave_private) wave_lds_order(); else lds_barrier();
    stage_scatter<R>(s, tm, buf, pitch, m, p);
    if (wave_private) wave_lds_order(); else lds_barrier();
}
__device__ __forceinline__ void stage_any(int R, float2 *buf, int lines, int pitch, int m, int p, const float2 *tw, int tw_len, int wp) {
    switch (R) {
        case 16: stage<16, false>(buf, lines, pitch, m, p, tw, tw_len, nullptr, wp); break;
        case 15: stage<15, false>(buf, lines, pitch, m, p, tw, tw_len, nullptr, wp); break;
        case 12: stage<12, false>(buf, lines, pitch, m, p, tw, tw_len, nullptr, wp); break;
        case 9: stage<9, false>(buf, lines, pitch, m, p, tw, tw_len, nullptr, wp); break;
        case 8: stage<8, false>(buf, lines, pitch, m, p, tw, tw_len, nullptr, wp); break;
        case 4: stage<4, false>(buf, lines, pitch, m, p, tw, tw_len, nullptr, wp); break;
        case 2: stage<2, false>(buf, lines, pitch, m, p, tw, tw_len, nullptr, wp); break;
        case 3: stage<3, false>(buf, lines, pitch, m, p, tw, tw_len, nullptr, wp); break;
        default: stage<5, false>(buf, lines, pitch, m, p, tw, tw_len, nullptr, wp); break;
    }
}
// The workgroup's copy of the twiddle table: pl.tw_len entries behind the lines (call once, before the first fft_lines;
// the barrier that publishes the filled lines publishes it too).
__device__ __forceinline__ void load_twiddles(float2 *twl, const Plan1D &pl) {
    for (int k0 = 0; k0 < pl.tw_len; k0 += 4 * kThreads) {      // 4 loads in flight per thread
        float2 t[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int k = k0 + j * kThreads + (int)threadIdx.x;
            t[j] = pl.tw[k < pl.tw_len ? k : 0];
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int k = k0 + j * kThreads + (int)threadIdx.x;
            if (k < pl.tw_len) twl[k] = t[j];
        }
    }
}
// `lines` FORWARD transforms at buf[l * pitch + phys(j)]; results in place (see load_value / store_value).
// twl: the LDS twiddle table filled by load_twiddles.  The lines must be filled and published by a workgroup barrier;
// on return every line is complete only after the CALLER's next workgroup barrier when the wave-private mode ran
// (lines % 4 == 0): the callers barrier before draining anyway.
__device__ __forceinline__ void fft_lines(float2 *buf, int lines, int pitch, const Plan1D &pl, const float2 *twl) {
    const int wp = (lines & (kThreads / 64 - 1)) == 0 ? 1 : 0;
    int p = 1;
    for (int s = 0; s < pl.nstages; ++s) {
        stage_any(pl.radix[s], buf, lines, pitch, pl.m, p, twl, pl.tw_len, wp);
        p *= pl.radix[s];
    }
    if (pl.bluestein) {
        switch (pl.radix[0]) {                  // (a Bluestein length is 2^k or 3 * 2^k, powers of two first: 16, 8, 4, 2 only)
            case 16: stage<16, true>(buf, lines, pitch, pl.m, 1, twl, pl.tw_len, pl.bfilt, wp); break;
            case 8: stage<8, true>(buf, lines, pitch, pl.m, 1, twl, pl.tw_len, pl.bfilt, wp); break;
            case 4: stage<4, true>(buf, lines, pitch, pl.m, 1, twl, pl.tw_len, pl.bfilt, wp); break;
            default: stage<2, true>(buf, lines, pitch, pl.m, 1, twl, pl.tw_len, pl.bfilt, wp); break;
        }
        p = pl.radix[0];
        for (int s = 1; s < pl.nstages; ++s) {
            stage_any(pl.radix[s], buf, lines, pitch, pl.m, p, twl, pl.tw_len, wp);
            p *= pl.radix[s];
        }
    }
    if (wp) lds_barrier();      // the drain is cooperative again
}

// ---- host-side planning ----------------------------------------------------------------------------------------------
// radices for a smooth length: as few, as large stages as possible (16, 15, 9, 8, then 5, 4, 3, 2); false when n has
// another prime factor.  A power of two only gets powers of two (Bluestein's spectral product sits in its first stage).
inline bool factor_smooth(int n, int *radix, int *nstages) {
    int m = n, ns = 0;
    for (int r : {16, 15, 9, 8, 5, 4, 3, 2})
        while (m % r == 0) {
            if (ns >= kMaxStages) return false;
            radix[ns++] = r;
            m /= r;
        }
    if (ns >= 2 && radix[ns - 1] == 2)          // (..., 16, 2) -> (..., 8, 4): no stage with T = m/2 butterflies per line
        for (int i = ns - 2; i >= 0; --i)       // (its twiddle index range would exceed a quarter table)
            if (radix[i] == 16) { radix[i] = 8; radix[ns - 1] = 4; break; }
    *nstages = ns;
    return m == 1 && n >= 2;
}
// entries of the LDS twiddle table a plan needs at least: every stage reads W^e for e < m / R
inline int min_twiddle_entries(const Plan1D &pl) {
    int need = 1;
    for (int s = 1; s < pl.nstages; ++s) need = pl.m / pl.radix[s] > need ? pl.m / pl.radix[s] : need;
    return need;
}
// Bluestein's convolution length: the smallest of 2^k and 3 * 2^k (k >= 1) that holds 2n - 1 points -- three quarters of
// the next power of two when that is enough, at the same number of stages (factor_bluestein).
inline int bluestein_length(int n) {
    int M = 1;
    while (M < 2 * n - 1) M *= 2;
    return (M >= 32 && M / 4 * 3 >= 2 * n - 1) ? M / 4 * 3 : M;
}
// radices of a Bluestein length: the power-of-two stages first (the spectral product sits in the first stage, which the
// engine instantiates for radices 16, 8, 4, 2), one radix-12 stage last for 3 * 2^k
inline bool factor_bluestein(int m, int *radix, int *nstages) {
    if (m % 3) return factor_smooth(m, radix, nstages);
    if (m % 12 || m / 12 < 2 || !factor_smooth(m / 12, radix, nstages) || *nstages >= kMaxStages) return false;
    for (int i = 0; i < *nstages; ++i)
        if (radix[i] != 16 && radix[i] != 8 && radix[i] != 4 && radix[i] != 2) return false;
    radix[(*nstages)++] = 12;
    return true;
}
// LDS line pitch: row passes store lines back to back; column passes fill `tile` lines with consecutive lanes going
// ACROSS the lines, so the pitch is chosen = 32/tile (mod 32) elements: the lanes of one wave then cover all banks
inline int row_pitch(const Plan1D &pl) { return padded_length(pl.m); }
inline int col_pitch(const Plan1D &pl, int tile) { return ((padded_length(pl.m) + 31) & ~31) + (tile < 32 ? 32 / tile : 1); }
inline int max_elems(const Plan1D &pl) { return pl.bluestein ? kMaxElemsB : kMaxElems; }
// LDS twiddle entries: half the table (a quarter for long powers of two); powers that fall outside are squared up
inline int twiddle_entries(int m) { return (m % 4 == 0 && m >= 2048) ? m / 4 : (m + 1) / 2; }
// how many rows of this plan fit into one workgroup's buffer (lines + twiddle table <= kLdsElems)
inline int max_lines(const Plan1D &pl) {
    const int a = max_elems(pl) / pl.m, b = (kLdsElems - pl.tw_len) / row_pitch(pl);
    if (a >= 1 && b < 1 && row_pitch(pl) + 33 + pl.tw_len <= kLdsElemsMax) return 1;     // one workgroup per CU
    return a < b ? a : b;
}
// LDS bytes of a row / column pass (lines, the twiddle table, `extra` bytes of the caller's own)
inline size_t row_lds_bytes(const Plan1D &pl, int lines, size_t extra = 0) {
    return ((size_t)lines * row_pitch(pl) + pl.tw_len) * sizeof(float2) + extra;
}
inline size_t col_lds_bytes(const Plan1D &pl, int tile) { return ((size_t)tile * col_pitch(pl, tile) + pl.tw_len) * sizeof(float2); }

}  // namespace fft
}  // namespace vfi

// ---- generic batched passes (vfi_fft.hip) ---------------------------------------------------------------------------
namespace vfi {
namespace fft {

// how a row pass reads / writes its rows
enum RowLoad { kLoadComplex = 0, kLoadReal = 1, kLoadHalf = 2 };      // Half: first n/2+1 entries given, rest by Hermitian symmetry
enum RowStore { kStoreComplex = 0, kStoreReal = 1, kStoreHalf = 2 };  // Half: only the first n/2+1 entries are written

struct RowArgs {
    Plan1D pl;
    const void *src;
    void *dst;
    long long rows;            // rows of all planes together (they are `pl.n` long)
    int src_pitch, dst_pitch;  // elements between consecutive rows
    int lines;                 // rows per workgroup
    float scale;               // applied on store
};
struct ColArgs {
    Plan1D pl;                 // pl.n = column length (rows of a plane)
    float2 *data;              // [planes][pl.n][ld], transformed in place
    int planes, cols, ld;      // cols <= ld columns are transformed
    int tile;                  // columns per workgroup (power of two)
    float scale;
};

// builds the tables of one length on the device (double precision on the host); every allocation is appended to `owned`
int make_plan(int n, Plan1D *out, void (*own)(void *ctx, void *dev), void *ctx);

int launch_rows(const RowArgs &a, RowLoad load, RowStore store, bool inverse, hipStream_t s);
int launch_cols(const ColArgs &a, bool inverse, hipStream_t s);
int rows_per_group(const Plan1D &pl, long long total_rows);
int cols_per_group(const Plan1D &pl, int cols);

}  // namespace fft
}  // namespace vfi
