/*
 * ORACLE -- TEST INFRASTRUCTURE ONLY.  Never imported, linked or executed by the
 * product path (fusion-method-for-video-frame-interpolation_amd/); only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may use it.
 *
 * CPU restatement of the reference's AdaCoF deformable sampling and of the
 * occlusion blend + flow-variance mask that follow it.
 *
 *   oracle_adacof_forward      <- reference src/adacof/cupy_module/adacof.py:6-65
 *                                 (kernel_AdaCoF_updateOutput; the reference has NO CPU
 *                                 path, adacof.py:356-357 raises NotImplementedError)
 *   oracle_adacof_blend_mask   <- reference src/fusion_net/fusion_adacofnet.py:198-213
 *
 * Pinned by tests/golden/adacof_sampling_*.npz, which were produced in the build
 * container from the reference's own specialised kernel text (see
 * tests/golden/make_golden.py).  Built with -ffp-contract=off so the arithmetic is
 * the plain left-to-right fp32 evaluation of the reference expression.
 */
#include <stddef.h>

static inline int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

/* adacof.py:15-63: one output element per (n, c, i, j); F*F taps; (int) truncation
 * toward zero (adacof.py:28-29), each of the four tap indices clamped independently
 * (adacof.py:31-53), weights from the un-clamped fractional parts (adacof.py:55-60). */
void oracle_adacof_forward(const float *input, const float *weight,
                           const float *offset_i, const float *offset_j, float *output,
                           int N, int C, int Hin, int Win, int H, int W, int F, int dilation)
{
    const size_t plane = (size_t)H * W;
    for (int n = 0; n < N; ++n)
        for (int c = 0; c < C; ++c) {
            const float *in = input + ((size_t)n * C + c) * Hin * Win;
            for (int i = 0; i < H; ++i)
                for (int j = 0; j < W; ++j) {
                    float acc = 0.0f;
                    for (int k = 0; k < F; ++k)
                        for (int l = 0; l < F; ++l) {
                            const size_t t = ((size_t)n * F * F + (size_t)(k * F + l)) * plane + (size_t)i * W + j;
                            const float w = weight[t];
                            const float alpha = offset_i[t];
                            const float beta = offset_j[t];
                            const int A = (int)alpha;
                            const int B = (int)beta;
                            const int i0 = clampi(i + k * dilation + A, 0, Hin - 1);
                            const int j0 = clampi(j + l * dilation + B, 0, Win - 1);
                            const int i1 = clampi(i + k * dilation + A + 1, 0, Hin - 1);
                            const int j1 = clampi(j + l * dilation + B + 1, 0, Win - 1);
                            const float fa = alpha - (float)A;
                            const float fb = beta - (float)B;
                            acc += w * (in[(size_t)i0 * Win + j0] * (1 - fa) * (1 - fb) +
                                        in[(size_t)i1 * Win + j0] * fa * (1 - fb) +
                                        in[(size_t)i0 * Win + j1] * (1 - fa) * fb +
                                        in[(size_t)i1 * Win + j1] * fa * fb);
                        }
                    output[((size_t)n * C + c) * plane + (size_t)i * W + j] = acc;
                }
        }
}

/* fusion_adacofnet.py:198     frame1 = Occ*t1 + (1-Occ)*t2
 * fusion_adacofnet.py:201-208 Mean = sum_k W*dP ; Var = sum_k W*(Mean-dP)^2   (alpha and beta)
 * fusion_adacofnet.py:211-213 mask = clip(max(Var1.sum(0), Var2.sum(0)), 0, 20) / 20          */
void oracle_adacof_blend_mask(const float *t1, const float *t2, const float *occ,
                              const float *w1, const float *a1, const float *b1,
                              const float *w2, const float *a2, const float *b2,
                              float *frame, float *mask, int N, int C, int H, int W, int K)
{
    const size_t plane = (size_t)H * W;
    for (int n = 0; n < N; ++n)
        for (size_t p = 0; p < plane; ++p) {
            const float o = occ[(size_t)n * plane + p];
            for (int c = 0; c < C; ++c) {
                const size_t q = ((size_t)n * C + c) * plane + p;
                frame[q] = o * t1[q] + (1 - o) * t2[q];
            }
            float var[2];
            for (int side = 0; side < 2; ++side) {
                const float *w = side ? w2 : w1, *a = side ? a2 : a1, *b = side ? b2 : b1;
                float ma = 0.0f, mb = 0.0f;
                for (int k = 0; k < K; ++k) {
                    const size_t t = ((size_t)n * K + k) * plane + p;
                    ma += w[t] * a[t];
                    mb += w[t] * b[t];
                }
                float va = 0.0f, vb = 0.0f;
                for (int k = 0; k < K; ++k) {
                    const size_t t = ((size_t)n * K + k) * plane + p;
                    const float da = ma - a[t], db = mb - b[t];
                    va += w[t] * (da * da);
                    vb += w[t] * (db * db);
                }
                var[side] = va + vb;
            }
            float m = var[0] > var[1] ? var[0] : var[1];
            m = m < 0.0f ? 0.0f : (m > 20.0f ? 20.0f : m);
            mask[(size_t)n * plane + p] = m / 20.0f;
        }
}
