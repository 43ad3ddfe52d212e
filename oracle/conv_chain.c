/* TEST INFRASTRUCTURE ONLY (checker; never linked into or called by the product).
 *
 * Bit-exact CPU restatement of the contraction kernels' ARITHMETIC (speakerguard_amd/csrc/k_conv_gemm.hip): the dilated
 * Conv1d of the TDNN / AudioNet layers (reference model/_xv_plda/xvecTDNN.py:16-33, model/audionet_csine.py:66-118) and
 * the data gradient autograd derives for it, as ONE float32 fmaf chain per output element.
 *
 * tools/native/mfma_order.hip shows that v_mfma_f32_32x32x2_f32 and v_mfma_f32_16x16x4_f32 are sequential fused
 * multiply-adds over their k values, bit for bit.  Every kernel variant feeds the k values of a tap in the same order --
 * chunks of 32 ascending, k-groups of 8 ascending, inside a group (0, 4, 1, 5, 2, 6, 3, 7) -- taps ascending, and a tap
 * row outside its utterance enters as zeros (the MFMA is still executed: fmaf(0, w, acc)).  So the device result must
 * equal this loop bit for bit, for every launch strategy and batch size (tests/test_gpu_conv.py).
 *
 * This pins the kernels' own arithmetic contract; the agreement with the reference's conv1d (whose summation order is
 * cuDNN's / MKL's business) is checked separately in float64 (oracle/conv_rows.py, tests/test_oracle_conv.py).
 *
 *   gcc -O2 -mfma -fopenmp -ffp-contract=off -shared -fPIC oracle/conv_chain.c -o oracle/libconv_chain.so -lm
 *   (-mfma: fmaf becomes the hardware's fused multiply-add, still one rounding; -ffp-contract=off: nothing ELSE is fused)
 */
#include <math.h>
#include <stddef.h>

static const int kOrder[8] = {0, 4, 1, 5, 2, 6, 3, 7};

/* a (B*Ta, Kc), w (taps*Kc, N), out (B*Tc, N); epi 0: none, 1: max(acc + bias[n], 0), 2: mask[m][n] > 0 ? acc : 0 */
void sg_conv_chain(const float* a, const float* w, const float* bias, const float* mask, float* out, int B, int Ta, int Tc,
                   int Kc, int N, int taps, int tap_step, int tap_base, int epi) {
#pragma omp parallel for schedule(static)
    for (long bt = 0; bt < (long)B * Tc; ++bt) {
        const int b = (int)(bt / Tc), t = (int)(bt - (long)b * Tc);
        const size_t m = (size_t)bt;
        for (int n = 0; n < N; ++n) {
            float acc = 0.f;
            for (int j = 0; j < taps; ++j) {
                const int src = t + tap_base + j * tap_step;
                const float* row = (src >= 0 && src < Ta) ? a + ((size_t)b * Ta + src) * Kc : NULL;
                for (int kc = 0; kc < Kc; kc += 32)
                    for (int kg = 0; kg < 4; ++kg)
                        for (int s = 0; s < 8; ++s) {
                            const int k = kc + 8 * kg + kOrder[s];
                            acc = fmaf(row ? row[k] : 0.f, w[((size_t)j * Kc + k) * N + n], acc);
                        }
            }
            if (epi == 1) acc = fmaxf(acc + bias[n], 0.f);
            if (epi == 2) acc = mask[m * N + n] > 0.f ? acc : 0.f;
            out[m * N + n] = acc;
        }
    }
}

/* FeCo k-means assignment scores (speakerguard_amd/csrc/k_feco.hip, contract version 2): score(i, j) = h[j] + x'_i . c_j as
 * ONE fmaf chain in the k order above (Dp = 32 or 64 padded dimensions, pad entries zero).  The reference delegates the
 * clustering to a randomly initialised third-party k-means (defense/feature_level.py:185-203): this restates the
 * library's own contract, nothing of the reference.  xc (F, Dp), cc (k, Dp), h (k), out (F, k). */
void sg_feco_scores(const float* xc, const float* cc, const float* h, float* out, int F, int k, int Dp) {
#pragma omp parallel for schedule(static)
    for (int i = 0; i < F; ++i)
        for (int j = 0; j < k; ++j) {
            float acc = h[j];
            for (int g = 0; g < Dp; g += 8)
                for (int s = 0; s < 8; ++s) acc = fmaf(cc[(size_t)j * Dp + g + kOrder[s]], xc[(size_t)i * Dp + g + kOrder[s]], acc);
            out[(size_t)i * k + j] = acc;
        }
}
