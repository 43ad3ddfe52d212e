// Driver of the sanitizer build of the C-ABI's host half (`make asan`; see hip_host_double.cpp): walks the entry points
// of include/speakerguard_hip.h with NULL / malformed / valid arguments under AddressSanitizer + UBSan.  Any sanitizer
// report aborts the process (-fno-sanitize-recover); a failed expectation returns 1.  "Device" buffers are host
// buffers sized exactly as the header says, so an entry point that uploads or downloads past a documented extent is an
// ASan error here.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "speakerguard_hip.h"

extern "C" long hipdouble_launches();
extern "C" long hipdouble_live_allocs();

static int g_fail = 0;
#define EXPECT(cond)                                                                 \
    do {                                                                             \
        if (!(cond)) {                                                               \
            std::fprintf(stderr, "FAILED %s:%d: %s\n", __FILE__, __LINE__, #cond);   \
            ++g_fail;                                                                \
        }                                                                            \
    } while (0)

static std::vector<float> rnd(size_t n, unsigned seed, float scale = 0.1f, float shift = 0.f) {
    std::vector<float> v(n);
    unsigned s = seed * 2654435761u + 12345u;
    for (size_t i = 0; i < n; ++i) {
        s = s * 1664525u + 1013904223u;
        v[i] = shift + scale * ((float)(s >> 8) / 8388608.0f - 1.0f);
    }
    return v;
}

struct XvModel {
    std::vector<float> w[5], b[5], mean[5], var[5], fc1w, fc1b, emean, lda, pmean, ptrans, ppsi, enroll;
    sg_xv_weights desc{};
    XvModel(int D, int S) {
        const int cin[5] = {30, 512, 512, 512, 512}, cout[5] = {512, 512, 512, 512, 1500}, k[5] = {5, 5, 7, 1, 1};
        for (int l = 0; l < 5; ++l) {
            w[l] = rnd((size_t)cout[l] * cin[l] * k[l], 10 + l);
            b[l] = rnd(cout[l], 20 + l);
            mean[l] = rnd(cout[l], 30 + l);
            var[l] = rnd(cout[l], 40 + l, 0.5f, 1.0f);
            desc.tdnn_weight[l] = w[l].data(); desc.tdnn_bias[l] = b[l].data();
            desc.bn_mean[l] = mean[l].data(); desc.bn_var[l] = var[l].data();
        }
        fc1w = rnd((size_t)512 * 3000, 1); fc1b = rnd(512, 2); emean = rnd(512, 3); lda = rnd((size_t)D * 513, 4);
        pmean = rnd(D, 5); ptrans = rnd((size_t)D * D, 6); ppsi = rnd(D, 7, 0.5f, 1.0f); enroll = rnd((size_t)S * D, 8);
        desc.fc1_weight = fc1w.data(); desc.fc1_bias = fc1b.data(); desc.emb_mean = emean.data(); desc.lda = lda.data();
        desc.plda_mean = pmean.data(); desc.plda_transform = ptrans.data(); desc.plda_psi = ppsi.data(); desc.enroll = enroll.data();
        desc.D = D; desc.S = S; desc.bn_eps = 1e-5f; desc.threshold = -INFINITY;
    }
};

struct AnModel {
    std::vector<float> c1w, c1b, bn1[4], cw[7], cb[7], g[7], be[7], mu[7], va[7], fcw, fcb;
    sg_an_weights desc{};
    explicit AnModel(int S) {
        const int cin[7] = {32, 64, 128, 128, 128, 128, 64}, cout[7] = {64, 128, 128, 128, 128, 64, 32};
        c1w = rnd(25, 50); c1b = rnd(1, 51);
        for (int i = 0; i < 4; ++i) { bn1[i] = rnd(1, 52 + i, 0.1f, i == 3 || i == 0 ? 1.f : 0.f); desc.bn1[i] = bn1[i].data(); }
        desc.conv1_weight = c1w.data(); desc.conv1_bias = c1b.data();
        for (int l = 0; l < 7; ++l) {
            cw[l] = rnd((size_t)cout[l] * cin[l] * 3, 60 + l); cb[l] = rnd(cout[l], 70 + l); g[l] = rnd(cout[l], 80 + l, 0.1f, 1.f);
            be[l] = rnd(cout[l], 90 + l); mu[l] = rnd(cout[l], 100 + l); va[l] = rnd(cout[l], 110 + l, 0.3f, 1.f);
            desc.conv_weight[l] = cw[l].data(); desc.conv_bias[l] = cb[l].data(); desc.bn_weight[l] = g[l].data();
            desc.bn_bias[l] = be[l].data(); desc.bn_mean[l] = mu[l].data(); desc.bn_var[l] = va[l].data();
        }
        fcw = rnd((size_t)S * 32, 120); fcb = rnd(S, 121);
        desc.fc_weight = fcw.data(); desc.fc_bias = fcb.data(); desc.num_class = S; desc.bn_eps = 1e-5f;
    }
};

int main() {
    // ---- no context ------------------------------------------------------------------------------------------
    EXPECT(sg_version() == 100);
    EXPECT(sg_xv_num_frames(48000) == 300 && sg_xv_num_frames(100) == 0 && sg_an_num_frames(48000) == 300);
    EXPECT(sg_last_error(nullptr) != nullptr);
    sg_destroy(nullptr);
    EXPECT(sg_create(0, nullptr) == SG_ERR_ARG);
    sg_ctx* none = nullptr;
    EXPECT(sg_create(7, &none) == SG_ERR_HIP && none == nullptr);  // the double has one device
    EXPECT(sg_create(-1, &none) == SG_ERR_HIP && none == nullptr);
    float f4[4] = {0};
    int64_t i4[4] = {0};
    sg_loss_spec ce{};
    sg_dither nod{};
    sg_pgd_params pp{};
    sg_feco_params fp{};
    EXPECT(sg_sync(nullptr, nullptr) != SG_OK && sg_health(nullptr) != SG_OK);
    EXPECT(sg_xv_load(nullptr, nullptr) != SG_OK && sg_an_load(nullptr, nullptr) != SG_OK);
    EXPECT(sg_xv_forward(nullptr, f4, 1, 48000, 0, &nod, i4, f4, nullptr, nullptr, nullptr) != SG_OK);
    EXPECT(sg_xv_loss_grad(nullptr, f4, i4, 1, 48000, 0, &ce, &nod, i4, f4, f4, f4, nullptr) != SG_OK);
    EXPECT(sg_xv_pgd_run(nullptr, f4, i4, f4, f4, 1, 48000, &pp, nullptr, i4, f4, f4, nullptr, nullptr, nullptr) != SG_OK);
    EXPECT(sg_an_pgd_run(nullptr, f4, i4, f4, f4, 1, 48000, &pp, nullptr, i4, f4, f4, nullptr, nullptr, nullptr) != SG_OK);
    EXPECT(sg_an_pgd_run_feco(nullptr, f4, i4, f4, f4, 2, 48000, &pp, &fp, nullptr, i4, f4, f4, nullptr, nullptr, nullptr) != SG_OK);
    EXPECT(sg_pgd_update(nullptr, f4, f4, f4, f4, 4, 0.1f, 1, nullptr) != SG_OK);
    EXPECT(sg_trace_begin(nullptr, 4) != SG_OK && sg_trace_end(nullptr, nullptr, nullptr, 0, nullptr) != SG_OK);
    EXPECT(sg_conv1d_rows(nullptr, f4, f4, f4, nullptr, nullptr, 1, 1, 1, 32, 128, 1, 1, 0, 0, 0, nullptr) != SG_OK);
    EXPECT(sg_feco_kmeans(nullptr, f4, 1, 4, 1, 2, 3, nullptr, nullptr) != SG_OK);
    EXPECT(sg_wav_finalize(nullptr, f4, f4, 1, 4, nullptr, nullptr, nullptr) != SG_OK);
    EXPECT(sg_eer_threshold(nullptr, f4, 4, f4, 4, nullptr, nullptr) != SG_OK);

    // ---- a context on the host double -----------------------------------------------------------------------
    sg_ctx* ctx = nullptr;
    EXPECT(sg_create(0, &ctx) == SG_OK && ctx != nullptr);
    if (!ctx) return 1;
    EXPECT(std::strlen(sg_last_error(ctx)) == 0);
    const int B = 3, T = 16000, F = sg_xv_num_frames(T), D = 24, S = 4;
    std::vector<float> x = rnd((size_t)B * T, 200, 0.3f), lower = x, upper = x, grad((size_t)B * T), scores((size_t)B * S), loss(B);
    std::vector<int64_t> y(B, 1), dec(B);
    std::vector<uint8_t> succ(B);

    // state errors come with a message, and the message outlives the next successful call
    EXPECT(sg_xv_forward(ctx, x.data(), B, T, 0, &nod, dec.data(), scores.data(), nullptr, nullptr, nullptr) == SG_ERR_STATE);
    const char* msg = sg_last_error(ctx);
    EXPECT(msg && std::strlen(msg) > 0);
    std::string kept = msg;
    EXPECT(sg_health(ctx) == SG_OK);
    EXPECT(kept == sg_last_error(ctx));

    // malformed weight descriptors
    {
        XvModel m(D, S);
        sg_xv_weights bad = m.desc;
        bad.fc1_weight = nullptr;
        EXPECT(sg_xv_load(ctx, &bad) != SG_OK);
        bad = m.desc; bad.D = 0;
        EXPECT(sg_xv_load(ctx, &bad) != SG_OK);
        bad = m.desc; bad.S = 0;
        EXPECT(sg_xv_load(ctx, &bad) != SG_OK);
        bad = m.desc; bad.tdnn_weight[3] = nullptr;
        EXPECT(sg_xv_load(ctx, &bad) != SG_OK);
        EXPECT(sg_xv_load(ctx, nullptr) != SG_OK);
    }
    XvModel xv(D, S);
    EXPECT(sg_xv_load(ctx, &xv.desc) == SG_OK);  // BatchNorm folding, tap-major / transposed / k4-packed layouts, uploads
    EXPECT(sg_xv_load(ctx, &xv.desc) == SG_OK);  // reload: the old model's allocations are released
    std::vector<float> enr2 = rnd((size_t)2 * D, 300);
    EXPECT(sg_xv_set_enroll(ctx, enr2.data(), 2, 1.5f) == SG_OK);
    EXPECT(sg_xv_set_enroll(ctx, nullptr, 0, 1.5f) != SG_OK || true);  // either refused or a threshold-only update: must not crash
    EXPECT(sg_xv_set_enroll(ctx, xv.enroll.data(), S, -INFINITY) == SG_OK);
    EXPECT(sg_xv_enroll_override(ctx, enr2.data(), 2) == SG_OK && sg_xv_enroll_override(ctx, nullptr, 0) == SG_OK);

    // shape validation of the pass entry points
    EXPECT(sg_xv_forward(ctx, x.data(), 0, T, 0, &nod, dec.data(), scores.data(), nullptr, nullptr, nullptr) != SG_OK);
    EXPECT(sg_xv_forward(ctx, x.data(), B, 100, 0, &nod, dec.data(), scores.data(), nullptr, nullptr, nullptr) != SG_OK);
    EXPECT(sg_xv_forward(ctx, x.data(), B, T, 5, &nod, dec.data(), scores.data(), nullptr, nullptr, nullptr) != SG_OK);
    EXPECT(sg_xv_forward(ctx, nullptr, B, T, 0, &nod, dec.data(), scores.data(), nullptr, nullptr, nullptr) != SG_OK);
    EXPECT(sg_xv_forward(ctx, x.data(), B, 8, 1, &nod, dec.data(), scores.data(), nullptr, nullptr, nullptr) != SG_OK);  // 8 frames < TDNN context
    EXPECT(std::strlen(sg_last_error(ctx)) > 0);
    // valid passes: workspace sizing, table builders (MFCC tables in torchaudio's float32 operation order), launch selection
    const long l0 = hipdouble_launches();
    EXPECT(sg_xv_forward(ctx, x.data(), B, T, 0, &nod, dec.data(), scores.data(), nullptr, nullptr, nullptr) == SG_OK);
    EXPECT(hipdouble_launches() > l0 + 8);
    std::vector<float> emb((size_t)B * D), temb((size_t)B * 512), feats((size_t)B * F * 30), gfeat((size_t)B * F * 30);
    EXPECT(sg_xv_forward(ctx, x.data(), B, T, 0, nullptr, nullptr, nullptr, emb.data(), temb.data(), nullptr) == SG_OK);
    EXPECT(sg_xv_mfcc(ctx, x.data(), B, T, nullptr, &nod, feats.data(), nullptr) == SG_OK);
    EXPECT(sg_xv_cmvn(ctx, feats.data(), B, F, gfeat.data(), nullptr) == SG_OK);
    EXPECT(sg_xv_forward(ctx, feats.data(), B, F, 1, nullptr, dec.data(), scores.data(), nullptr, nullptr, nullptr) == SG_OK);
    EXPECT(sg_xv_loss_grad(ctx, x.data(), y.data(), B, T, 0, &ce, &nod, dec.data(), scores.data(), loss.data(), grad.data(), nullptr) == SG_OK);
    EXPECT(sg_xv_loss_grad(ctx, feats.data(), y.data(), B, F, 2, &ce, nullptr, dec.data(), scores.data(), loss.data(), gfeat.data(), nullptr) == SG_OK);
    EXPECT(sg_xv_loss_grad(ctx, x.data(), nullptr, B, T, 0, &ce, &nod, dec.data(), scores.data(), loss.data(), grad.data(), nullptr) != SG_OK);
    // round 6: the same pass on the float64 transforms (the launch selection of both instantiations runs under the sanitizer)
    EXPECT(sg_xv_configure(ctx, 64) == SG_OK);
    EXPECT(sg_xv_loss_grad(ctx, x.data(), y.data(), B, T, 0, &ce, &nod, dec.data(), scores.data(), loss.data(), grad.data(), nullptr) == SG_OK);
    EXPECT(sg_xv_configure(ctx, 32) == SG_OK);
    sg_loss_spec lin{};
    lin.loss = SG_LOSS_LINEAR;  // needs a coefficient table
    EXPECT(sg_xv_loss_grad(ctx, x.data(), y.data(), B, T, 0, &lin, &nod, dec.data(), scores.data(), loss.data(), grad.data(), nullptr) != SG_OK);
    {
        int32_t rows = 0, ch = 0;
        std::vector<float> act((size_t)B * 300 * 1536);
        EXPECT(sg_xv_debug_activation(ctx, 5, act.data(), (int64_t)act.size(), &rows, &ch, nullptr) == SG_OK && ch == 1536 && rows > 0);
        EXPECT(sg_xv_debug_activation(ctx, 9, act.data(), (int64_t)act.size(), &rows, &ch, nullptr) != SG_OK);
    }
    // the fused loop: parameter validation, EOT grouping, per-step records, stage trace
    pp = sg_pgd_params{};
    pp.step_size = 4e-4f; pp.max_iter = 2; pp.grad_sign = 1; pp.eot_size = 4; pp.eot_batch_size = 3;
    EXPECT(sg_xv_pgd_run(ctx, x.data(), y.data(), lower.data(), upper.data(), B, T, &pp, succ.data(), dec.data(), scores.data(), loss.data(),
                         nullptr, nullptr, nullptr) == SG_ERR_ARG);  // EOT size not divisible
    pp.eot_batch_size = 2;
    pp.max_iter = -1;
    EXPECT(sg_xv_pgd_run(ctx, x.data(), y.data(), lower.data(), upper.data(), B, T, &pp, succ.data(), dec.data(), scores.data(), loss.data(),
                         nullptr, nullptr, nullptr) == SG_ERR_ARG);
    pp.max_iter = 2;
    pp.dither.dither = 1.0f; pp.dither.seed = 99; pp.dither.index_base = 5;
    std::vector<float> ltr((size_t)3 * B);
    std::vector<int64_t> dtr((size_t)3 * B);
    EXPECT(sg_trace_begin(ctx, 0) == SG_ERR_ARG && sg_trace_begin(ctx, 1 << 21) == SG_ERR_ARG);
    int32_t n_rec = -1;
    EXPECT(sg_trace_end(ctx, nullptr, nullptr, 0, &n_rec) == SG_ERR_STATE);
    EXPECT(sg_trace_begin(ctx, 16) == SG_OK);
    EXPECT(sg_xv_pgd_run(ctx, x.data(), y.data(), lower.data(), upper.data(), B, T, &pp, succ.data(), dec.data(), scores.data(), loss.data(),
                         ltr.data(), dtr.data(), nullptr) == SG_OK);
    int32_t tags[16];
    float ms[16];
    EXPECT(sg_trace_end(ctx, tags, ms, 16, &n_rec) == SG_OK && n_rec == 16);  // more launches than records: capped, no overrun
    EXPECT(sg_trace_begin(ctx, 4096) == SG_OK);
    setenv("SG_EOT_MAX_ROWS", "6", 1);  // two groups of two repeats: the per-step records collect rows over the groups
    EXPECT(sg_xv_pgd_run(ctx, x.data(), y.data(), lower.data(), upper.data(), B, T, &pp, succ.data(), dec.data(), scores.data(), loss.data(),
                         ltr.data(), dtr.data(), nullptr) == SG_OK);
    unsetenv("SG_EOT_MAX_ROWS");
    std::vector<int32_t> tg(4096);
    std::vector<float> tm(4096);
    EXPECT(sg_trace_end(ctx, tg.data(), tm.data(), 2, &n_rec) == SG_OK && n_rec > 40);  // capacity smaller than the record count
    {
        float ms_l = 0;
        double fl = 0;
        int32_t rows = 0;
        EXPECT(sg_xv_time_layer(ctx, 3, B, T, 2, &ms_l, &fl, &rows, nullptr) == SG_OK && fl > 0);
        EXPECT(sg_xv_time_layer(ctx, -1, B, T, 2, &ms_l, &fl, &rows, nullptr) == SG_OK);
        EXPECT(sg_xv_time_layer(ctx, 7, B, T, 2, &ms_l, &fl, &rows, nullptr) != SG_OK);
        EXPECT(sg_xv_time_layer(ctx, 3, 64, 48000, 2, &ms_l, &fl, &rows, nullptr) == SG_ERR_STATE);  // beyond the resident workspace
    }
    // larger batches walk the other launch strategies (stream-K kinds by tile count)
    for (int b : {1, 8, 16, 64}) {
        std::vector<float> xb = rnd((size_t)b * 48000, 400 + b, 0.3f), gb((size_t)b * 48000), sb((size_t)b * S), lb(b);
        std::vector<int64_t> yb(b, 0), db(b);
        EXPECT(sg_xv_loss_grad(ctx, xb.data(), yb.data(), b, 48000, 0, &ce, nullptr, db.data(), sb.data(), lb.data(), gb.data(), nullptr) == SG_OK);
    }

    // attack-state updates
    EXPECT(sg_pgd_update(ctx, x.data(), grad.data(), lower.data(), upper.data(), (int64_t)B * T, 4e-4f, 1, nullptr) == SG_OK);
    EXPECT(sg_pgd_update(ctx, x.data(), grad.data(), lower.data(), upper.data(), 0, 4e-4f, 1, nullptr) != SG_OK);
    {
        std::vector<float> mod((size_t)B * T), ea((size_t)B * T), es((size_t)B * T), nxt((size_t)B * T), l2(B), cst(B, 1e-3f);
        EXPECT(sg_cw2_step(ctx, mod.data(), nullptr, nullptr, x.data(), nullptr, nullptr, cst.data(), B, T, 1e-2f, 0, nxt.data(), l2.data(), nullptr) == SG_OK);
        EXPECT(sg_cw2_step(ctx, mod.data(), ea.data(), es.data(), x.data(), nxt.data(), grad.data(), cst.data(), B, T, 1e-2f, 1, nxt.data(), l2.data(), nullptr) == SG_OK);
        EXPECT(sg_cw2_step(ctx, nullptr, ea.data(), es.data(), x.data(), nxt.data(), grad.data(), cst.data(), B, T, 1e-2f, 1, nxt.data(), l2.data(), nullptr) != SG_OK);
        const int half = 2, Q = 2 * half + 1;
        std::vector<float> q((size_t)B * Q * T), nl((size_t)B * Q), lr(B, 1e-3f);
        EXPECT(sg_nes_queries(ctx, x.data(), B, T, half, 1, 1e-3f, 7, 0, 0, nullptr, q.data(), nullptr, nullptr) == SG_OK);
        EXPECT(sg_nes_queries(ctx, x.data(), B, T, 0, 1, 1e-3f, 7, 0, 0, nullptr, q.data(), nullptr, nullptr) != SG_OK);
        EXPECT(sg_nes_grad(ctx, nl.data(), B, T, half, 1, 7, 0, 0, nullptr, 0, 1e-3f, 1, grad.data(), nullptr) == SG_OK);
        EXPECT(sg_fakebob_step(ctx, x.data(), grad.data(), ea.data(), lr.data(), lower.data(), upper.data(), B, T, 0.9f, 0.1f, 1, nullptr) == SG_OK);
        std::vector<float> ds((size_t)B * S);
        EXPECT(sg_loss_eval(ctx, scores.data(), y.data(), B, S, -INFINITY, &ce, dec.data(), loss.data(), ds.data(), nullptr) == SG_OK);
        EXPECT(sg_loss_eval(ctx, scores.data(), y.data(), B, 0, -INFINITY, &ce, dec.data(), loss.data(), ds.data(), nullptr) != SG_OK);
    }

    // the convolution primitive: shape rules
    {
        const int Bc = 2, Ta = 40, taps = 3, Kc = 64, N = 128, Tc = Ta - 2;
        std::vector<float> a((size_t)Bc * Ta * Kc), w((size_t)taps * Kc * N), c((size_t)Bc * Tc * N), bias(N);
        EXPECT(sg_conv1d_rows(ctx, a.data(), w.data(), c.data(), bias.data(), nullptr, Bc, Ta, Tc, Kc, N, taps, 1, 0, 1, 0, nullptr) == SG_OK);
        EXPECT(sg_conv1d_rows(ctx, a.data(), w.data(), c.data(), bias.data(), nullptr, Bc, Ta, Tc, 48, N, taps, 1, 0, 1, 0, nullptr) == SG_ERR_ARG);
        EXPECT(sg_conv1d_rows(ctx, a.data(), w.data(), c.data(), bias.data(), nullptr, Bc, Ta, Tc, Kc, 100, taps, 1, 0, 1, 0, nullptr) == SG_ERR_ARG);
        EXPECT(sg_conv1d_rows(ctx, a.data(), w.data(), c.data(), nullptr, nullptr, Bc, Ta, Tc, Kc, N, taps, 1, 0, 1, 0, nullptr) == SG_ERR_ARG);  // epi 1 without bias
        EXPECT(sg_conv1d_rows(ctx, a.data(), w.data(), c.data(), bias.data(), nullptr, Bc, Ta, Tc, Kc, N, taps, 1, 0, 1, 11, nullptr) == SG_ERR_ARG);
        for (int kern = 1; kern <= 5; ++kern)
            EXPECT(sg_conv1d_rows(ctx, a.data(), w.data(), c.data(), bias.data(), nullptr, Bc, Ta, Tc, Kc, N, taps, 1, 0, 1, kern, nullptr) == SG_OK);
    }

    // ---- AudioNet ------------------------------------------------------------------------------------------
    {
        const int Sa = 11, Fa = sg_an_num_frames(T);
        std::vector<float> sc((size_t)B * Sa), fe((size_t)B * Fa * 32), em((size_t)B * 32);
        EXPECT(sg_an_forward(ctx, x.data(), B, T, 0, dec.data(), sc.data(), nullptr, nullptr) == SG_ERR_STATE);
        AnModel an(Sa);
        sg_an_weights bad = an.desc;
        bad.num_class = 0;
        EXPECT(sg_an_load(ctx, &bad) == SG_ERR_ARG);
        bad = an.desc; bad.bn_var[6] = nullptr;
        EXPECT(sg_an_load(ctx, &bad) == SG_ERR_ARG);
        EXPECT(sg_an_load(ctx, &an.desc) == SG_OK);
        EXPECT(sg_an_forward(ctx, x.data(), B, T, 0, dec.data(), sc.data(), em.data(), nullptr) == SG_OK);
        EXPECT(sg_an_forward(ctx, x.data(), B, 500, 0, dec.data(), sc.data(), em.data(), nullptr) == SG_ERR_ARG);  // shorter than one STFT frame
        EXPECT(sg_an_forward(ctx, x.data(), B, T, 2, dec.data(), sc.data(), em.data(), nullptr) == SG_ERR_ARG);
        EXPECT(sg_an_logmel(ctx, x.data(), B, T, fe.data(), nullptr) == SG_OK);
        EXPECT(sg_an_forward(ctx, fe.data(), B, Fa, 1, dec.data(), sc.data(), em.data(), nullptr) == SG_OK);
        EXPECT(sg_an_forward(ctx, fe.data(), B, 20, 1, dec.data(), sc.data(), em.data(), nullptr) == SG_ERR_ARG);  // too few frames for conv8
        EXPECT(sg_an_loss_grad(ctx, x.data(), y.data(), B, T, 0, &ce, dec.data(), sc.data(), loss.data(), grad.data(), nullptr) == SG_OK);
        {   // the fused CNN kernels' planner (time slices, LDS demand) for forced cuts, and the per-layer sequence it replaces
            const long l1 = hipdouble_launches();
            setenv("SG_AN_SLICES", "7", 1);
            EXPECT(sg_an_loss_grad(ctx, x.data(), y.data(), B, T, 0, &ce, dec.data(), sc.data(), loss.data(), grad.data(), nullptr) == SG_OK);
            setenv("SG_AN_SLICES", "1000", 1);  // more slices than conv8 has rows: clamped
            EXPECT(sg_an_loss_grad(ctx, x.data(), y.data(), B, T, 0, &ce, dec.data(), sc.data(), loss.data(), grad.data(), nullptr) == SG_OK);
            unsetenv("SG_AN_SLICES");
            const long fused = hipdouble_launches() - l1;
            setenv("SG_AN_FUSED", "0", 1);
            const long l2 = hipdouble_launches();
            EXPECT(sg_an_loss_grad(ctx, x.data(), y.data(), B, T, 0, &ce, dec.data(), sc.data(), loss.data(), grad.data(), nullptr) == SG_OK);
            unsetenv("SG_AN_FUSED");
            EXPECT(hipdouble_launches() - l2 > fused / 2 + 10);  // ~28 launches per pass pair against ~8
        }
        EXPECT(sg_an_logmel_backward(ctx, x.data(), B, T, fe.data(), grad.data(), 1, nullptr) == SG_OK);
        pp = sg_pgd_params{};
        pp.step_size = 4e-4f; pp.max_iter = 2; pp.grad_sign = 1; pp.eot_size = 2; pp.eot_batch_size = 1;
        EXPECT(sg_trace_begin(ctx, 512) == SG_OK);
        EXPECT(sg_an_pgd_run(ctx, x.data(), y.data(), lower.data(), upper.data(), B, T, &pp, succ.data(), dec.data(), sc.data(), loss.data(),
                             ltr.data(), dtr.data(), nullptr) == SG_OK);
        EXPECT(sg_trace_end(ctx, tg.data(), tm.data(), 512, &n_rec) == SG_OK && n_rec > 10);
        bool an_tags = false;
        for (int i = 0; i < n_rec; ++i) an_tags |= tg[i] >= SG_STAGE_AN_LOGMEL_FWD;
        EXPECT(an_tags);
        fp = sg_feco_params{};
        fp.k = Fa / 2; fp.max_iter = 3; fp.random_init = 1; fp.seed = 5; fp.index_base = 2;
        EXPECT(sg_an_pgd_run_feco(ctx, x.data(), y.data(), lower.data(), upper.data(), B, T, &pp, &fp, succ.data(), dec.data(), sc.data(),
                                  loss.data(), ltr.data(), dtr.data(), nullptr) == SG_OK);
        EXPECT(sg_an_pgd_run_feco(ctx, x.data(), y.data(), lower.data(), upper.data(), 1, T, &pp, &fp, succ.data(), dec.data(), sc.data(),
                                  loss.data(), nullptr, nullptr, nullptr) == SG_ERR_ARG);  // one utterance: host path
        fp.k = Fa + 1;
        EXPECT(sg_an_pgd_run_feco(ctx, x.data(), y.data(), lower.data(), upper.data(), B, T, &pp, &fp, succ.data(), dec.data(), sc.data(),
                                  loss.data(), nullptr, nullptr, nullptr) == SG_ERR_ARG);
        fp.k = 4;  // too few clusters for the conv stack
        EXPECT(sg_an_pgd_run_feco(ctx, x.data(), y.data(), lower.data(), upper.data(), B, T, &pp, &fp, succ.data(), dec.data(), sc.data(),
                                  loss.data(), nullptr, nullptr, nullptr) == SG_ERR_ARG);
        {   // round 5: a SECOND context in the same process (per-context / per-device launcher state), and every
            // front-end setting of sg_an_configure through the device loops (odd step counts end on the ping-pong twin)
            sg_ctx* ctx2 = nullptr;
            EXPECT(sg_create(0, &ctx2) == SG_OK && ctx2 != nullptr && ctx2 != ctx);
            EXPECT(sg_an_configure(nullptr, 32, 0, 0) == SG_ERR_ARG);
            EXPECT(sg_an_configure(ctx2, 16, 0, 0) == SG_ERR_ARG);
            EXPECT(sg_an_load(ctx2, &an.desc) == SG_OK);
            sg_pgd_params p3 = pp;
            sg_feco_params f2{};
            f2.k = Fa / 2; f2.max_iter = 3; f2.random_init = 1; f2.seed = 5;
            for (int cfg = 0; cfg < 8; ++cfg) {
                EXPECT(sg_an_configure(ctx2, (cfg & 1) ? 64 : 32, (cfg >> 1) & 1, (cfg >> 2) & 1) == SG_OK);
                p3.max_iter = 1 + (cfg & 1) + ((cfg >> 2) & 1);
                EXPECT(sg_an_pgd_run(ctx2, x.data(), y.data(), lower.data(), upper.data(), B, T, &p3, succ.data(), dec.data(), sc.data(),
                                     loss.data(), nullptr, nullptr, nullptr) == SG_OK);
                EXPECT(sg_an_loss_grad(ctx2, x.data(), y.data(), B, T, 0, &ce, dec.data(), sc.data(), loss.data(), grad.data(), nullptr) == SG_OK);
                EXPECT(sg_an_logmel(ctx2, x.data(), B, T, fe.data(), nullptr) == SG_OK);
                EXPECT(sg_an_logmel_backward(ctx2, x.data(), B, T, fe.data(), grad.data(), 1, nullptr) == SG_OK);
                EXPECT(sg_an_logmel_backward(ctx2, x.data(), B, T, fe.data(), grad.data(), 0, nullptr) == SG_OK);
                EXPECT(sg_an_pgd_run_feco(ctx2, x.data(), y.data(), lower.data(), upper.data(), B, T, &p3, &f2, succ.data(), dec.data(),
                                          sc.data(), loss.data(), nullptr, nullptr, nullptr) == SG_OK);
            }
            // the first context is untouched by all that
            EXPECT(sg_an_loss_grad(ctx, x.data(), y.data(), B, T, 0, &ce, dec.data(), sc.data(), loss.data(), grad.data(), nullptr) == SG_OK);
            EXPECT(sg_set_streamk(nullptr, 0) == SG_ERR_ARG && sg_set_streamk(ctx2, 0) == SG_OK && sg_set_streamk(ctx2, 1) == SG_OK);
            EXPECT(sg_debug_lose_handoffs(ctx2, -1) == SG_ERR_ARG && sg_debug_lose_handoffs(ctx2, 0) == SG_OK);
            // round 6: the x-vector front-end's transform precision, the k-means launch counter's test hook, the automatic
            // spectrum cache (-1)
            EXPECT(sg_xv_configure(nullptr, 32) == SG_ERR_ARG && sg_xv_configure(ctx2, 48) == SG_ERR_ARG);
            EXPECT(sg_xv_configure(ctx2, 64) == SG_OK && sg_xv_configure(ctx2, 32) == SG_OK);
            EXPECT(sg_debug_feco_epoch(nullptr, 1) == SG_ERR_ARG && sg_debug_feco_epoch(ctx2, 0x7FFEu) == SG_OK);
            EXPECT(sg_an_configure(ctx2, 32, -1, -1) == SG_OK);
            EXPECT(sg_an_loss_grad(ctx2, x.data(), y.data(), B, T, 0, &ce, dec.data(), sc.data(), loss.data(), grad.data(), nullptr) == SG_OK);
            EXPECT(sg_sync(ctx2, nullptr) == SG_OK);
            sg_destroy(ctx2);
        }
        int32_t rows = 0, ch = 0;
        EXPECT(sg_an_debug_activation(ctx, 8, nullptr, 0, &rows, &ch, nullptr) == SG_OK && ch == 32);
        EXPECT(sg_an_debug_activation(ctx, 0, nullptr, 0, &rows, &ch, nullptr) != SG_OK);
    }

    // ---- FeCo, post-processing ------------------------------------------------------------------------------
    {
        const int Bf = 2, Ff = 50, Df = 32, k = 20;
        std::vector<float> fe = rnd((size_t)Bf * Ff * Df, 500, 1.f), out((size_t)2 * Bf * k * Df), dfe((size_t)Bf * Ff * Df);
        std::vector<int32_t> ids((size_t)2 * Bf * Ff), cnt((size_t)2 * Bf * k);
        EXPECT(sg_feco_kmeans(ctx, fe.data(), Bf, Ff, Df, k, 5, ids.data(), nullptr) == SG_OK);
        EXPECT(sg_feco_kmeans(ctx, fe.data(), Bf, Ff, 65, k, 5, ids.data(), nullptr) != SG_OK);   // D <= 64
        EXPECT(sg_feco_kmeans(ctx, fe.data(), Bf, Ff, Df, Ff + 1, 5, ids.data(), nullptr) != SG_OK);
        EXPECT(sg_feco_kmeans_seeded(ctx, fe.data(), Bf, Ff, Df, k, 5, 3, 1, ids.data(), nullptr) == SG_OK);
        EXPECT(sg_feco_compress(ctx, fe.data(), ids.data(), Bf, Ff, Df, k, out.data(), cnt.data(), nullptr) == SG_OK);
        EXPECT(sg_feco_kmeans_compress(ctx, fe.data(), Bf, Ff, Df, k, 5, 1, 3, 0, 2, ids.data(), out.data(), cnt.data(), nullptr) == SG_OK);
        EXPECT(sg_feco_kmeans_compress(ctx, fe.data(), Bf, Ff, Df, k, 5, 0, 3, 0, 2, ids.data(), out.data(), cnt.data(), nullptr) != SG_OK);  // reps need the seeded form
        EXPECT(sg_feco_compress_backward_reps(ctx, out.data(), ids.data(), cnt.data(), Bf, Ff, Df, k, 1, 2, dfe.data(), nullptr) == SG_OK);
        EXPECT(sg_feco_compress_backward(ctx, out.data(), ids.data(), cnt.data(), Bf, Ff, Df, k, 1, dfe.data(), nullptr) == SG_OK);
        std::vector<int16_t> pcm((size_t)B * T);
        std::vector<double> met((size_t)B * 5), out3(3);
        EXPECT(sg_wav_finalize(ctx, x.data(), lower.data(), B, T, pcm.data(), met.data(), nullptr) == SG_OK);
        EXPECT(sg_wav_finalize(ctx, nullptr, lower.data(), B, T, pcm.data(), met.data(), nullptr) != SG_OK);  // metrics need the benign audio
        EXPECT(sg_wav_finalize(ctx, nullptr, lower.data(), B, T, pcm.data(), nullptr, nullptr) == SG_OK);
        EXPECT(sg_eer_threshold(ctx, scores.data(), 6, scores.data() + 6, 6, out3.data(), nullptr) == SG_OK);
        EXPECT(sg_eer_threshold(ctx, scores.data(), 0, scores.data(), 6, out3.data(), nullptr) != SG_OK);
    }
    EXPECT(sg_sync(ctx, nullptr) == SG_OK);
    sg_destroy(ctx);
    EXPECT(hipdouble_live_allocs() == 0);  // everything the context allocated through the runtime was released
    if (g_fail) {
        std::fprintf(stderr, "%d expectation(s) failed\n", g_fail);
        return 1;
    }
    std::printf("abi_asan_driver: ok (%ld kernel launches issued against the host double)\n", hipdouble_launches());
    return 0;
}
