// C-ABI entry points (include/speakerguard_hip.h): context, model load (BatchNorm folding and
// weight re-layout), workspace, and the kernel sequences of one forward / backward / PGD pass.
#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <type_traits>

#include "sg_internal.h"

using namespace sg;

namespace {

int fail(sg_ctx* ctx, int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    if (ctx) ctx->err = buf;
    return code;
}

#define SG_HIP(expr)                                                                                   \
    do {                                                                                               \
        hipError_t e_ = (expr);                                                                        \
        if (e_ != hipSuccess) return fail(ctx, SG_ERR_HIP, "%s failed: %s (%s:%d)", #expr,             \
                                          hipGetErrorString(e_), __FILE__, __LINE__);                  \
    } while (0)

#define SG_STAGE(tag, expr)          \
    do {                             \
        trace_mark(ctx, (tag), s, 0); \
        SG_HIP(expr);                \
        trace_mark(ctx, (tag), s, 1); \
    } while (0)

template <typename T>
int dev_alloc(sg_ctx* ctx, std::vector<void*>& pool, T** out, size_t count) {
    void* p = nullptr;
    SG_HIP(hipMalloc(&p, count * sizeof(T) + 256));
    pool.push_back(p);
    *out = reinterpret_cast<T*>(p);
    return SG_OK;
}

template <typename T>
int dev_upload(sg_ctx* ctx, std::vector<void*>& pool, T** out, const std::vector<T>& host) {
    int rc = dev_alloc(ctx, pool, out, host.size());
    if (rc) return rc;
    SG_HIP(hipMemcpy(*out, host.data(), host.size() * sizeof(T), hipMemcpyHostToDevice));
    return SG_OK;
}

// [K][N] row-major -> k4-major [K/4][N][4] (Wq[k/4][n][k%4]): four consecutive k of one column become one
// 16-byte LDS read in the quad-fed GEMM (k_conv_gemm.hip); K is padded up to a multiple of 4 with zeros.
std::vector<float> pack_k4(const std::vector<float>& w, int K, int N) {
    const int K4 = (K + 3) / 4;
    std::vector<float> q((size_t)K4 * N * 4, 0.f);
    for (int k = 0; k < K; ++k)
        for (int n = 0; n < N; ++n) q[((size_t)(k / 4) * N + n) * 4 + (k & 3)] = w[(size_t)k * N + n];
    return q;
}

void free_pool(std::vector<void*>& pool) {
    for (void* p : pool) (void)hipFree(p);
    pool.clear();
}

// ------------------------------------------------------------------------------------------
// MFCC constant tables (torchaudio kaldi.py v0.6.0: _feature_window_function, get_mel_banks,
// _get_dct_matrix, _get_lifter_coeffs).  torchaudio builds them with float32 tensor arithmetic,
// and the float32 rounding of the ARGUMENTS (e.g. pi/30*(n+0.5)*k up to 94 rad) moves the DCT /
// lifter / mel entries by up to 5e-6 -- 3e-4 on a cepstrum, enough to flip ReLU masks downstream.
// So the tables are computed here in float32 with the same operation order, not in double.
int build_tables(sg_ctx* ctx) {
    if (ctx->tables_ready) return SG_OK;
    const double PI = 3.14159265358979323846;
    std::vector<float> window(kWin), melw(kMel * 256, 0.f), dct(kMel * kCep), lifter(kCep), w0(256, 0.f), w1(256, 0.f);
    std::vector<int> lo(kMel), hi(kMel), m0(256, -1);
    std::vector<double2> tw(256);
    {   // torch.hann_window(400, periodic=False): arange * (2 pi / 399) -> cos -> * -0.5 + 0.5; then pow 0.85
        const float step = (float)(PI * 2.0 / (double)(kWin - 1));
        for (int n = 0; n < kWin; ++n) {
            const float c = cosf((float)n * step);
            const float h = c * -0.5f + 0.5f;
            window[n] = powf(h, 0.85f);
        }
    }
    // get_mel_banks: python-double scalars, float32 tensors
    auto mel_scalar = [](double f) { return 1127.0 * std::log(1.0 + f / 700.0); };
    const double mel_low = mel_scalar(20.0), mel_high = mel_scalar(7600.0);
    const double delta = (mel_high - mel_low) / (kMel + 1);
    const float bin_width = (float)(16000.0 / kFft);
    for (int m = 0; m < kMel; ++m) {
        const float fm = (float)m;
        const float left = (float)mel_low + fm * (float)delta;
        const float center = (float)mel_low + (fm + 1.0f) * (float)delta;
        const float right = (float)mel_low + (fm + 2.0f) * (float)delta;
        lo[m] = 256;
        hi[m] = 0;
        for (int k = 0; k < 256; ++k) {
            const float fr = bin_width * (float)k;
            const float mk = 1127.0f * logf(1.0f + fr / 700.0f);
            const float up = (mk - left) / (center - left), down = (right - mk) / (right - center);
            const float w = fmaxf(0.f, fminf(up, down));
            melw[m * 256 + k] = w;
            if (w > 0.f) {
                if (k < lo[m]) lo[m] = k;
                hi[m] = k + 1;
            }
        }
        if (lo[m] > hi[m]) lo[m] = hi[m] = 0;
        // each of the two lanes of a filter keeps the weights of its half in registers (k_mfcc.hip: kMelLaneBins)
        if ((hi[m] - lo[m] + 1) / 2 > kMelLaneBins) return fail(ctx, SG_ERR_STATE, "mel filter %d spans %d bins: more than the MFCC kernel holds", m, hi[m] - lo[m]);
    }
    for (int k = 0; k < 256; ++k) {
        int first = -1, cnt = 0;
        for (int m = 0; m < kMel; ++m)
            if (melw[m * 256 + k] > 0.f) {
                if (first < 0) first = m;
                ++cnt;
            }
        if (cnt > 2 || (cnt == 2 && melw[(first + 1) * 256 + k] <= 0.f))
            return fail(ctx, SG_ERR_STATE, "mel filterbank is not a two-overlap triangular bank");
        m0[k] = first;
        if (first >= 0) {
            w0[k] = melw[first * 256 + k];
            w1[k] = first + 1 < kMel ? melw[(first + 1) * 256 + k] : 0.f;
        }
    }
    {   // _get_dct_matrix: cos(pi/N * (n + 0.5) * k), row 0 * 1/sqrt(2), all * sqrt(2/N), transposed,
        // first column overwritten by sqrt(1/N)
        const float a = (float)(PI / (double)kMel);
        for (int m = 0; m < kMel; ++m)
            for (int c = 0; c < kCep; ++c) {
                float v = cosf((a * ((float)m + 0.5f)) * (float)c);
                if (c == 0) v *= (float)(1.0 / std::sqrt(2.0));
                v *= (float)std::sqrt(2.0 / (double)kMel);
                dct[m * kCep + c] = c == 0 ? (float)std::sqrt(1.0 / (double)kMel) : v;
            }
        // _get_lifter_coeffs: 1 + 0.5 * Q * sin(pi * i / Q)
        for (int c = 0; c < kCep; ++c) lifter[c] = 1.0f + 11.0f * sinf(((float)PI * (float)c) / 22.0f);
    }
    for (int k = 0; k < 256; ++k) tw[k] = make_double2(std::cos(2.0 * PI * k / kFft), -std::sin(2.0 * PI * k / kFft));
    MfccTables& t = ctx->tab;
    // the kernels' LDS table images (MfccLdsImage), one per transform precision
    auto fill_image = [&](auto& img) {
        using R = std::remove_reference_t<decltype(img.tw1[0])>;
        std::memset(&img, 0, sizeof(img));
        auto w512 = [&](int m, R* out) {  // W512^m from the half circle (W^(m + 256) = -W^m), rounded once to R
            const double2 w = tw[m & 255];
            const double f = (m & 256) ? -1.0 : 1.0;
            out[0] = (R)(f * w.x);
            out[1] = (R)(f * w.y);
        };
        for (int i = 0; i < kMfccTw1; ++i) w512((i / 64 + 1) * (i % 64), &img.tw1[2 * i]);
        for (int i = 0; i < kMfccTw2; ++i) {
            const int b = i / 9, c = i - 9 * b;
            if (c < 8) w512(8 * b * c, &img.tw2[2 * i]);
        }
        for (int n = 0; n < kWin; ++n) img.window[n] = window[n];
        for (int i = 0; i < kMel * kCep; ++i) img.dct[i] = dct[i];
        for (int c = 0; c < kCep; ++c)
            for (int m = 0; m < kMel; ++m) img.dct_t[c * 32 + m] = dct[m * kCep + c];
        for (int c = 0; c < kCep; ++c) img.lifter[c] = lifter[c];
        // the weights of the bins a lane's half of a mel filter sums (two lanes per filter), ascending bin order, zero-padded
        for (int lane = 0; lane < 64; ++lane) {
            const int m = lane >> 1, h = lane & 1;
            if (m >= kMel) continue;
            const int mid = lo[m] + (hi[m] - lo[m] + 1) / 2;
            const int k0 = h ? mid : lo[m], cnt = (h ? hi[m] : mid) - k0;
            img.mel_k0[lane] = k0;
            for (int j = 0; j < kMelLaneBins && j < cnt; ++j) {
                const int k = std::min(k0 + j, 255);
                img.melw_lane[j * 64 + lane] = m0[k] == m ? w0[k] : w1[k];
            }
        }
    };
    auto img32 = std::make_unique<MfccLdsImage<float>>();
    auto img64 = std::make_unique<MfccLdsImage<double>>();
    fill_image(*img32);
    fill_image(*img64);
    int rc = 0;
    rc |= dev_upload(ctx, ctx->model_allocs, &t.bin_m0, m0);
    rc |= dev_upload(ctx, ctx->model_allocs, &t.bin_w0, w0);
    rc |= dev_upload(ctx, ctx->model_allocs, &t.bin_w1, w1);
    {
        std::vector<MfccLdsImage<float>> v32(1, *img32);
        std::vector<MfccLdsImage<double>> v64(1, *img64);
        MfccLdsImage<float>* d32 = nullptr;
        MfccLdsImage<double>* d64 = nullptr;
        rc |= dev_upload(ctx, ctx->model_allocs, &d32, v32);
        rc |= dev_upload(ctx, ctx->model_allocs, &d64, v64);
        t.lds_f32 = d32;
        t.lds_f64 = d64;
    }
    rc |= dev_alloc(ctx, ctx->model_allocs, &ctx->range_scratch, 512);
    rc |= dev_alloc(ctx, ctx->model_allocs, &ctx->cw2_scratch, 1024 * 32);
    rc |= dev_alloc(ctx, ctx->model_allocs, &ctx->sk_slabs, (size_t)512 * 128 * 128);  // >= 768 * 64 * 128
    rc |= dev_alloc(ctx, ctx->model_allocs, &ctx->sk_flags, 1024);
    if (!rc && hipMemset(ctx->sk_flags, 0, 1024 * sizeof(unsigned)) != hipSuccess) rc = SG_ERR_HIP;
    if (rc) return SG_ERR_HIP;
    ctx->tables_ready = true;
    return SG_OK;
}

// TDNN frames after each layer for F input frames
bool layer_frames(int F, int* Fl) {
    int f = F;
    for (int l = 0; l < kLayers; ++l) {
        f -= (kTaps[l] - 1) * kDil[l];
        Fl[l] = f;
    }
    return f >= 2;  // unbiased std needs two frames
}

int ensure_workspace(sg_ctx* ctx, int B, int T, int F) {
    Workspace& w = ctx->ws;
    if (B <= w.B && F <= w.F && T <= w.T && w.scale) {
        layer_frames(F, w.Fl);
        return SG_OK;
    }
    (void)hipDeviceSynchronize();
    free_pool(w.allocs);
    const int cb = B > w.B ? B : w.B, cf = F > w.F ? F : w.F, ct = T > w.T ? T : w.T;
    w = Workspace();
    w.B = cb; w.F = cf; w.T = ct;
    int Fl[kLayers];
    layer_frames(cf, Fl);
    const size_t b = (size_t)cb;
    int rc = 0;
    rc |= dev_alloc(ctx, w.allocs, &w.scale, 4);
    rc |= dev_alloc(ctx, w.allocs, &w.feats_raw, b * cf * kCep);
    rc |= dev_alloc(ctx, w.allocs, &w.feats, b * cf * kFeatPad);
    for (int l = 0; l < kLayers; ++l) {
        const size_t rows = b * (size_t)(Fl[l] > 0 ? Fl[l] : 1);
        rc |= dev_alloc(ctx, w.allocs, &w.act[l], rows * kCoutPad[l]);
        rc |= dev_alloc(ctx, w.allocs, &w.dact[l], rows * kCoutPad[l]);
    }
    rc |= dev_alloc(ctx, w.allocs, &w.dfeats, (size_t)kL1BwdSplitK * b * cf * kFeatPad);
    rc |= dev_alloc(ctx, w.allocs, &w.dfeats_raw, b * cf * kCep);
    rc |= dev_alloc(ctx, w.allocs, &w.dframes, b * cf * kWin);
    rc |= dev_alloc(ctx, w.allocs, &w.spec_cache, b * cf * 256);
    rc |= dev_alloc(ctx, w.allocs, &w.mel_cache, b * cf * 32);
    rc |= dev_alloc(ctx, w.allocs, &w.stats, b * kStats);
    rc |= dev_alloc(ctx, w.allocs, &w.fc1_part, (size_t)kFc1SplitK * b * kEmb);
    rc |= dev_alloc(ctx, w.allocs, &w.demb, b * kEmb);
    rc |= dev_alloc(ctx, w.allocs, &w.dstats_part, (size_t)kFc1BwdSplitK * b * kStats);
    rc |= dev_alloc(ctx, w.allocs, &w.tdnn_emb, b * kEmb);
    rc |= dev_alloc(ctx, w.allocs, &w.emb, b * 512);
    rc |= dev_alloc(ctx, w.allocs, &w.scores, b * 1024);
    rc |= dev_alloc(ctx, w.allocs, &w.loss, b);
    rc |= dev_alloc(ctx, w.allocs, &w.decisions, b);
    rc |= dev_alloc(ctx, w.allocs, &w.grad, b * (size_t)(ct > 0 ? ct : 1));
    rc |= dev_alloc(ctx, w.allocs, &w.y_rep, b);
    if (rc) {
        free_pool(w.allocs);
        w = Workspace();
        return SG_ERR_HIP;
    }
    layer_frames(F, w.Fl);
    return SG_OK;
}

struct PassDims {
    int B, T, F;              // B: rows of the pass
    bool keep_scale = false;  // reuse ws.scale from the previous pass (fused loop: x stays in [-1, 1])
    int Bu = 0;               // > 0: the rows are B / Bu EOT repeats of Bu utterances (row = repeat * Bu + utterance); the
                              // waveform has Bu rows and d loss / d x sums the repeats (sg_xv_pgd_run)
};

// waveform / features -> padded CMVN features in ws.feats
// want_grad: a backward of this pass follows (the forward leaves its spectra and mel energies for it: 39 MB at 64 utterances,
// not written by forward-only calls -- decisions, NES / FAKEBOB queries, the last pass of an attack)
int run_frontend(sg_ctx* ctx, const float* x, const PassDims& d, int flag, const sg_dither* dz, bool want_grad, hipStream_t s) {
    Workspace& w = ctx->ws;
    if (flag == SG_FLAG_WAV) {
        if (!d.keep_scale) SG_HIP(launch_input_scale(x, (int64_t)(d.Bu > 0 ? d.Bu : d.B) * d.T, ctx->range_scratch, w.scale, 0, s));
        MfccTables tab = ctx->tab;
        tab.spec_cache = want_grad ? w.spec_cache : nullptr;  // the backward of this pass starts from the stored spectrum
        tab.mel_cache = want_grad ? w.mel_cache : nullptr;
        tab.rep_utts = d.Bu;
        SG_STAGE(SG_STAGE_MFCC_FWD, launch_mfcc_fwd(tab, x, d.B, d.T, d.F, w.scale, dz, w.feats_raw, s));
        SG_STAGE(SG_STAGE_CMVN_FWD, launch_cmvn_fwd(w.feats_raw, kCep, w.feats, kFeatPad, d.B, d.F, s));
    } else if (flag == SG_FLAG_RAW) {
        SG_HIP(launch_cmvn_fwd(x, kCep, w.feats, kFeatPad, d.B, d.F, s));
    } else {
        SG_HIP(launch_copy_cols(x, kCep, w.feats, kFeatPad, (int64_t)d.B * d.F, kCep, s));
    }
    return SG_OK;
}

ConvGemmArgs fwd_layer_args(sg_ctx* ctx, int l, int B, int F) {
    const Workspace& w = ctx->ws;
    ConvGemmArgs a{};
    a.A = l == 0 ? w.feats : w.act[l - 1];
    a.W = ctx->xv.wf[l];
    a.Wq = ctx->xv.wfq[l];
    a.C = w.act[l];
    a.bias = ctx->xv.bias[l];
    a.mask = nullptr;
    a.Ta = l == 0 ? F : w.Fl[l - 1];
    a.Tc = w.Fl[l];
    a.M = B * a.Tc;
    a.N = kCoutPad[l];
    a.Kc = kCinPad[l];
    a.lda = kCinPad[l];
    a.ldw = kCoutPad[l];
    a.ldc = kCoutPad[l];
    a.taps = kTaps[l];
    a.tap_step = kDil[l];
    a.total_chunks = a.taps * (a.Kc / 32);
    a.chunks_per_split = a.total_chunks;
    a.split_stride = 0;
    conv_ctx_args(ctx, a);
    return a;
}

int run_tdnn_forward(sg_ctx* ctx, const PassDims& d, hipStream_t s) {
    Workspace& w = ctx->ws;
    for (int l = 0; l < kLayers; ++l) {
        ConvGemmArgs a = fwd_layer_args(ctx, l, d.B, d.F);
        SG_STAGE(l + 1, launch_conv_gemm(a, 0, EPI_BIAS_RELU, 1, s));
    }
    SG_STAGE(SG_STAGE_POOL_FWD, launch_pool_fwd(w.act[4], d.B, w.Fl[4], w.stats, s));
    // fc1 as a split-K contraction: (B x 3072) x (3072 x 512) -> kFc1SplitK partial slabs
    ConvGemmArgs a{};
    a.A = w.stats; a.W = ctx->xv.fc1_w; a.C = w.fc1_part; a.bias = nullptr; a.mask = nullptr;
    a.M = d.B; a.N = kEmb; a.Ta = 1; a.Tc = 1; a.Kc = kStats; a.lda = kStats; a.ldw = kEmb; a.ldc = kEmb;
    a.taps = 1; a.tap_step = 0; a.total_chunks = kStats / 32;
    a.chunks_per_split = a.total_chunks / kFc1SplitK;
    a.split_stride = (long long)d.B * kEmb;
    SG_STAGE(SG_STAGE_FC1_FWD, launch_conv_gemm(a, 2, EPI_NONE, kFc1SplitK, s));
    return SG_OK;
}

// d loss / d fc1-output (ws.demb) -> d loss / d padded CMVN features (ws.dfeats)
int run_tdnn_backward(sg_ctx* ctx, const PassDims& d, hipStream_t s) {
    Workspace& w = ctx->ws;
    {
        ConvGemmArgs a{};
        a.A = w.demb; a.W = ctx->xv.fc1_wt; a.C = w.dstats_part; a.bias = nullptr; a.mask = nullptr;
        a.M = d.B; a.N = kStats; a.Ta = 1; a.Tc = 1; a.Kc = kEmb; a.lda = kEmb; a.ldw = kStats; a.ldc = kStats;
        a.taps = 1; a.tap_step = 0; a.total_chunks = kEmb / 32;
        a.chunks_per_split = a.total_chunks / kFc1BwdSplitK;
        a.split_stride = (long long)d.B * kStats;
        SG_STAGE(SG_STAGE_FC1_BWD, launch_conv_gemm(a, 2, EPI_NONE, kFc1BwdSplitK, s));
    }
    SG_STAGE(SG_STAGE_POOL_BWD, launch_pool_bwd(w.act[4], w.stats, w.dstats_part, kFc1BwdSplitK, d.B, w.Fl[4], w.dact[4], s));
    for (int l = kLayers - 1; l >= 0; --l) {
        ConvGemmArgs a{};
        a.A = w.dact[l];
        a.W = ctx->xv.wb[l];
        a.Wq = ctx->xv.wbq[l];
            a.C = l == 0 ? w.dfeats : w.dact[l - 1];
        a.bias = nullptr;
        a.mask = l == 0 ? nullptr : w.act[l - 1];
        a.Ta = w.Fl[l];
        a.Tc = l == 0 ? d.F : w.Fl[l - 1];
        a.M = d.B * a.Tc;
        a.N = kCinPad[l];
        a.Kc = kCoutPad[l];
        a.lda = kCoutPad[l];
        a.ldw = kCinPad[l];
        a.ldc = kCinPad[l];
        a.taps = kTaps[l];
        a.tap_step = -kDil[l];
        a.total_chunks = a.taps * (a.Kc / 32);
        a.chunks_per_split = a.total_chunks;
        a.split_stride = 0;
        conv_ctx_args(ctx, a);
        int splits = 1, tile = 0;
        if (l == 0) {  // 32 output columns: 148 tiles at B = 64 -- split K per tap to fill the chip
            tile = 1;
            splits = kL1BwdSplitK;
            a.chunks_per_split = a.total_chunks / kL1BwdSplitK;
            a.split_stride = (long long)d.B * d.F * kFeatPad;
        }
        SG_STAGE(-(l + 1), launch_conv_gemm(a, tile, l == 0 ? EPI_NONE : EPI_RELU_MASK, splits, s));
    }
    return SG_OK;
}

int check_dims(sg_ctx* ctx, int B, int TF, int flag, PassDims* d) {
    if (!ctx) return SG_ERR_ARG;
    if (!ctx->xv.loaded) return fail(ctx, SG_ERR_STATE, "no x-vector model loaded (call sg_xv_load)");
    if (int h = sg_health(ctx)) return h;
    SG_HIP(hipSetDevice(ctx->device));
    if (B < 1) return fail(ctx, SG_ERR_ARG, "B must be >= 1");
    if (flag < 0 || flag > 2) return fail(ctx, SG_ERR_ARG, "flag must be 0 (wav), 1 (raw feat) or 2 (cmvn feat)");
    d->B = B;
    if (flag == SG_FLAG_WAV) {
        if (TF < kWin) return fail(ctx, SG_ERR_ARG, "waveform shorter than one 25 ms window");
        d->T = TF;
        d->F = num_frames(TF);
    } else {
        d->T = 0;
        d->F = TF;
    }
    int Fl[kLayers];
    if (!layer_frames(d->F, Fl)) return fail(ctx, SG_ERR_ARG, "%d frames are too few for the TDNN context", d->F);
    // the contraction kernels address an activation tensor through 32-bit buffer offsets (k_conv_gemm.hip): the
    // largest one, the (B, F5, 1536) tdnn5 output, has to stay below 2 GiB -- B <= 1294 at 3 s, far past the
    // throughput plateau (B = 512); callers chunk larger sets (attack.* does, by batch_size)
    {
        size_t worst = 0;
        for (int l = 0; l < kLayers; ++l) {
            const size_t bytes = (size_t)B * Fl[l] * kCoutPad[l] * sizeof(float);
            if (bytes > worst) worst = bytes;
        }
        if (worst >= 0x80000000ull)
            return fail(ctx, SG_ERR_ARG, "batch of %d x %d frames needs a %.1f GiB activation tensor; one pass handles < 2 GiB "
                        "(at most %d utterances of this length): split the batch", B, d->F, (double)worst / (1ull << 30),
                        (int)(0x7FFFFFFFull / (worst / (size_t)B)));
    }
    int rc = build_tables(ctx);
    if (rc) return rc;
    rc = ensure_workspace(ctx, d->B, d->T, d->F);
    if (rc) return fail(ctx, rc, "workspace allocation failed: %s", ctx->err.c_str());
    return SG_OK;
}

// after the tail produced ws.demb: chain back to the caller's input level
int run_backward_to_input(sg_ctx* ctx, const float* x, const PassDims& d, int flag, const sg_dither* dz,
                          float* grad_out, float* x_update, const float* lower, const float* upper, float step,
                          int grad_sign, hipStream_t s, const float* grad_acc_in = nullptr) {
    Workspace& w = ctx->ws;
    int rc = run_tdnn_backward(ctx, d, s);
    if (rc) return rc;
    if (flag == SG_FLAG_CMVN) {
        SG_HIP(launch_sum_cols(w.dfeats, kFeatPad, kL1BwdSplitK, (long long)d.B * d.F * kFeatPad, grad_out, kCep,
                               (int64_t)d.B * d.F, kCep, s));
    } else if (flag == SG_FLAG_RAW) {
        SG_HIP(launch_cmvn_bwd(w.dfeats, kFeatPad, kL1BwdSplitK, (long long)d.B * d.F * kFeatPad, grad_out, kCep, d.B,
                               d.F, s));
    } else {
        SG_STAGE(SG_STAGE_CMVN_BWD, launch_cmvn_bwd(w.dfeats, kFeatPad, kL1BwdSplitK, (long long)d.B * d.F * kFeatPad,
                                                    w.dfeats_raw, kCep, d.B, d.F, s));
        MfccTables tab = ctx->tab;
        static const bool use_cache = [] {
            const char* e = sg_tune_env("SG_MFCC_CACHE");  // 0 = recompute the forward in the backward kernel
            return !e || atoi(e) != 0;
        }();
        if (use_cache) {
            tab.spec_cache = w.spec_cache;
            tab.mel_cache = w.mel_cache;
        }
        tab.rep_utts = d.Bu;
        const int utts = d.Bu > 0 ? d.Bu : d.B;
        SG_STAGE(SG_STAGE_MFCC_BWD, launch_mfcc_bwd(tab, x, d.B, d.T, d.F, w.scale, dz, w.dfeats_raw, w.dframes, s));
        SG_STAGE(SG_STAGE_OVERLAP_ADD, launch_frames_to_wave(w.dframes, utts, d.T, d.F, d.B / utts, grad_acc_in, grad_out, x_update,
                                                             lower, upper, step, grad_sign, s));
    }
    return SG_OK;
}

}  // namespace

// ============================================================================== C-ABI
extern "C" {

int sg_version(void) { return 100; }

int sg_create(int device, sg_ctx** out) {
    if (!out) return SG_ERR_ARG;
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || device < 0 || device >= n) return SG_ERR_HIP;
    if (hipSetDevice(device) != hipSuccess) return SG_ERR_HIP;
    sg_ctx* ctx = new (std::nothrow) sg_ctx();
    if (!ctx) return SG_ERR_HIP;
    ctx->device = device;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess && prop.multiProcessorCount > 0)
        ctx->num_cus = prop.multiProcessorCount;
    if (hipEventCreate(&ctx->ev0) != hipSuccess || hipEventCreate(&ctx->ev1) != hipSuccess) {
        delete ctx;
        return SG_ERR_HIP;
    }
    void* hw = nullptr;
    if (hipHostMalloc(&hw, 64, hipHostMallocMapped | hipHostMallocCoherent) == hipSuccess) {
        ctx->err_host = static_cast<unsigned*>(hw);
        *ctx->err_host = 0;
        void* dp = nullptr;
        if (hipHostGetDevicePointer(&dp, hw, 0) == hipSuccess) ctx->err_dev = static_cast<unsigned*>(dp);
    }
    *out = ctx;
    return SG_OK;
}

void sg_destroy(sg_ctx* ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    (void)hipDeviceSynchronize();
    free_pool(ctx->ws.allocs);
    free_pool(ctx->an_ws.allocs);  // (leaked until round 4: found by the sanitizer build's leak check, `make asan`)
    free_pool(ctx->xv.allocs);
    free_pool(ctx->model_allocs);
    if (ctx->err_host) (void)hipHostFree(ctx->err_host);
    if (ctx->ev0) (void)hipEventDestroy(ctx->ev0);
    if (ctx->ev1) (void)hipEventDestroy(ctx->ev1);
    for (hipEvent_t e : ctx->trace_ev) (void)hipEventDestroy(e);
    delete ctx;
}

const char* sg_last_error(const sg_ctx* ctx) { return ctx ? ctx->err.c_str() : "null context"; }

int sg_sync(sg_ctx* ctx, void* stream) {
    if (!ctx) return SG_ERR_ARG;
    SG_HIP(hipStreamSynchronize((hipStream_t)stream));
    return sg_health(ctx);
}

int sg_health(sg_ctx* ctx) {
    if (!ctx) return SG_ERR_ARG;
    if (ctx->err_host && __atomic_load_n(ctx->err_host, __ATOMIC_RELAXED) != 0) {
        __atomic_store_n(ctx->err_host, 0u, __ATOMIC_RELAXED);
        return fail(ctx, SG_ERR_HIP, "a stream-K hand-off wait timed out inside an earlier contraction launch (partner block "
                    "not resident: CU mask or a competing kernel?); results produced since then are invalid. "
                    "sg_set_streamk(ctx, 0) selects the one-block-per-tile launches");
    }
    return SG_OK;
}

int sg_set_streamk(sg_ctx* ctx, int32_t enable) {
    if (!ctx) return SG_ERR_ARG;
    ctx->use_streamk = enable != 0;
    return SG_OK;
}

int sg_debug_lose_handoffs(sg_ctx* ctx, int32_t launches) {
    if (!ctx || launches < 0) return SG_ERR_ARG;
    ctx->lose_handoffs = launches;
    ctx->lose_feco = launches;
    return SG_OK;
}

int sg_xv_configure(sg_ctx* ctx, int32_t fft_bits) {
    if (!ctx) return SG_ERR_ARG;
    if (fft_bits != 32 && fft_bits != 64) return fail(ctx, SG_ERR_ARG, "fft_bits must be 32 or 64");
    ctx->tab.fft64 = fft_bits == 64;
    return SG_OK;
}

int sg_debug_feco_epoch(sg_ctx* ctx, uint32_t epoch) {
    if (!ctx) return SG_ERR_ARG;
    ctx->feco_epoch = epoch;
    return SG_OK;
}

int32_t sg_xv_num_frames(int32_t T) { return T < kWin ? 0 : num_frames(T); }

int sg_xv_load(sg_ctx* ctx, const sg_xv_weights* w) {
    if (!ctx || !w) return SG_ERR_ARG;
    if (w->D < 1 || w->D > 512 || w->S < 1 || w->S > 1024) return fail(ctx, SG_ERR_ARG, "D must be 1..512 and S 1..1024");
    for (int l = 0; l < kLayers; ++l)
        if (!w->tdnn_weight[l] || !w->tdnn_bias[l] || !w->bn_mean[l] || !w->bn_var[l])
            return fail(ctx, SG_ERR_ARG, "missing TDNN tensor for layer %d", l + 1);
    if (!w->fc1_weight || !w->fc1_bias || !w->emb_mean || !w->lda || !w->plda_mean || !w->plda_transform ||
        !w->plda_psi || !w->enroll)
        return fail(ctx, SG_ERR_ARG, "missing back-end tensor");
    SG_HIP(hipSetDevice(ctx->device));
    int rc = build_tables(ctx);
    if (rc) return rc;
    XvModel& m = ctx->xv;
    if (!m.allocs.empty()) {  // reload: release the previous model's tensors (nothing may still be reading them)
        SG_HIP(hipDeviceSynchronize());
        free_pool(m.allocs);
    }
    m = XvModel();
    std::vector<void*>& pool = m.allocs;
    const double eps = w->bn_eps > 0.f ? w->bn_eps : 1e-5;

    // BatchNorm1d(affine=False) in eval mode is y = (a - mean) * r, r = 1/sqrt(var + eps), applied
    // AFTER the ReLU (xvecTDNN.py:49-53).  It is folded into the NEXT layer: W'[co][ci][j] =
    // W[co][ci][j] * r[ci], b'[co] = b[co] - sum_{ci,j} W[co][ci][j] * mean[ci] * r[ci]; no padding
    // is used by the convolutions, so the fold is exact at every frame.  Stored activations are
    // therefore the ReLU outputs, and the ReLU mask of the backward pass is simply act > 0.
    std::vector<double> r_prev, m_prev;
    for (int l = 0; l < kLayers; ++l) {
        const int cin = kCin[l], cout = kCout[l], k = kTaps[l], cip = kCinPad[l], cop = kCoutPad[l];
        std::vector<float> wf((size_t)k * cip * cop, 0.f), wb((size_t)k * cop * cip, 0.f), bias(cop, 0.f);
        for (int co = 0; co < cout; ++co) {
            double bacc = w->tdnn_bias[l][co];
            for (int ci = 0; ci < cin; ++ci) {
                for (int j = 0; j < k; ++j) {
                    double v = w->tdnn_weight[l][((size_t)co * cin + ci) * k + j];
                    if (l > 0) {
                        bacc -= v * m_prev[ci] * r_prev[ci];
                        v *= r_prev[ci];
                    }
                    wf[((size_t)j * cip + ci) * cop + co] = (float)v;
                    wb[((size_t)j * cop + co) * cip + ci] = (float)v;
                }
            }
            bias[co] = (float)bacc;
        }
        rc |= dev_upload(ctx, pool, &m.wf[l], wf);
        rc |= dev_upload(ctx, pool, &m.wb[l], wb);
        rc |= dev_upload(ctx, pool, &m.wfq[l], pack_k4(wf, k * cip, cop));
        rc |= dev_upload(ctx, pool, &m.wbq[l], pack_k4(wb, k * cop, cip));
        rc |= dev_upload(ctx, pool, &m.bias[l], bias);
        r_prev.assign(cout, 0.0);
        m_prev.assign(cout, 0.0);
        for (int c = 0; c < cout; ++c) {
            r_prev[c] = 1.0 / std::sqrt((double)w->bn_var[l][c] + eps);
            m_prev[c] = w->bn_mean[l][c];
        }
    }
    {
        // fc1 over stats = [mean | std] of bn5 output: mean_y = (mean_a - m) r, std_y = std_a r
        std::vector<float> fw((size_t)kStats * kEmb, 0.f), fwt((size_t)kEmb * kStats, 0.f), fb(kEmb);
        const int c5 = kCout[4];
        for (int n = 0; n < kEmb; ++n) {
            double bacc = w->fc1_bias[n];
            for (int c = 0; c < c5; ++c) {
                const double wm = w->fc1_weight[(size_t)n * 2 * c5 + c];
                const double wsd = w->fc1_weight[(size_t)n * 2 * c5 + c5 + c];
                bacc -= wm * m_prev[c] * r_prev[c];
                const float fm = (float)(wm * r_prev[c]), fs = (float)(wsd * r_prev[c]);
                fw[(size_t)c * kEmb + n] = fm;
                fw[(size_t)(kPoolC + c) * kEmb + n] = fs;
                fwt[(size_t)n * kStats + c] = fm;
                fwt[(size_t)n * kStats + kPoolC + c] = fs;
            }
            fb[n] = (float)bacc;
        }
        rc |= dev_upload(ctx, pool, &m.fc1_w, fw);
        rc |= dev_upload(ctx, pool, &m.fc1_wt, fwt);
        rc |= dev_upload(ctx, pool, &m.fc1_b, fb);
    }
    const int D = w->D, S = w->S;
    {
        // device layouts for the tail kernel's 16-byte column loads: rows 16-byte aligned, zero padding
        const int Dp = (D + 3) & ~3;
        m.Dp = Dp;
        std::vector<float> lda((size_t)D * kLdaLd, 0.f), ldat((size_t)(kEmb + 1) * Dp, 0.f);
        for (int d = 0; d < D; ++d)
            for (int i = 0; i <= kEmb; ++i) {
                lda[(size_t)d * kLdaLd + i] = w->lda[(size_t)d * (kEmb + 1) + i];
                ldat[(size_t)i * Dp + d] = w->lda[(size_t)d * (kEmb + 1) + i];
            }
        std::vector<float> p((size_t)D * Dp, 0.f), pt((size_t)D * Dp, 0.f);
        for (int d = 0; d < D; ++d)
            for (int j = 0; j < D; ++j) {
                p[(size_t)d * Dp + j] = w->plda_transform[(size_t)d * D + j];
                pt[(size_t)j * Dp + d] = w->plda_transform[(size_t)d * D + j];
            }
        // (P A)[d][i] = sum_j P[d][j] A[j][i]: the tail's backward applies P^T and LDA^T as one product (k_tail.hip)
        std::vector<float> pa((size_t)D * kLdaLd, 0.f);
        {
            std::vector<double> row(kEmb);
            for (int d = 0; d < D; ++d) {
                std::fill(row.begin(), row.end(), 0.0);
                for (int j = 0; j < D; ++j) {
                    const double pdj = w->plda_transform[(size_t)d * D + j];
                    const float* aj = w->lda + (size_t)j * (kEmb + 1);
                    for (int i = 0; i < kEmb; ++i) row[i] += pdj * (double)aj[i];
                }
                for (int i = 0; i < kEmb; ++i) pa[(size_t)d * kLdaLd + i] = (float)row[i];
            }
        }
        double ldg = 0.0, ldw = 0.0;
        for (int d = 0; d < D; ++d) {
            const double psi = w->plda_psi[d];
            ldg += std::log(1.0 + psi / (psi + 1.0));
            ldw += std::log(psi + 1.0);
        }
        m.logdet_given = (float)ldg;
        m.logdet_without = (float)ldw;
        rc |= dev_upload(ctx, pool, &m.emb_mean, std::vector<float>(w->emb_mean, w->emb_mean + kEmb));
        rc |= dev_upload(ctx, pool, &m.lda, lda);
        rc |= dev_upload(ctx, pool, &m.lda_t, ldat);
        rc |= dev_upload(ctx, pool, &m.plda_mean, std::vector<float>(w->plda_mean, w->plda_mean + D));
        rc |= dev_upload(ctx, pool, &m.plda_p, p);
        rc |= dev_upload(ctx, pool, &m.plda_pt, pt);
        rc |= dev_upload(ctx, pool, &m.pa, pa);
        rc |= dev_upload(ctx, pool, &m.plda_psi, std::vector<float>(w->plda_psi, w->plda_psi + D));
        rc |= dev_upload(ctx, pool, &m.enroll, std::vector<float>(w->enroll, w->enroll + (size_t)S * D));
    }
    if (rc) return fail(ctx, SG_ERR_HIP, "model upload failed: %s", ctx->err.c_str());
    m.D = D;
    m.S = S;
    m.enroll_cap = S;
    m.threshold = w->threshold;
    m.loaded = true;
    return SG_OK;
}

int sg_xv_set_enroll(sg_ctx* ctx, const float* enroll_host, int32_t S, float threshold) {
    if (!ctx || !ctx->xv.loaded) return fail(ctx, SG_ERR_STATE, "no model loaded");
    if (S < 1 || S > 1024) return fail(ctx, SG_ERR_ARG, "S must be 1..1024");
    XvModel& m = ctx->xv;
    if (enroll_host) {
        SG_HIP(hipSetDevice(ctx->device));
        SG_HIP(hipDeviceSynchronize());  // earlier passes may still read the table
        if (S > m.enroll_cap) {  // grow: the old table goes back to the allocator, nothing accumulates
            float* fresh = nullptr;
            int rc = dev_alloc(ctx, m.allocs, &fresh, (size_t)S * m.D);
            if (rc) return rc;
            for (size_t i = 0; i < m.allocs.size(); ++i)
                if (m.allocs[i] == m.enroll) {
                    (void)hipFree(m.enroll);
                    m.allocs.erase(m.allocs.begin() + i);
                    break;
                }
            m.enroll = fresh;
            m.enroll_cap = S;
        }
        SG_HIP(hipMemcpy(m.enroll, enroll_host, (size_t)S * m.D * sizeof(float), hipMemcpyHostToDevice));
        m.S = S;
    }
    m.threshold = threshold;
    return SG_OK;
}

int sg_xv_enroll_override(sg_ctx* ctx, const float* enroll_dev, int32_t S) {
    if (!ctx || !ctx->xv.loaded) return fail(ctx, SG_ERR_STATE, "no model loaded");
    if (enroll_dev && (S < 1 || S > 1024)) return fail(ctx, SG_ERR_ARG, "S must be 1..1024");
    ctx->xv.enroll_override = enroll_dev;
    ctx->xv.S_override = enroll_dev ? S : 0;
    return SG_OK;
}

int sg_input_scale(sg_ctx* ctx, const float* x_dev, int64_t n, float* scale_dev, void* stream) {
    if (!ctx || !x_dev || !scale_dev || n < 1) return fail(ctx, SG_ERR_ARG, "bad argument");
    int rc = build_tables(ctx);
    if (rc) return rc;
    SG_HIP(launch_input_scale(x_dev, n, ctx->range_scratch, scale_dev, 0, (hipStream_t)stream));
    return SG_OK;
}

int sg_xv_mfcc(sg_ctx* ctx, const float* x_dev, int32_t B, int32_t T, const float* scale_dev, const sg_dither* dither,
               float* feats_dev, void* stream) {
    if (!ctx || !x_dev || !feats_dev || B < 1 || T < kWin) return fail(ctx, SG_ERR_ARG, "bad argument");
    int rc = build_tables(ctx);
    if (rc) return rc;
    SG_HIP(launch_mfcc_fwd(ctx->tab, x_dev, B, T, num_frames(T), scale_dev, dither, feats_dev, (hipStream_t)stream));
    return SG_OK;
}

int sg_xv_cmvn(sg_ctx* ctx, const float* feats_dev, int32_t B, int32_t F, float* out_dev, void* stream) {
    if (!ctx || !feats_dev || !out_dev || B < 1 || F < 1) return fail(ctx, SG_ERR_ARG, "bad argument");
    SG_HIP(launch_cmvn_fwd(feats_dev, kCep, out_dev, kCep, B, F, (hipStream_t)stream));
    return SG_OK;
}

int sg_xv_mfcc_backward(sg_ctx* ctx, const float* x_dev, int32_t B, int32_t T, const float* scale_dev,
                        const sg_dither* dither, const float* dfeats_dev, float* grad_dev, void* stream) {
    if (!x_dev || !dfeats_dev || !grad_dev) return fail(ctx, SG_ERR_ARG, "bad argument");
    PassDims d;
    int rc = check_dims(ctx, B, T, SG_FLAG_WAV, &d);  // sizes the per-frame gradient scratch
    if (rc) return rc;
    hipStream_t s = (hipStream_t)stream;
    Workspace& w = ctx->ws;
    SG_HIP(launch_mfcc_bwd(ctx->tab, x_dev, d.B, d.T, d.F, scale_dev, dither, dfeats_dev, w.dframes, s));
    SG_HIP(launch_frames_to_wave(w.dframes, d.B, d.T, d.F, 1, nullptr, grad_dev, nullptr, nullptr, nullptr, 0.f, 1, s));
    return SG_OK;
}

int sg_xv_cmvn_backward(sg_ctx* ctx, const float* dout_dev, int32_t B, int32_t F, float* din_dev, void* stream) {
    if (!ctx || !dout_dev || !din_dev || B < 1 || F < 1) return fail(ctx, SG_ERR_ARG, "bad argument");
    SG_HIP(launch_cmvn_bwd(dout_dev, kCep, 1, 0, din_dev, kCep, B, F, (hipStream_t)stream));
    return SG_OK;
}

int sg_xv_forward(sg_ctx* ctx, const float* x_dev, int32_t B, int32_t T_or_F, int32_t flag, const sg_dither* dither,
                  int64_t* decisions_dev, float* scores_dev, float* emb_dev, float* tdnn_emb_dev, void* stream) {
    PassDims d;
    int rc = check_dims(ctx, B, T_or_F, flag, &d);
    if (rc) return rc;
    if (!x_dev) return fail(ctx, SG_ERR_ARG, "x is NULL");
    hipStream_t s = (hipStream_t)stream;
    if ((rc = run_frontend(ctx, x_dev, d, flag, dither, false, s))) return rc;
    if ((rc = run_tdnn_forward(ctx, d, s))) return rc;
    Workspace& w = ctx->ws;
    TailArgs t{};
    t.fc1_part = w.fc1_part; t.nsplit = kFc1SplitK; t.B = B; t.m = &ctx->xv; t.y = nullptr; t.want_grad = 0;
    t.tdnn_emb = tdnn_emb_dev; t.emb = emb_dev; t.scores = scores_dev; t.decisions = decisions_dev;
    SG_STAGE(SG_STAGE_TAIL, launch_tail(t, s));
    return SG_OK;
}

int sg_xv_debug_activation(sg_ctx* ctx, int32_t layer, float* out_dev, int64_t capacity_floats, int32_t* rows_per_utt,
                           int32_t* channels, void* stream) {
    if (!ctx || layer < 1 || layer > kLayers || !ctx->ws.scale) return fail(ctx, SG_ERR_ARG, "bad layer or no pass run");
    const Workspace& w = ctx->ws;
    const int l = layer - 1;
    if (rows_per_utt) *rows_per_utt = w.Fl[l];
    if (channels) *channels = kCoutPad[l];
    if (out_dev) {
        if (capacity_floats <= 0) return fail(ctx, SG_ERR_ARG, "capacity must be positive");
        // never read past the activation buffer: a caller's capacity may exceed what the workspace holds (found by the
        // sanitizer build, `make asan`)
        int cap_fl[kLayers];
        layer_frames(w.F, cap_fl);
        const size_t held = (size_t)w.B * (size_t)(cap_fl[l] > 0 ? cap_fl[l] : 1) * kCoutPad[l];
        const size_t n = std::min<size_t>((size_t)capacity_floats, held);
        SG_HIP(hipMemcpyAsync(out_dev, w.act[l], n * sizeof(float), hipMemcpyDeviceToDevice, (hipStream_t)stream));
    }
    return SG_OK;
}

int sg_xv_loss_grad(sg_ctx* ctx, const float* x_dev, const int64_t* y_dev, int32_t B, int32_t T_or_F, int32_t flag,
                    const sg_loss_spec* loss, const sg_dither* dither, int64_t* decisions_dev, float* scores_dev,
                    float* loss_dev, float* grad_dev, void* stream) {
    PassDims d;
    int rc = check_dims(ctx, B, T_or_F, flag, &d);
    if (rc) return rc;
    if (!x_dev || !y_dev || !loss) return fail(ctx, SG_ERR_ARG, "x, y and loss are required");
    if (loss->loss == SG_LOSS_LINEAR && !loss->coef_dev) return fail(ctx, SG_ERR_ARG, "SG_LOSS_LINEAR needs coef_dev");
    hipStream_t s = (hipStream_t)stream;
    if ((rc = run_frontend(ctx, x_dev, d, flag, dither, grad_dev != nullptr, s))) return rc;
    if ((rc = run_tdnn_forward(ctx, d, s))) return rc;
    Workspace& w = ctx->ws;
    TailArgs t{};
    t.fc1_part = w.fc1_part; t.nsplit = kFc1SplitK; t.B = B; t.m = &ctx->xv; t.y = y_dev; t.loss = *loss;
    t.want_grad = grad_dev != nullptr;
    t.scores = scores_dev; t.decisions = decisions_dev; t.loss_out = loss_dev; t.demb = w.demb;
    SG_STAGE(SG_STAGE_TAIL, launch_tail(t, s));
    if (grad_dev) {
        rc = run_backward_to_input(ctx, x_dev, d, flag, dither, grad_dev, nullptr, nullptr, nullptr, 0.f, 0, s);
        if (rc) return rc;
    }
    return SG_OK;
}

int sg_pgd_update(sg_ctx* ctx, float* x_dev, const float* grad_dev, const float* lower_dev, const float* upper_dev,
                  int64_t n, float step_size, int32_t grad_sign, void* stream) {
    if (!ctx || !x_dev || !grad_dev || !lower_dev || !upper_dev || n < 1) return fail(ctx, SG_ERR_ARG, "bad argument");
    SG_HIP(launch_pgd_update(x_dev, grad_dev, lower_dev, upper_dev, n, step_size, grad_sign, (hipStream_t)stream));
    return SG_OK;
}

int sg_loss_eval(sg_ctx* ctx, const float* scores_dev, const int64_t* y_dev, int32_t B, int32_t S, float threshold,
                 const sg_loss_spec* loss, int64_t* decisions_dev, float* loss_dev, float* dscores_dev, void* stream) {
    if (!ctx || !scores_dev || !loss || !dscores_dev || B < 1 || S < 1) return fail(ctx, SG_ERR_ARG, "bad argument");
    if (loss->loss == SG_LOSS_LINEAR && !loss->coef_dev) return fail(ctx, SG_ERR_ARG, "SG_LOSS_LINEAR needs coef_dev");
    SG_HIP(hipSetDevice(ctx->device));
    SG_HIP(launch_loss_eval(scores_dev, y_dev, B, S, threshold, *loss, decisions_dev, loss_dev, dscores_dev, (hipStream_t)stream));
    return SG_OK;
}

int sg_cw2_step(sg_ctx* ctx, float* modifier_dev, float* exp_avg_dev, float* exp_avg_sq_dev, const float* x_dev,
                const float* input_cur_dev, const float* grad1_dev, const float* const_dev, int32_t B, int32_t T, float lr,
                int32_t step_t, float* input_next_dev, float* loss2_dev, void* stream) {
    if (!ctx || !modifier_dev || !x_dev || !input_next_dev || !loss2_dev || B < 1 || B > 1024 || T < 1)
        return fail(ctx, SG_ERR_ARG, "bad argument");
    if (grad1_dev && (!exp_avg_dev || !exp_avg_sq_dev || !input_cur_dev || !const_dev || step_t < 1))
        return fail(ctx, SG_ERR_ARG, "an update needs the Adam state, the current input, const and step >= 1");
    int rc = build_tables(ctx);
    if (rc) return rc;
    SG_HIP(launch_cw2_step(modifier_dev, exp_avg_dev, exp_avg_sq_dev, x_dev, input_cur_dev, grad1_dev, const_dev, B, T, lr,
                           step_t, input_next_dev, loss2_dev, ctx->cw2_scratch, (hipStream_t)stream));
    return SG_OK;
}

int sg_nes_queries(sg_ctx* ctx, const float* x_dev, int32_t n, int32_t T, int32_t half, int32_t with_clean, float sigma,
                   uint64_t seed, int64_t index_base, int32_t pair_base, const float* noise_in_dev, float* queries_dev,
                   float* noise_out_dev, void* stream) {
    if (!ctx || !x_dev || !queries_dev || n < 1 || T < 1 || half < 1) return fail(ctx, SG_ERR_ARG, "bad argument");
    SG_HIP(launch_nes_queries(x_dev, n, T, half, with_clean != 0, sigma, seed, index_base, pair_base, noise_in_dev,
                              queries_dev, noise_out_dev, (hipStream_t)stream));
    return SG_OK;
}

int sg_nes_grad(sg_ctx* ctx, const float* loss_dev, int32_t n, int32_t T, int32_t half, int32_t with_clean, uint64_t seed,
                int64_t index_base, int32_t pair_base, const float* noise_in_dev, int32_t accumulate, float final_sigma,
                int32_t final_batches, float* grad_dev, void* stream) {
    if (!ctx || !loss_dev || !grad_dev || n < 1 || T < 1 || half < 1) return fail(ctx, SG_ERR_ARG, "bad argument");
    SG_HIP(launch_nes_grad(loss_dev, n, T, half, with_clean != 0, seed, index_base, pair_base, noise_in_dev, accumulate,
                           final_sigma, final_batches > 0 ? final_batches : 1, grad_dev, (hipStream_t)stream));
    return SG_OK;
}

int sg_fakebob_step(sg_ctx* ctx, float* x_dev, float* grad_dev, const float* prev_grad_dev, const float* lr_dev,
                    const float* lower_dev, const float* upper_dev, int32_t n, int32_t T, float momentum,
                    float one_minus_momentum, int32_t grad_sign, void* stream) {
    if (!ctx || !x_dev || !grad_dev || !prev_grad_dev || !lr_dev || !lower_dev || !upper_dev || n < 1 || T < 1)
        return fail(ctx, SG_ERR_ARG, "bad argument");
    SG_HIP(launch_fakebob_step(x_dev, grad_dev, prev_grad_dev, lr_dev, lower_dev, upper_dev, n, T, momentum,
                               one_minus_momentum, grad_sign, (hipStream_t)stream));
    return SG_OK;
}

int sg_xv_pgd_run(sg_ctx* ctx, float* x_adv_dev, const int64_t* y_dev, const float* lower_dev, const float* upper_dev,
                  int32_t B, int32_t T, const sg_pgd_params* p, uint8_t* success_dev, int64_t* decisions_dev,
                  float* scores_dev, float* loss_dev, float* loss_trace_dev, int64_t* decision_trace_dev, void* stream) {
    int rc;
    if (!ctx) return SG_ERR_ARG;
    if (!x_adv_dev || !y_dev || !lower_dev || !upper_dev || !p) return fail(ctx, SG_ERR_ARG, "NULL argument");
    if (B < 1 || T < kWin) return fail(ctx, SG_ERR_ARG, "need B >= 1 and a waveform of at least one 25 ms window");
    if (p->max_iter < 0) return fail(ctx, SG_ERR_ARG, "max_iter must be >= 0");
    if (p->loss.loss == SG_LOSS_LINEAR && !p->loss.coef_dev) return fail(ctx, SG_ERR_ARG, "SG_LOSS_LINEAR needs coef_dev");
    const int eot_size = p->eot_size > 0 ? p->eot_size : 1, eot_bs = p->eot_batch_size > 0 ? p->eot_batch_size : 1;
    if (eot_size % eot_bs) return fail(ctx, SG_ERR_ARG, "EOT size should be divisible by EOT batch size");
    // Expectation over the front-end's random dither (adaptive_attack/EOT.py:16-54; the reference hard-codes
    // dither = 1.0 at xv_plda.py:119, so this IS its default behaviour): every gradient step runs eot_size passes
    // with fresh noise, the data gradients are summed in pass order and the sign step is taken on the sum
    // (sign(mean) == sign(sum); the EOT_batch_size grouping only changes how the reference batches the repeats).
    // The final pass of the attack is a single forward (FGSM.py:45-47).  Without dither the model is
    // deterministic, every repeat is the same computation and their mean is the single-pass result: one pass.
    const int reps = p->dither.dither != 0.f ? eot_size : 1;
    hipStream_t s = (hipStream_t)stream;
    // The repeats of a step are independent passes over the same audio: they run as ONE batch of G x B rows (row = repeat
    // * B + utterance; the MFCC kernels read the utterance's waveform for every repeat and key repeat r's dither by
    // seed + r * 0xC2B2AE3D27D4EB4F), where the contractions are more efficient than at B rows (24.7 k utterance-steps/s
    // at 128 rows, 25.3 k at 256 against 23.2 k at 64).  Per-row arithmetic does not depend on the batch and the
    // overlap-add sums the repeats in repeat order, so the result is bit for bit that of the repeats run one after the
    // other.  G: as many repeats as one pass may hold (activation tensors < 2 GiB); more run as further groups, the sum
    // handed on through ws.grad.
    int G = 1;
    if (reps > 1) {
        int Fl[kLayers];
        const int F = num_frames(T);
        if (!layer_frames(F, Fl)) return fail(ctx, SG_ERR_ARG, "%d frames are too few for the TDNN context", F);
        size_t per_utt = 0;
        for (int l = 0; l < kLayers; ++l) per_utt = std::max(per_utt, (size_t)Fl[l] * kCoutPad[l] * sizeof(float));
        long max_rows = (long)(0x7FFFFFFFull / per_utt);
        if (const char* e = sg_tune_env("SG_EOT_MAX_ROWS")) max_rows = std::min<long>(max_rows, atol(e));  // tests: force groups
        G = (int)std::min<long>(reps, std::max<long>(1, max_rows / B));
    }
    PassDims d;
    rc = check_dims(ctx, B * G, T, SG_FLAG_WAV, &d);  // workspace for the largest pass
    if (rc) return rc;
    Workspace& w = ctx->ws;
    for (int r = 0; r < G; ++r)
        SG_HIP(hipMemcpyAsync(w.y_rep + (size_t)r * B, y_dev, (size_t)B * sizeof(int64_t), hipMemcpyDeviceToDevice, s));
    // per-step records (attack/FGSM.py:50-58: loss averaged, decision voted over ALL EOT repeats of the step): when the
    // repeats of a step run as several passes (G < reps) every pass leaves its rows here and the last one reduces them
    const bool want_rec = loss_trace_dev || decision_trace_dev;
    if (want_rec && G < reps && w.eot_rows_cap < (size_t)reps * B) {
        // grown: the old pair is released (nothing enqueued still uses it once the stream has drained), not left in the
        // workspace's pool until sg_destroy
        if (w.eot_loss_rows || w.eot_dec_rows) {
            SG_HIP(hipStreamSynchronize(s));
            for (void* old : {static_cast<void*>(w.eot_loss_rows), static_cast<void*>(w.eot_dec_rows)}) {
                auto it = std::find(w.allocs.begin(), w.allocs.end(), old);
                if (it != w.allocs.end()) {
                    (void)hipFree(old);
                    w.allocs.erase(it);
                }
            }
            w.eot_loss_rows = nullptr;
            w.eot_dec_rows = nullptr;
            w.eot_rows_cap = 0;
        }
        if ((rc = dev_alloc(ctx, w.allocs, &w.eot_loss_rows, (size_t)reps * B))) return rc;
        if ((rc = dev_alloc(ctx, w.allocs, &w.eot_dec_rows, (size_t)reps * B))) return rc;
        w.eot_rows_cap = (size_t)reps * B;
    }
    for (int it = 0; it <= p->max_iter; ++it) {
        const bool last = it == p->max_iter;
        const int nrep = last ? 1 : reps;
        for (int g0 = 0; g0 < nrep; g0 += G) {
            const int Gi = std::min(G, nrep - g0);
            sg_dither dz = p->dither;
            dz.seed += (uint64_t)it * 0x9E3779B97F4A7C15ull + (uint64_t)g0 * 0xC2B2AE3D27D4EB4Full;
            d.B = B * Gi;
            d.Bu = Gi > 1 ? B : 0;
            // every iterate is clamped into [lower, upper] within [-1, 1], so check_input_range takes the
            // same branch as for the start point: decide once
            d.keep_scale = it > 0 || g0 > 0;
            if ((rc = run_frontend(ctx, x_adv_dev, d, SG_FLAG_WAV, &dz, !last, s))) return rc;
            if ((rc = run_tdnn_forward(ctx, d, s))) return rc;
            TailArgs t{};
            t.fc1_part = w.fc1_part; t.nsplit = kFc1SplitK; t.B = d.B; t.m = &ctx->xv; t.y = w.y_rep; t.loss = p->loss;
            t.want_grad = !last; t.demb = w.demb;
            t.scores = last ? scores_dev : nullptr;
            t.decisions = last ? decisions_dev : nullptr;
            t.loss_out = last ? loss_dev : nullptr;
            t.success = last ? success_dev : nullptr;
            // per-step records as the reference prints them (attack/FGSM.py:50-58): the loss averaged over the step's EOT
            // repeats and the decision voted over them (attack/utils.py:118-125).  A pass of several repeats records its
            // rows into the workspace and a small reduction writes the step's row; with more repeats than one pass holds
            // (G < reps: activations past 2 GiB, or SG_EOT_MAX_ROWS) the passes of the step collect their rows in repeat
            // order and the reduction runs after the last one, over all `nrep` repeats.
            const bool direct = nrep == 1, grouped = nrep > G;
            float* lrows = grouped ? w.eot_loss_rows + (size_t)g0 * B : w.loss;
            int64_t* drows = grouped ? w.eot_dec_rows + (size_t)g0 * B : w.decisions;
            t.loss_trace = !want_rec ? nullptr : (direct && loss_trace_dev ? loss_trace_dev + (size_t)it * B : lrows);
            t.decision_trace = !want_rec ? nullptr : (direct && decision_trace_dev ? decision_trace_dev + (size_t)it * B : drows);
            t.coef_rows = B;  // SG_LOSS_LINEAR: the caller's (B, S) table serves every repeat of an utterance
            SG_STAGE(SG_STAGE_TAIL, launch_tail(t, s));
            if (want_rec && !direct && g0 + Gi >= nrep)
                SG_HIP(launch_eot_trace_reduce(grouped ? w.eot_loss_rows : w.loss, grouped ? w.eot_dec_rows : w.decisions, nrep, B,
                                               loss_trace_dev ? loss_trace_dev + (size_t)it * B : nullptr,
                                               decision_trace_dev ? decision_trace_dev + (size_t)it * B : nullptr, s));
            if (!last) {
                const bool final_group = g0 + Gi >= nrep;
                rc = run_backward_to_input(ctx, x_adv_dev, d, SG_FLAG_WAV, &dz, final_group ? nullptr : w.grad,
                                           final_group ? x_adv_dev : nullptr, lower_dev, upper_dev, p->step_size, p->grad_sign, s,
                                           g0 > 0 ? w.grad : nullptr);
                if (rc) return rc;
            }
        }
    }
    return SG_OK;
}

int sg_trace_begin(sg_ctx* ctx, int32_t max_records) {
    if (!ctx) return SG_ERR_ARG;
    if (max_records < 1 || max_records > (1 << 20)) return fail(ctx, SG_ERR_ARG, "max_records must be 1 .. 2^20");
    SG_HIP(hipSetDevice(ctx->device));
    while ((int)ctx->trace_ev.size() < 2 * max_records) {
        hipEvent_t e = nullptr;
        SG_HIP(hipEventCreate(&e));
        ctx->trace_ev.push_back(e);
    }
    ctx->trace_tag.assign((size_t)max_records, 0);
    ctx->trace_used = 0;
    ctx->trace_open = false;
    ctx->trace_dropped = 0;
    ctx->trace_on = true;
    return SG_OK;
}

int sg_trace_end(sg_ctx* ctx, int32_t* tags_out, float* ms_out, int32_t capacity, int32_t* n_out) {
    if (!ctx || !n_out) return SG_ERR_ARG;
    if (!ctx->trace_on) return fail(ctx, SG_ERR_STATE, "sg_trace_end without sg_trace_begin");
    ctx->trace_on = false;
    ctx->trace_open = false;
    const int n = ctx->trace_used;
    *n_out = n;
    if (ctx->trace_dropped)
        return fail(ctx, SG_ERR_HIP, "stage trace: %d event record(s) failed, %d launch records kept", ctx->trace_dropped, n);
    if (n > 0) SG_HIP(hipEventSynchronize(ctx->trace_ev[2 * (n - 1) + 1]));
    for (int i = 0; i < n && i < capacity; ++i) {
        float ms = 0.f;
        SG_HIP(hipEventElapsedTime(&ms, ctx->trace_ev[2 * i], ctx->trace_ev[2 * i + 1]));
        if (tags_out) tags_out[i] = ctx->trace_tag[i];
        if (ms_out) ms_out[i] = ms;
    }
    return SG_OK;
}

int sg_conv1d_rows(sg_ctx* ctx, const float* a_dev, const float* w_dev, float* c_dev, const float* bias_dev,
                   const float* mask_dev, int32_t B, int32_t Ta, int32_t Tc, int32_t Kc, int32_t N, int32_t taps,
                   int32_t tap_step, int32_t tap_base, int32_t epi, int32_t kernel, void* stream) {
    if (!ctx) return SG_ERR_ARG;
    if (!a_dev || !w_dev || !c_dev || B <= 0 || Ta <= 0 || Tc <= 0 || taps <= 0 || Kc <= 0 || Kc % 32 || N <= 0 || N % 128)
        return fail(ctx, SG_ERR_ARG, "sg_conv1d_rows: need Kc %% 32 == 0 and N %% 128 == 0 (got Kc=%d N=%d)", Kc, N);
    if (epi < 0 || epi > 2 || (epi == 1 && !bias_dev) || (epi == 2 && !mask_dev) || kernel < 0 || kernel > 10)
        return fail(ctx, SG_ERR_ARG, "sg_conv1d_rows: bad epi/kernel");
    if (build_tables(ctx) != SG_OK) return SG_ERR_HIP;
    hipStream_t s = (hipStream_t)stream;
    float* wq = nullptr;
    SG_HIP(hipMalloc(reinterpret_cast<void**>(&wq), (size_t)taps * Kc * N * sizeof(float)));
    hipError_t e = launch_pack_k4(w_dev, taps * Kc, N, wq, s);
    ConvGemmArgs a{};
    a.A = a_dev; a.W = w_dev; a.Wq = wq; a.C = c_dev; a.bias = bias_dev; a.mask = mask_dev;
    a.M = B * Tc; a.N = N; a.Ta = Ta; a.Tc = Tc; a.Kc = Kc; a.lda = Kc; a.ldw = N; a.ldc = N;
    a.taps = taps; a.tap_step = tap_step; a.tap_base = tap_base;
    a.total_chunks = taps * (Kc / 32); a.chunks_per_split = a.total_chunks;
    conv_ctx_args(ctx, a);
    a.force = kernel;
    if (e == hipSuccess) e = launch_conv_gemm(a, 0, epi, 1, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    (void)hipFree(wq);
    if (e != hipSuccess) return fail(ctx, SG_ERR_HIP, "sg_conv1d_rows: %s", hipGetErrorString(e));
    return SG_OK;
}

int sg_xv_time_layer(sg_ctx* ctx, int32_t layer, int32_t B, int32_t T, int32_t iters, float* ms_per_launch,
                     double* flops, int32_t* tile_rows, void* stream) {
    if (!ctx || !ctx->xv.loaded) return fail(ctx, SG_ERR_STATE, "no model loaded");
    const int l = (layer < 0 ? -layer : layer) - 1;
    if (l < 0 || l >= kLayers || iters < 1 || !ms_per_launch) return fail(ctx, SG_ERR_ARG, "bad argument");
    const int F = num_frames(T);
    Workspace& w = ctx->ws;
    if (!w.scale || B > w.B || F > w.F)
        return fail(ctx, SG_ERR_STATE, "run a forward pass with this (B, T) first so the activations are resident");
    hipStream_t s = (hipStream_t)stream;
    layer_frames(F, w.Fl);
    ConvGemmArgs a = fwd_layer_args(ctx, l, B, F);
    int tile = 0, epi = EPI_BIAS_RELU, splits = 1;
    if (layer < 0) {  // data-gradient contraction of the same layer (reads d(out), writes d(in))
        a = ConvGemmArgs{};
        a.A = w.dact[l]; a.W = ctx->xv.wb[l]; a.Wq = ctx->xv.wbq[l]; a.C = l == 0 ? w.dfeats : w.dact[l - 1];
        a.mask = l == 0 ? nullptr : w.act[l - 1];
        a.Ta = w.Fl[l]; a.Tc = l == 0 ? F : w.Fl[l - 1]; a.M = B * a.Tc; a.N = kCinPad[l]; a.Kc = kCoutPad[l];
        a.lda = kCoutPad[l]; a.ldw = kCinPad[l]; a.ldc = kCinPad[l]; a.taps = kTaps[l]; a.tap_step = -kDil[l];
        a.total_chunks = a.taps * (a.Kc / 32); a.chunks_per_split = a.total_chunks;
        conv_ctx_args(ctx, a);
        tile = l == 0 ? 1 : 0;
        epi = l == 0 ? EPI_NONE : EPI_RELU_MASK;
        if (l == 0) {
            splits = kL1BwdSplitK;
            a.chunks_per_split = a.total_chunks / kL1BwdSplitK;
            a.split_stride = (long long)B * F * kFeatPad;
        }
    }
    SG_HIP(launch_conv_gemm(a, tile, epi, splits, s));  // warm
    SG_HIP(hipEventRecord(ctx->ev0, s));
    for (int i = 0; i < iters; ++i) SG_HIP(launch_conv_gemm(a, tile, epi, splits, s));
    SG_HIP(hipEventRecord(ctx->ev1, s));
    SG_HIP(hipEventSynchronize(ctx->ev1));
    float ms = 0.f;
    SG_HIP(hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1));
    *ms_per_launch = ms / (float)iters;
    if (tile_rows) *tile_rows = tile == 0 ? conv_gemm_tile_rows(a.M, a.N) : 128;
    if (flops) *flops = 2.0 * (double)a.M * (double)kCout[l] * (double)kCin[l] * (double)kTaps[l];
    return SG_OK;
}

}  // extern "C"
