// C-ABI entry points of the AudioNet CSI-NE path (include/speakerguard_hip.h, "sg_an_*"):
// model load (BatchNorm folding), workspace, forward / backward / PGD kernel sequences.
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>

#include "sg_internal.h"

using namespace sg;

namespace {

int an_fail(sg_ctx* ctx, int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    if (ctx) ctx->err = buf;
    return code;
}

#define AN_HIP(expr)                                                                                  \
    do {                                                                                              \
        hipError_t e_ = (expr);                                                                       \
        if (e_ != hipSuccess) return an_fail(ctx, SG_ERR_HIP, "%s failed: %s (%s:%d)", #expr,         \
                                             hipGetErrorString(e_), __FILE__, __LINE__);              \
    } while (0)

#define AN_STAGE(tag, expr)           \
    do {                              \
        trace_mark(ctx, (tag), s, 0); \
        AN_HIP(expr);                 \
        trace_mark(ctx, (tag), s, 1); \
    } while (0)

template <typename T>
int an_alloc(sg_ctx* ctx, std::vector<void*>& pool, T** out, size_t count) {
    void* p = nullptr;
    AN_HIP(hipMalloc(&p, count * sizeof(T) + 256));
    pool.push_back(p);
    *out = reinterpret_cast<T*>(p);
    return SG_OK;
}
template <typename T>
int an_upload(sg_ctx* ctx, std::vector<void*>& pool, T** out, const std::vector<T>& host) {
    int rc = an_alloc(ctx, pool, out, host.size());
    if (rc) return rc;
    AN_HIP(hipMemcpy(*out, host.data(), host.size() * sizeof(T), hipMemcpyHostToDevice));
    return SG_OK;
}

// ---- front-end tables: periodic hann(800), librosa-0.8.0 slaney mel basis (Preprocessor.py:57-60) ----
double hz_to_mel(double f) {
    const double f_sp = 200.0 / 3.0, min_log_hz = 1000.0, logstep = std::log(6.4) / 27.0;
    return f >= min_log_hz ? min_log_hz / f_sp + std::log(f / min_log_hz) / logstep : f / f_sp;
}
double mel_to_hz(double m) {
    const double f_sp = 200.0 / 3.0, min_log_hz = 1000.0, min_log_mel = min_log_hz / f_sp, logstep = std::log(6.4) / 27.0;
    return m >= min_log_mel ? min_log_hz * std::exp(logstep * (m - min_log_mel)) : f_sp * m;
}

int an_build_tables(sg_ctx* ctx) {
    if (ctx->an_tables_ready) return SG_OK;
    const double PI = 3.14159265358979323846;
    std::vector<float> window(kAnWin), melw((size_t)kAnMel * kAnBins, 0.f), w0(kAnBins, 0.f), w1(kAnBins, 0.f);
    std::vector<int> lo(kAnMel), hi(kAnMel), m0(kAnBins, -1);
    std::vector<double2> tw(kAnFft / 2);
    std::vector<uint16_t> br(kAnFft);
    {   // torch.hann_window(800) (periodic): arange * (2 pi / 800) -> cos -> * -0.5 + 0.5, float32
        const float step = (float)(PI * 2.0 / (double)kAnWin);
        for (int n = 0; n < kAnWin; ++n) window[n] = cosf((float)n * step) * -0.5f + 0.5f;
    }
    {   // librosa.filters.mel(16000, 1024, 32, fmin=0, fmax=8000): float64 arithmetic, cast to float32
        std::vector<double> mel_f(kAnMel + 2);
        const double mmin = hz_to_mel(0.0), mmax = hz_to_mel(8000.0);
        for (int i = 0; i < kAnMel + 2; ++i) mel_f[i] = mel_to_hz(mmin + (mmax - mmin) * i / (kAnMel + 1));
        for (int m = 0; m < kAnMel; ++m) {
            const double enorm = 2.0 / (mel_f[m + 2] - mel_f[m]);
            lo[m] = kAnBins;
            hi[m] = 0;
            for (int k = 0; k < kAnBins; ++k) {
                const double fr = 8000.0 * k / (kAnBins - 1);
                const double lower = (fr - mel_f[m]) / (mel_f[m + 1] - mel_f[m]);
                const double upper = (mel_f[m + 2] - fr) / (mel_f[m + 2] - mel_f[m + 1]);
                const double w = std::fmax(0.0, std::fmin(lower, upper)) * enorm;
                melw[(size_t)m * kAnBins + k] = (float)w;
                if (w > 0.0) {
                    if (k < lo[m]) lo[m] = k;
                    hi[m] = k + 1;
                }
            }
            if (lo[m] > hi[m]) lo[m] = hi[m] = 0;
        }
        for (int k = 0; k < kAnBins; ++k) {
            int first = -1, cnt = 0;
            for (int m = 0; m < kAnMel; ++m)
                if (melw[(size_t)m * kAnBins + k] > 0.f) {
                    if (first < 0) first = m;
                    ++cnt;
                }
            if (cnt > 2 || (cnt == 2 && melw[(size_t)(first + 1) * kAnBins + k] <= 0.f))
                return an_fail(ctx, SG_ERR_STATE, "mel filterbank is not a two-overlap triangular bank");
            m0[k] = first;
            if (first >= 0) {
                w0[k] = melw[(size_t)first * kAnBins + k];
                w1[k] = first + 1 < kAnMel ? melw[(size_t)(first + 1) * kAnBins + k] : 0.f;
            }
        }
    }
    for (int k = 0; k < kAnFft / 2; ++k) tw[k] = make_double2(std::cos(2.0 * PI * k / kAnFft), -std::sin(2.0 * PI * k / kAnFft));
    for (int i = 0; i < kAnFft; ++i) {
        int r = 0;
        for (int bit = 0; bit < 10; ++bit)
            if (i & (1 << bit)) r |= 1 << (9 - bit);
        br[i] = (uint16_t)r;
    }
    AnTables& t = ctx->an_tab;
    std::vector<void*>& pool = ctx->model_allocs;
    int rc = 0;
    rc |= an_upload(ctx, pool, &t.window, window);
    rc |= an_upload(ctx, pool, &t.mel_w, melw);
    rc |= an_upload(ctx, pool, &t.mel_lo, lo);
    rc |= an_upload(ctx, pool, &t.mel_hi, hi);
    {   // the per-lane views of window and filterbank the front-end kernels keep in LDS (k_audionet.hip, AnLaneTab)
        // mel: every filter is cut into ceil(width / 20) runs of consecutive bins of (nearly) equal length, one lane per run
        // (971 filter taps over 63 lanes, <= 20 each; the two-lanes-per-filter split of rounds 3-4 was bound by its widest
        // half, 44 taps, 64 % of them zero padding).  mel_seg[m] = first lane | runs << 8: lane m adds the runs' sums up in
        // ascending order.
        constexpr int kLaneBins = 20;  // = kAnMelLaneBins
        std::vector<float> lwin(16 * 64), lmelw((size_t)kLaneBins * 64, 0.f);
        std::vector<int> lk0(64, 0), seg(kAnMel, 0);
        for (int tap = 0; tap < 16; ++tap)
            for (int l = 0; l < 64; ++l) {
                const int n = 2 * (l + 64 * (tap >> 1)) + (tap & 1) - (kAnFft - kAnWin) / 2;
                lwin[tap * 64 + l] = (n >= 0 && n < kAnWin) ? window[n] : 0.f;
            }
        int lanes = 0;
        for (int m = 0; m < kAnMel; ++m) {
            const int w = hi[m] - lo[m], runs = w > 0 ? (w + kLaneBins - 1) / kLaneBins : 1;
            if (lanes + runs > 64 || runs > 5) return an_fail(ctx, SG_ERR_STATE, "mel filterbank needs more than 64 runs of %d bins", kLaneBins);
            seg[m] = lanes | (runs << 8);
            for (int i = 0; i < runs; ++i, ++lanes) {
                const int k0 = lo[m] + (int)((long long)w * i / runs), k1 = lo[m] + (int)((long long)w * (i + 1) / runs);
                lk0[lanes] = k0;
                for (int j = 0; j < k1 - k0; ++j) lmelw[(size_t)j * 64 + lanes] = melw[(size_t)m * kAnBins + k0 + j];
            }
        }
        rc |= an_upload(ctx, pool, &t.mel_seg, seg);
        rc |= an_upload(ctx, pool, &t.lane_win, lwin);
        rc |= an_upload(ctx, pool, &t.lane_melw, lmelw);
        rc |= an_upload(ctx, pool, &t.lane_k0, lk0);
    }
    {   // twiddle tables of the 512-point complex transform (W512^m = W1024^(2 m)), as fft512_fill_tables lays them out
        auto w512 = [&](int m) {
            const double2 w = tw[(2 * (m & 255)) % (kAnFft / 2)];
            return (m & 256) ? make_double2(-w.x, -w.y) : w;
        };
        std::vector<double2> t1(7 * 64), t2(72);
        std::vector<float2> t1f(7 * 64), t2f(72);
        for (int i = 0; i < 7 * 64; ++i) t1[i] = w512((i / 64 + 1) * (i % 64));
        for (int i = 0; i < 72; ++i) {
            const int b = i / 9, c = i - 9 * b;
            t2[i] = c < 8 ? w512(8 * b * c) : make_double2(0.0, 0.0);
        }
        for (int i = 0; i < 7 * 64; ++i) t1f[i] = make_float2((float)t1[i].x, (float)t1[i].y);
        for (int i = 0; i < 72; ++i) t2f[i] = make_float2((float)t2[i].x, (float)t2[i].y);
        rc |= an_upload(ctx, pool, &t.tw1d, t1);
        rc |= an_upload(ctx, pool, &t.tw2d, t2);
        rc |= an_upload(ctx, pool, &t.tw1f, t1f);
        rc |= an_upload(ctx, pool, &t.tw2f, t2f);
    }
    rc |= an_upload(ctx, pool, &t.bin_m0, m0);
    rc |= an_upload(ctx, pool, &t.bin_w0, w0);
    rc |= an_upload(ctx, pool, &t.bin_w1, w1);
    rc |= an_upload(ctx, pool, &t.twiddle, tw);
    rc |= an_upload(ctx, pool, &t.bitrev, br);
    if (!ctx->range_scratch) rc |= an_alloc(ctx, pool, &ctx->range_scratch, 512);
    if (rc) return SG_ERR_HIP;
    ctx->an_tables_ready = true;
    return SG_OK;
}

// frames entering / leaving every conv; false if the utterance is too short for conv8 (kernel 3, no pad)
bool an_layer_frames(int F, int* Tin, int* Tout) {
    int t = F;
    for (int l = 0; l < kAnConv; ++l) {
        Tin[l] = t;
        Tout[l] = t + 2 * kAnPad[l] - 2;
        if (Tout[l] < 1) return false;
        t = kAnPool[l] ? Tout[l] / 2 : Tout[l];
        if (t < 1) return false;
    }
    return Tin[kAnConv - 1] >= 3;
}

int an_ensure_workspace(sg_ctx* ctx, int B, int T, int F) {
    AnWorkspace& w = ctx->an_ws;
    if (B <= w.B && F <= w.F && T <= w.T && w.scale) {
        an_layer_frames(F, w.Tin, w.Tout);
        return SG_OK;
    }
    (void)hipDeviceSynchronize();
    for (void* p : w.allocs) (void)hipFree(p);
    const int cb = B > w.B ? B : w.B, cf = F > w.F ? F : w.F, ct = T > w.T ? T : w.T;
    w = AnWorkspace();
    w.B = cb; w.F = cf; w.T = ct;
    int Tin[kAnConv], Tout[kAnConv];
    an_layer_frames(cf, Tin, Tout);
    const size_t b = (size_t)cb;
    int rc = 0;
    rc |= an_alloc(ctx, w.allocs, &w.scale, 4);
    rc |= an_alloc(ctx, w.allocs, &w.feats, b * cf * kAnMel);
    rc |= an_alloc(ctx, w.allocs, &w.pre, b * cf * kAnMel);
    rc |= an_alloc(ctx, w.allocs, &w.dpre, b * cf * kAnMel);
    rc |= an_alloc(ctx, w.allocs, &w.dfeats, b * cf * kAnMel);
    rc |= an_alloc(ctx, w.allocs, &w.dframes, b * cf * kAnWin);
    rc |= an_alloc(ctx, w.allocs, &w.mel_cache, b * cf * kAnMel);
    rc |= an_alloc(ctx, w.allocs, &w.feco_ids, b * cf);
    rc |= an_alloc(ctx, w.allocs, &w.feco_cnt, b * cf);
    rc |= an_alloc(ctx, w.allocs, &w.feco_out, b * cf * kAnMel);
    rc |= an_alloc(ctx, w.allocs, &w.dfeco, b * cf * kAnMel);
    rc |= an_alloc(ctx, w.allocs, &w.y_rep, b);
    rc |= an_alloc(ctx, w.allocs, &w.trace_l, b);
    rc |= an_alloc(ctx, w.allocs, &w.trace_d, b);
    for (int l = 0; l < kAnConv; ++l) {
        const size_t n = b * (size_t)(Tout[l] > 0 ? Tout[l] : 1) * kAnCout[l];
        rc |= an_alloc(ctx, w.allocs, &w.act[l], n);
        rc |= an_alloc(ctx, w.allocs, &w.dact[l], n);
        if (kAnPool[l]) {
            rc |= an_alloc(ctx, w.allocs, &w.pool[l], n / 2 + kAnCout[l]);
            rc |= an_alloc(ctx, w.allocs, &w.dpool[l], n / 2 + kAnCout[l]);
        }
    }
    if (rc) {
        for (void* p : w.allocs) (void)hipFree(p);
        w = AnWorkspace();
        return SG_ERR_HIP;
    }
    an_layer_frames(F, w.Tin, w.Tout);
    return SG_OK;
}

struct AnDims {
    int B, T, F;
    bool keep_scale = false;
};

int an_check(sg_ctx* ctx, int B, int TF, int flag, AnDims* d) {
    if (!ctx) return SG_ERR_ARG;
    if (!ctx->an.loaded) return an_fail(ctx, SG_ERR_STATE, "no AudioNet model loaded (call sg_an_load)");
    AN_HIP(hipSetDevice(ctx->device));
    if (B < 1) return an_fail(ctx, SG_ERR_ARG, "B must be >= 1");
    if (flag != 0 && flag != 1) return an_fail(ctx, SG_ERR_ARG, "flag must be 0 (wav) or 1 (log-mel feat)");
    d->B = B;
    if (flag == 0) {
        if (TF < kAnFft) return an_fail(ctx, SG_ERR_ARG, "waveform shorter than one 1024-sample STFT frame");
        d->T = TF;
        d->F = an_num_frames(TF);
    } else {
        d->T = 0;
        d->F = TF;
    }
    int Tin[kAnConv], Tout[kAnConv];
    if (!an_layer_frames(d->F, Tin, Tout))
        return an_fail(ctx, SG_ERR_ARG, "%d frames are too few for the AudioNet stack (need >= 3 frames at conv8)", d->F);
    // 32-bit buffer offsets in the contraction kernels (k_conv_gemm.hip): every activation tensor < 2 GiB
    for (int l = 0; l < kAnConv; ++l)
        if ((size_t)B * Tout[l] * kAnCout[l] * sizeof(float) >= 0x80000000ull || (size_t)B * d->F * 32 * sizeof(float) >= 0x80000000ull)
            return an_fail(ctx, SG_ERR_ARG, "batch of %d x %d frames exceeds the 2 GiB per-tensor limit of one pass: split the batch",
                           B, d->F);
    int rc = an_build_tables(ctx);
    if (rc) return rc;
    rc = an_ensure_workspace(ctx, d->B, d->T, d->F);
    if (rc) return an_fail(ctx, rc, "workspace allocation failed: %s", ctx->err.c_str());
    return SG_OK;
}

const float* an_layer_input(const AnWorkspace& w, int l) {
    if (l == 0) return w.pre;
    return kAnPool[l - 1] ? w.pool[l - 1] : w.act[l - 1];
}

// Does the forward keep the frames' spectra for the adjoint of the same pass?  sg_an_configure: 1 / 0, or -1 = by size
// (profiles/r06_an_frontend_ab.txt, PGD-20, float32 transforms): the cache wins at every size measured (2-6 % of a step)
// except between 24 000 and 40 000 frames (80 .. 133 utterances of 3 s: -1 .. -6 %), where the chip is just full of
// forward waves and the cache's write traffic costs the forward more than the second transform costs the adjoint.
static bool an_use_spec_cache(const sg_ctx* ctx, int B, int F) {
    if (ctx->an_cfg.spec_cache >= 0) return ctx->an_cfg.spec_cache != 0;
    const long frames = (long)B * F;
    return !(frames >= 24000 && frames < 40000);
}

// waveform -> log-mel features in ws.feats (the backward of the same pass starts from the stored mel energies)
int an_frontend_forward(sg_ctx* ctx, const float* x, const AnDims& d, hipStream_t s) {
    AnWorkspace& w = ctx->an_ws;
    if (!d.keep_scale) AN_HIP(launch_input_scale(x, (int64_t)d.B * d.T, ctx->range_scratch, w.scale, 1, s));
    AnTables tab = ctx->an_tab;
    tab.mel_cache = w.mel_cache;
    if (an_use_spec_cache(ctx, d.B, d.F)) {
        // the cache is an optional speed-up: sized for this call (grown when a larger one comes), and a refused allocation
        // leaves the pass on the backward's own transforms instead of failing it
        const size_t need = (size_t)d.B * d.F * (kAnFft / 2) * sizeof(float2);
        if (need > w.spec_cache_bytes && !w.spec_cache_refused) {
            void* p = nullptr;
            if (hipMalloc(&p, need) == hipSuccess) {
                if (w.spec_cache) {
                    AN_HIP(hipStreamSynchronize(s));  // (an earlier pass may still read the old buffer)
                    (void)hipFree(w.spec_cache);
                    w.allocs.erase(std::remove(w.allocs.begin(), w.allocs.end(), static_cast<void*>(w.spec_cache)), w.allocs.end());
                }
                w.allocs.push_back(p);
                w.spec_cache = static_cast<float2*>(p);
                w.spec_cache_bytes = need;
            } else {
                (void)hipGetLastError();
                w.spec_cache_refused = true;
                fprintf(stderr, "speakerguard: no room for the %zu MB spectrum cache of the log-mel front-end; the adjoint transforms again\n", need >> 20);
            }
        }
        if (need <= w.spec_cache_bytes) tab.spec_cache = w.spec_cache;
    }
    w.cache_x = x; w.cache_B = d.B; w.cache_T = d.T; w.cache_spec = tab.spec_cache != nullptr;
    AN_STAGE(SG_STAGE_AN_LOGMEL_FWD, launch_an_logmel_fwd(tab, x, d.B, d.T, d.F, w.scale, w.feats, ctx->an_cfg.fft32, s));
    return SG_OK;
}

// features (B, Fnet, 32) -> conv stack; Fnet is the frame count the network sees (the front-end's, or the number of
// FeCo clusters when the defense sits between front-end and network)
// SG_AN_FUSED=0: the per-layer launch sequence of rounds 1-3 (the fused kernels' bit-exact counterpart: tests compare the
// two); SG_AN_SLICES=n: n time slices per utterance instead of the planner's choice.  Read per call (tests flip them).
bool an_use_fused(sg_ctx* ctx, int rows, int Fnet) {
    const char* e = sg_tune_env("SG_AN_FUSED");
    if (e && atoi(e) == 0) return false;
    const AnWorkspace& w = ctx->an_ws;
    return an_fused_supported(w.Tin, w.Tout, Fnet, rows, ctx->num_cus);
}
int an_forced_slices() {
    const char* e = sg_tune_env("SG_AN_SLICES");
    return e ? atoi(e) : 0;
}
AnFusedArgs an_fused_args(sg_ctx* ctx, int Fnet) {
    AnWorkspace& w = ctx->an_ws;
    const AnModel& m = ctx->an;
    AnFusedArgs a{};
    a.pre = w.pre;
    for (int l = 0; l < kAnConv; ++l) {
        a.act[l] = w.act[l];
        a.pool[l] = w.pool[l];
        a.bias[l] = m.bias[l];
        a.Tin[l] = w.Tin[l];
        a.Tout[l] = w.Tout[l];
    }
    a.w25 = m.w25;
    a.pre_bias = m.pre_bias;
    a.Fnet = Fnet;
    return a;
}

// Where the network's head runs (round 6).  Default: inside the fused backward launch whenever a gradient follows (every
// block computes its utterance's head: no an_tail launch, no d conv8 round trip) -- 3-5 % of a feature-level pass
// (tools/an_head_ab.py: 188 -> 182 us at 64 utterances, 702 -> 665 at 512), neutral inside the device loops
// (tools/an_head_loop_ab.py).  Forward + head + backward of whole utterances as ONE launch (S = 1) exists and is bit-equal,
// but measured 3.5 % SLOWER inside the PGD loop at 256 / 512 utterances (profiles/r06_an_head_ab.txt): off unless
// SG_AN_ONE=1.  SG_AN_HEAD=0 brings the separate an_tail launch back (both knobs behind SG_TUNE=1; the tests compare the
// three forms bit for bit).
static bool an_head_in_backward(sg_ctx* ctx, int rows, int Fnet) {
    const char* e = sg_tune_env("SG_AN_HEAD");
    if (e && atoi(e) == 0) return false;
    return an_use_fused(ctx, rows, Fnet);
}
static bool an_one_launch(sg_ctx* ctx, int rows, int Fnet) {
    const char* e = sg_tune_env("SG_AN_ONE");
    if (!e || atoi(e) == 0) return false;
    if (!an_head_in_backward(ctx, rows, Fnet)) return false;
    AnWorkspace& w = ctx->an_ws;
    an_layer_frames(Fnet, w.Tin, w.Tout);
    return an_fused_slices(w.Tin, w.Tout, Fnet, rows, ctx->num_cus, an_forced_slices()) == 1;
}
static AnHeadArgs an_head_args(sg_ctx* ctx, const int64_t* y, const sg_loss_spec& ls, int coef_rows, float* scores, int64_t* decisions,
                               float* loss, float* loss_trace, int64_t* dec_trace, uint8_t* success) {
    AnHeadArgs h{};
    h.on = 1;
    h.fc_w = ctx->an.fc_w;
    h.fc_b = ctx->an.fc_b;
    h.S = ctx->an.S;
    h.threshold = -INFINITY;
    h.y = y;
    h.ls = ls;
    h.coef_rows = coef_rows;
    h.scores_out = scores;
    h.dec_out = decisions;
    h.loss_out = loss;
    h.loss_trace = loss_trace;
    h.dec_trace = dec_trace;
    h.success = success;
    return h;
}

// features -> conv stack -> head -> d loss / d features, one launch (an_one_launch said yes)
int an_net_forward_backward(sg_ctx* ctx, const float* feats, int B, int Fnet, const AnHeadArgs& head, float* dfeats_out, hipStream_t s) {
    AnWorkspace& w = ctx->an_ws;
    const AnModel& m = ctx->an;
    an_layer_frames(Fnet, w.Tin, w.Tout);
    AnFusedArgs a = an_fused_args(ctx, Fnet);
    a.feats = feats;
    a.dfeats = dfeats_out;
    a.head = head;
    for (int l = 0; l < kAnConv; ++l) {
        a.wq[l] = m.wfq[l];
        a.wq_bwd[l] = m.wbq[l];
    }
    AN_STAGE(SG_STAGE_AN_FUSED_FWDBWD, launch_an_cnn_fwdbwd(a, B, ctx->num_cus, an_forced_slices(), s));
    return SG_OK;
}

int an_net_forward(sg_ctx* ctx, const float* feats, int B, int Fnet, hipStream_t s) {
    AnWorkspace& w = ctx->an_ws;
    const AnModel& m = ctx->an;
    an_layer_frames(Fnet, w.Tin, w.Tout);
    if (an_use_fused(ctx, B, Fnet)) {
        AnFusedArgs a = an_fused_args(ctx, Fnet);
        a.feats = feats;
        for (int l = 0; l < kAnConv; ++l) a.wq[l] = m.wfq[l];
        AN_STAGE(SG_STAGE_AN_FUSED_FWD, launch_an_cnn_fused(a, B, ctx->num_cus, false, an_forced_slices(), s));
        return SG_OK;
    }
    AN_STAGE(SG_STAGE_AN_PREFILTER_FWD, launch_an_prefilter(feats, w.pre, B, Fnet, m.w25, m.pre_bias, 0, s));
    for (int l = 0; l < kAnConv; ++l) {
        ConvGemmArgs a{};
        a.A = an_layer_input(w, l);
        a.W = m.wf[l];
        a.C = w.act[l];
        a.bias = m.bias[l];
        a.Ta = w.Tin[l];
        a.Tc = w.Tout[l];
        a.M = B * a.Tc;
        a.N = kAnCout[l];
        a.Kc = kAnCin[l];
        a.lda = kAnCin[l];
        a.ldw = kAnCout[l];
        a.ldc = kAnCout[l];
        a.taps = 3;
        a.tap_step = 1;
        a.tap_base = -kAnPad[l];
        a.total_chunks = 3 * (a.Kc / 32);
        a.chunks_per_split = a.total_chunks;
        a.Wq = a.N % 128 == 0 ? m.wfq[l] : nullptr;
        AN_STAGE(SG_STAGE_AN_CONV_FWD + l, launch_conv_gemm(a, a.N % 128 == 0 ? 2 : 1, EPI_BIAS_RELU, 1, s));
        if (kAnPool[l]) AN_STAGE(SG_STAGE_AN_POOL_FWD, launch_an_pool_fwd(w.act[l], w.pool[l], B, w.Tout[l], kAnCout[l], s));
    }
    return SG_OK;
}

int an_forward_net(sg_ctx* ctx, const float* x, const AnDims& d, int flag, hipStream_t s) {
    const float* feats = x;
    if (flag == 0) {
        int rc = an_frontend_forward(ctx, x, d, s);
        if (rc) return rc;
        feats = ctx->an_ws.feats;
    }
    return an_net_forward(ctx, feats, d.B, d.F, s);
}

// d loss / d conv8 pre-activation (ws.dact[6]) -> d loss / d features (B, Fnet, 32) in dfeats_out
int an_net_backward(sg_ctx* ctx, int B, int Fnet, float* dfeats_out, hipStream_t s, const AnHeadArgs* head = nullptr) {
    AnWorkspace& w = ctx->an_ws;
    const AnModel& m = ctx->an;
    if (an_use_fused(ctx, B, Fnet)) {
        AnFusedArgs a = an_fused_args(ctx, Fnet);
        a.dtop = w.dact[kAnConv - 1];
        a.dfeats = dfeats_out;
        if (head) a.head = *head;
        for (int l = 0; l < kAnConv; ++l) a.wq[l] = m.wbq[l];
        AN_STAGE(SG_STAGE_AN_FUSED_BWD, launch_an_cnn_fused(a, B, ctx->num_cus, true, an_forced_slices(), s));
        return SG_OK;
    }
    for (int l = kAnConv - 1; l >= 0; --l) {
        // data gradient of conv l: reads dact[l] (B, Tout, Cout), writes the gradient of its input
        const bool in_pooled = l > 0 && kAnPool[l - 1];
        ConvGemmArgs a{};
        a.A = w.dact[l];
        a.W = m.wb[l];
        a.C = l == 0 ? w.dpre : (in_pooled ? w.dpool[l - 1] : w.dact[l - 1]);
        a.mask = (l == 0 || in_pooled) ? nullptr : w.act[l - 1];
        a.Ta = w.Tout[l];
        a.Tc = w.Tin[l];
        a.M = B * a.Tc;
        a.N = kAnCin[l];
        a.Kc = kAnCout[l];
        a.lda = kAnCout[l];
        a.ldw = kAnCin[l];
        a.ldc = kAnCin[l];
        a.taps = 3;
        a.tap_step = -1;
        a.tap_base = kAnPad[l];  // d in[t] = sum_j W_j^T d out[t + pad - j]
        a.total_chunks = 3 * (a.Kc / 32);
        a.chunks_per_split = a.total_chunks;
        a.Wq = a.N % 128 == 0 ? m.wbq[l] : nullptr;
        AN_STAGE(SG_STAGE_AN_CONV_BWD + l, launch_conv_gemm(a, a.N % 128 == 0 ? 2 : 1, a.mask ? EPI_RELU_MASK : EPI_NONE, 1, s));
        if (in_pooled)
            AN_STAGE(SG_STAGE_AN_POOL_BWD, launch_an_pool_bwd(w.act[l - 1], w.dpool[l - 1], w.dact[l - 1], B, w.Tout[l - 1], kAnCout[l - 1], s));
    }
    AN_STAGE(SG_STAGE_AN_PREFILTER_BWD, launch_an_prefilter(w.dpre, dfeats_out, B, Fnet, m.w25, 0.f, 1, s));
    return SG_OK;
}

// the overlap-add inside the adjoint, or the separate pair?  (AnFrontCfg::ola: 1 / 0, or -1 = whichever the batch favours:
// the fused form cuts utterances into runs with 5 halo frames each, k_audionet.hip an_ola_pays)
static bool an_use_ola(const sg_ctx* ctx, int B, int F) {
    return ctx->an_cfg.ola > 0 || (ctx->an_cfg.ola < 0 && an_ola_pays(B, F, ctx->an_cfg.fft32, ctx->num_cus));
}

// d loss / d log-mel (B, F, 32) -> d loss / d waveform: written to grad_out and / or applied as the fused PGD update
// x_update: the iterate to step from (== x); x_next: where the stepped iterate goes.  The fused overlap-add needs
// x_next != x_update (neighbour blocks still read x around their cut); the separate pair updates in place and copies if
// the caller asked for another buffer.
int an_frontend_backward(sg_ctx* ctx, const float* x, const AnDims& d, const float* dfeats, float* grad_out, float* x_update, float* x_next,
                         const float* lower, const float* upper, float step, int grad_sign, hipStream_t s) {
    AnWorkspace& w = ctx->an_ws;
    AnTables tab = ctx->an_tab;
    tab.mel_cache = w.mel_cache;
    tab.spec_cache = w.cache_spec ? w.spec_cache : nullptr;
    if (an_use_ola(ctx, d.B, d.F) && (!x_update || x_next != x_update)) {
        AnOlaArgs a{};
        a.x = x; a.dfeats = dfeats; a.dframes = w.dframes; a.grad_out = grad_out;
        a.x_in = x_update; a.x_out = x_update ? x_next : nullptr; a.lower = lower; a.upper = upper; a.scale_p = w.scale;
        a.step = step; a.grad_sign = grad_sign; a.B = d.B; a.T = d.T; a.F = d.F;
        AN_STAGE(SG_STAGE_AN_LOGMEL_BWD, launch_an_logmel_bwd_ola(tab, a, ctx->an_cfg.fft32, ctx->num_cus, s));
        return SG_OK;
    }
    AN_STAGE(SG_STAGE_AN_LOGMEL_BWD, launch_an_logmel_bwd(tab, x, d.B, d.T, d.F, w.scale, dfeats, w.dframes, ctx->an_cfg.fft32, s));
    AN_STAGE(SG_STAGE_AN_OVERLAP_ADD, launch_an_frames_to_wave(w.dframes, d.B, d.T, d.F, w.scale, grad_out, x_update, lower, upper, step, grad_sign, s));
    if (x_update && x_next != x_update)
        AN_HIP(hipMemcpyAsync(x_next, x_update, (size_t)d.B * d.T * sizeof(float), hipMemcpyDeviceToDevice, s));
    return SG_OK;
}

// the buffer the fused overlap-add steps into (null: the separate pair updates in place)
float* an_step_target(sg_ctx* ctx, const AnDims& d) {
    AnWorkspace& w = ctx->an_ws;
    if (!an_use_ola(ctx, d.B, d.F)) return nullptr;
    if (!w.x_alt) {
        void* p = nullptr;
        if (hipMalloc(&p, (size_t)w.B * w.T * sizeof(float)) != hipSuccess) return nullptr;  // falls back to the separate pair
        w.allocs.push_back(p);
        w.x_alt = static_cast<float*>(p);
    }
    return w.x_alt;
}

int an_backward_net(sg_ctx* ctx, const float* x, const AnDims& d, int flag, float* grad_out, float* x_update, float* x_next,
                    const float* lower, const float* upper, float step, int grad_sign, hipStream_t s, const AnHeadArgs* head = nullptr,
                    bool net_done = false) {
    AnWorkspace& w = ctx->an_ws;
    int rc = net_done ? SG_OK : an_net_backward(ctx, d.B, d.F, flag == 1 ? grad_out : w.dfeats, s, head);
    if (rc || flag == 1) return rc;
    return an_frontend_backward(ctx, x, d, w.dfeats, grad_out, x_update, x_next, lower, upper, step, grad_sign, s);
}

}  // namespace

extern "C" {

int32_t sg_an_num_frames(int32_t T) { return an_num_frames(T); }

int sg_an_load(sg_ctx* ctx, const sg_an_weights* w) {
    if (!ctx || !w) return SG_ERR_ARG;
    if (w->num_class < 1 || w->num_class > 1024) return an_fail(ctx, SG_ERR_ARG, "num_class must be 1..1024");
    if (!w->conv1_weight || !w->conv1_bias || !w->fc_weight || !w->fc_bias) return an_fail(ctx, SG_ERR_ARG, "missing tensor");
    for (int i = 0; i < 4; ++i)
        if (!w->bn1[i]) return an_fail(ctx, SG_ERR_ARG, "missing conv1 BatchNorm tensor");
    for (int l = 0; l < kAnConv; ++l)
        if (!w->conv_weight[l] || !w->conv_bias[l] || !w->bn_weight[l] || !w->bn_bias[l] || !w->bn_mean[l] || !w->bn_var[l])
            return an_fail(ctx, SG_ERR_ARG, "missing tensor for conv%d", l + 2);
    AN_HIP(hipSetDevice(ctx->device));
    int rc = an_build_tables(ctx);
    if (rc) return rc;
    const double eps = w->bn_eps > 0.f ? w->bn_eps : 1e-5;
    AnModel& m = ctx->an;
    m = AnModel();
    std::vector<void*>& pool = ctx->model_allocs;
    // BatchNorm (eval, affine) directly follows every convolution and precedes the ReLU
    // (audionet_csine.py:68-115): y = g (conv + b - mean) / sqrt(var + eps) + beta folds into W, b.
    {
        const double g = w->bn1[0][0], beta = w->bn1[1][0], mean = w->bn1[2][0], var = w->bn1[3][0];
        const double sc = g / std::sqrt(var + eps);
        std::vector<float> w25(25);
        for (int i = 0; i < 25; ++i) w25[i] = (float)(w->conv1_weight[i] * sc);  // [mel offset][time offset]
        m.pre_bias = (float)((w->conv1_bias[0] - mean) * sc + beta);
        rc |= an_upload(ctx, pool, &m.w25, w25);
    }
    for (int l = 0; l < kAnConv; ++l) {
        const int cin = kAnCin[l], cout = kAnCout[l];
        std::vector<float> wf((size_t)3 * cin * cout), wb((size_t)3 * cout * cin), bias(cout);
        for (int co = 0; co < cout; ++co) {
            const double sc = w->bn_weight[l][co] / std::sqrt((double)w->bn_var[l][co] + eps);
            bias[co] = (float)((w->conv_bias[l][co] - w->bn_mean[l][co]) * sc + w->bn_bias[l][co]);
            for (int ci = 0; ci < cin; ++ci)
                for (int j = 0; j < 3; ++j) {
                    const float v = (float)(w->conv_weight[l][((size_t)co * cin + ci) * 3 + j] * sc);
                    wf[((size_t)j * cin + ci) * cout + co] = v;
                    wb[((size_t)j * cout + co) * cin + ci] = v;
                }
        }
        rc |= an_upload(ctx, pool, &m.wf[l], wf);
        rc |= an_upload(ctx, pool, &m.wb[l], wb);
        auto packed = [](const std::vector<float>& w, int K, int N) {  // [K][N] -> k4-major [K/4][N][4]
            std::vector<float> q((size_t)K * N);
            for (int k = 0; k < K; ++k)
                for (int n = 0; n < N; ++n) q[((size_t)(k / 4) * N + n) * 4 + (k & 3)] = w[(size_t)k * N + n];
            return q;
        };
        // (every layer: the fused CNN kernels take all their weights k4-packed; the per-layer sequence uses the packed copy
        // where its quad-fed tile kernel applies, N % 128 == 0)
        rc |= an_upload(ctx, pool, &m.wfq[l], packed(wf, 3 * cin, cout));
        rc |= an_upload(ctx, pool, &m.wbq[l], packed(wb, 3 * cout, cin));
        rc |= an_upload(ctx, pool, &m.bias[l], bias);
    }
    rc |= an_upload(ctx, pool, &m.fc_w, std::vector<float>(w->fc_weight, w->fc_weight + (size_t)w->num_class * 32));
    rc |= an_upload(ctx, pool, &m.fc_b, std::vector<float>(w->fc_bias, w->fc_bias + w->num_class));
    if (rc) return an_fail(ctx, SG_ERR_HIP, "model upload failed: %s", ctx->err.c_str());
    m.S = w->num_class;
    m.loaded = true;
    return SG_OK;
}

int sg_an_logmel(sg_ctx* ctx, const float* x_dev, int32_t B, int32_t T, float* feats_dev, void* stream) {
    if (!ctx || !x_dev || !feats_dev || B < 1 || T < kAnFft) return an_fail(ctx, SG_ERR_ARG, "bad argument");
    int rc = an_build_tables(ctx);
    if (rc) return rc;
    float* scale = nullptr;
    hipStream_t s = (hipStream_t)stream;
    AnTables tab = ctx->an_tab;
    if (ctx->an.loaded) {
        // with a model loaded the workspace is sized for (B, T) anyway: leave the mel energies there so that
        // sg_an_logmel_backward(reuse_forward) on the same input need not recompute the forward
        AnDims d;
        if ((rc = an_check(ctx, B, T, 0, &d))) return rc;
        AnWorkspace& w = ctx->an_ws;
        tab.mel_cache = w.mel_cache;
        w.cache_x = x_dev; w.cache_B = B; w.cache_T = T; w.cache_spec = false;
    } else if (!ctx->an_ws.scale) {
        rc = an_ensure_workspace(ctx, 1, kAnFft, an_num_frames(kAnFft) + 40);
        if (rc) return rc;
    }
    scale = ctx->an_ws.scale;
    AN_HIP(launch_input_scale(x_dev, (int64_t)B * T, ctx->range_scratch, scale, 1, s));
    AN_HIP(launch_an_logmel_fwd(tab, x_dev, B, T, an_num_frames(T), scale, feats_dev, ctx->an_cfg.fft32, s));
    return SG_OK;
}

int sg_an_logmel_backward(sg_ctx* ctx, const float* x_dev, int32_t B, int32_t T, const float* dfeats_dev, float* grad_dev,
                          int32_t reuse_forward, void* stream) {
    if (!x_dev || !dfeats_dev || !grad_dev) return an_fail(ctx, SG_ERR_ARG, "bad argument");
    AnDims d;
    int rc = an_check(ctx, B, T, 0, &d);  // sizes the per-frame gradient scratch
    if (rc) return rc;
    hipStream_t s = (hipStream_t)stream;
    AnWorkspace& w = ctx->an_ws;
    AnTables tab = ctx->an_tab;
    if (reuse_forward && w.cache_x == x_dev && w.cache_B == B && w.cache_T == T) {
        tab.mel_cache = w.mel_cache;
        tab.spec_cache = w.cache_spec ? w.spec_cache : nullptr;
    }
    AN_HIP(launch_input_scale(x_dev, (int64_t)B * T, ctx->range_scratch, w.scale, 1, s));
    if (tab.mel_cache && an_use_ola(ctx, d.B, d.F)) {  // the attack loops' form of the adjoint (same sums in the same order as the pair below)
        AnOlaArgs a{};
        a.x = x_dev; a.dfeats = dfeats_dev; a.dframes = w.dframes; a.grad_out = grad_dev; a.scale_p = w.scale;
        a.B = d.B; a.T = d.T; a.F = d.F;
        AN_HIP(launch_an_logmel_bwd_ola(tab, a, ctx->an_cfg.fft32, ctx->num_cus, s));
        return SG_OK;
    }
    AN_HIP(launch_an_logmel_bwd(tab, x_dev, d.B, d.T, d.F, w.scale, dfeats_dev, w.dframes, ctx->an_cfg.fft32, s));
    AN_HIP(launch_an_frames_to_wave(w.dframes, d.B, d.T, d.F, w.scale, grad_dev, nullptr, nullptr, nullptr, 0.f, 1, s));
    return SG_OK;
}

int sg_an_configure(sg_ctx* ctx, int32_t fft_bits, int32_t spectrum_cache, int32_t fused_overlap_add) {
    if (!ctx) return SG_ERR_ARG;
    if (fft_bits != 32 && fft_bits != 64) return an_fail(ctx, SG_ERR_ARG, "fft_bits must be 32 or 64");
    ctx->an_cfg.fft32 = fft_bits == 32;
    ctx->an_cfg.spec_cache = spectrum_cache < 0 ? -1 : (spectrum_cache != 0);
    ctx->an_cfg.ola = fused_overlap_add < 0 ? -1 : (fused_overlap_add != 0);
    ctx->an_ws.cache_x = nullptr;  // what an earlier forward left behind was computed under the old settings
    ctx->an_ws.cache_spec = false;
    return SG_OK;
}

int sg_an_forward(sg_ctx* ctx, const float* x_dev, int32_t B, int32_t T_or_F, int32_t flag, int64_t* decisions_dev,
                  float* scores_dev, float* emb_dev, void* stream) {
    AnDims d;
    int rc = an_check(ctx, B, T_or_F, flag, &d);
    if (rc) return rc;
    if (!x_dev) return an_fail(ctx, SG_ERR_ARG, "x is NULL");
    hipStream_t s = (hipStream_t)stream;
    if ((rc = an_forward_net(ctx, x_dev, d, flag, s))) return rc;
    AnWorkspace& w = ctx->an_ws;
    sg_loss_spec none{};
    AN_STAGE(SG_STAGE_AN_TAIL, launch_an_tail(w.act[kAnConv - 1], B, w.Tout[kAnConv - 1], ctx->an.fc_w, ctx->an.fc_b, ctx->an.S, -INFINITY, nullptr,
                          none, 0, emb_dev, scores_dev, decisions_dev, nullptr, nullptr, nullptr, nullptr, nullptr, s));
    return SG_OK;
}

int sg_an_debug_activation(sg_ctx* ctx, int32_t layer, float* out_dev, int64_t capacity_floats, int32_t* rows_per_utt,
                           int32_t* channels, void* stream) {
    if (!ctx || layer < 1 || layer > kAnConv + 1 || !ctx->an_ws.scale) return an_fail(ctx, SG_ERR_ARG, "bad layer or no pass run");
    const AnWorkspace& w = ctx->an_ws;
    const float* src;
    int rows, ch;
    if (layer == 1) { src = w.pre; rows = w.F >= 0 ? w.Tin[0] : 0; ch = kAnMel; }
    else {
        const int l = layer - 2;
        src = kAnPool[l] ? w.pool[l] : w.act[l];
        rows = kAnPool[l] ? w.Tout[l] / 2 : w.Tout[l];
        ch = kAnCout[l];
    }
    if (rows_per_utt) *rows_per_utt = rows;
    if (channels) *channels = ch;
    if (out_dev) {
        if (capacity_floats <= 0) return an_fail(ctx, SG_ERR_ARG, "capacity must be positive");
        const size_t held = (size_t)w.B * (size_t)(rows > 0 ? rows : 1) * ch;  // never read past what the workspace holds
        const size_t n = (size_t)capacity_floats < held ? (size_t)capacity_floats : held;
        AN_HIP(hipMemcpyAsync(out_dev, src, n * sizeof(float), hipMemcpyDeviceToDevice, (hipStream_t)stream));
    }
    return SG_OK;
}

int sg_an_loss_grad(sg_ctx* ctx, const float* x_dev, const int64_t* y_dev, int32_t B, int32_t T_or_F, int32_t flag,
                    const sg_loss_spec* loss, int64_t* decisions_dev, float* scores_dev, float* loss_dev, float* grad_dev,
                    void* stream) {
    AnDims d;
    int rc = an_check(ctx, B, T_or_F, flag, &d);
    if (rc) return rc;
    if (!x_dev || !y_dev || !loss) return an_fail(ctx, SG_ERR_ARG, "x, y and loss are required");
    if (loss->loss == SG_LOSS_LINEAR && !loss->coef_dev) return an_fail(ctx, SG_ERR_ARG, "SG_LOSS_LINEAR needs coef_dev");
    hipStream_t s = (hipStream_t)stream;
    AnWorkspace& w = ctx->an_ws;
    const int L = kAnConv - 1;
    if (grad_dev && an_head_in_backward(ctx, B, d.F)) {
        const AnHeadArgs head = an_head_args(ctx, y_dev, *loss, 0, scores_dev, decisions_dev, loss_dev, nullptr, nullptr, nullptr);
        if (an_one_launch(ctx, B, d.F)) {
            const float* feats = x_dev;
            if (flag == 0) {
                if ((rc = an_frontend_forward(ctx, x_dev, d, s))) return rc;
                feats = w.feats;
            }
            if ((rc = an_net_forward_backward(ctx, feats, B, d.F, head, flag == 1 ? grad_dev : w.dfeats, s))) return rc;
            return an_backward_net(ctx, x_dev, d, flag, grad_dev, nullptr, nullptr, nullptr, nullptr, 0.f, 0, s, nullptr, true);
        }
        if ((rc = an_forward_net(ctx, x_dev, d, flag, s))) return rc;
        return an_backward_net(ctx, x_dev, d, flag, grad_dev, nullptr, nullptr, nullptr, nullptr, 0.f, 0, s, &head);
    }
    if ((rc = an_forward_net(ctx, x_dev, d, flag, s))) return rc;
    AN_STAGE(SG_STAGE_AN_TAIL, launch_an_tail(w.act[L], B, w.Tout[L], ctx->an.fc_w, ctx->an.fc_b, ctx->an.S, -INFINITY, y_dev, *loss,
                          grad_dev != nullptr, nullptr, scores_dev, decisions_dev, loss_dev, w.dact[L], nullptr, nullptr,
                          nullptr, s));
    if (grad_dev) return an_backward_net(ctx, x_dev, d, flag, grad_dev, nullptr, nullptr, nullptr, nullptr, 0.f, 0, s);
    return SG_OK;
}

int sg_an_pgd_run(sg_ctx* ctx, float* x_adv_dev, const int64_t* y_dev, const float* lower_dev, const float* upper_dev,
                  int32_t B, int32_t T, const sg_pgd_params* p, uint8_t* success_dev, int64_t* decisions_dev,
                  float* scores_dev, float* loss_dev, float* loss_trace_dev, int64_t* decision_trace_dev, void* stream) {
    AnDims d;
    int rc = an_check(ctx, B, T, 0, &d);
    if (rc) return rc;
    if (!x_adv_dev || !y_dev || !lower_dev || !upper_dev || !p) return an_fail(ctx, SG_ERR_ARG, "NULL argument");
    if (p->max_iter < 0) return an_fail(ctx, SG_ERR_ARG, "max_iter must be >= 0");
    if (p->loss.loss == SG_LOSS_LINEAR && !p->loss.coef_dev) return an_fail(ctx, SG_ERR_ARG, "SG_LOSS_LINEAR needs coef_dev");
    hipStream_t s = (hipStream_t)stream;
    AnWorkspace& w = ctx->an_ws;
    const int L = kAnConv - 1;
    // the fused overlap-add steps from one waveform buffer into another: the iterate alternates between the caller's
    // buffer and a workspace twin and is copied home once if the attack ends on the twin
    float* xc = x_adv_dev;
    float* xn = an_step_target(ctx, d);
    if (!xn) xn = x_adv_dev;
    for (int it = 0; it <= p->max_iter; ++it) {
        const bool last = it == p->max_iter;
        d.keep_scale = it > 0;  // iterates stay in [-1, 1]
        if (!last && an_head_in_backward(ctx, B, d.F)) {
            // a gradient step: the head runs inside the backward launch (whole utterances per block: inside the one launch)
            const AnHeadArgs head = an_head_args(ctx, y_dev, p->loss, 0, nullptr, nullptr, nullptr,
                                                 loss_trace_dev ? loss_trace_dev + (size_t)it * B : nullptr,
                                                 decision_trace_dev ? decision_trace_dev + (size_t)it * B : nullptr, nullptr);
            if (an_one_launch(ctx, B, d.F)) {
                if ((rc = an_frontend_forward(ctx, xc, d, s))) return rc;
                if ((rc = an_net_forward_backward(ctx, w.feats, B, d.F, head, w.dfeats, s))) return rc;
                rc = an_backward_net(ctx, xc, d, 0, nullptr, xc, xn, lower_dev, upper_dev, p->step_size, p->grad_sign, s, nullptr, true);
            } else {
                if ((rc = an_forward_net(ctx, xc, d, 0, s))) return rc;
                rc = an_backward_net(ctx, xc, d, 0, nullptr, xc, xn, lower_dev, upper_dev, p->step_size, p->grad_sign, s, &head);
            }
            if (rc) return rc;
            if (xn != xc) std::swap(xc, xn);
            continue;
        }
        if ((rc = an_forward_net(ctx, xc, d, 0, s))) return rc;
        AN_STAGE(SG_STAGE_AN_TAIL, launch_an_tail(w.act[L], B, w.Tout[L], ctx->an.fc_w, ctx->an.fc_b, ctx->an.S, -INFINITY, y_dev, p->loss, !last,
                              nullptr, last ? scores_dev : nullptr, last ? decisions_dev : nullptr, last ? loss_dev : nullptr,
                              w.dact[L], loss_trace_dev ? loss_trace_dev + (size_t)it * B : nullptr,
                              decision_trace_dev ? decision_trace_dev + (size_t)it * B : nullptr,
                              last ? success_dev : nullptr, s));
        if (!last) {
            rc = an_backward_net(ctx, xc, d, 0, nullptr, xc, xn, lower_dev, upper_dev, p->step_size, p->grad_sign, s);
            if (rc) return rc;
            if (xn != xc) std::swap(xc, xn);
        }
    }
    if (xc != x_adv_dev) AN_HIP(hipMemcpyAsync(x_adv_dev, xc, (size_t)B * T * sizeof(float), hipMemcpyDeviceToDevice, s));
    return SG_OK;
}

int sg_an_pgd_run_feco(sg_ctx* ctx, float* x_adv_dev, const int64_t* y_dev, const float* lower_dev, const float* upper_dev,
                       int32_t B, int32_t T, const sg_pgd_params* p, const sg_feco_params* f, uint8_t* success_dev,
                       int64_t* decisions_dev, float* scores_dev, float* loss_dev, float* loss_trace_dev,
                       int64_t* decision_trace_dev, void* stream) {
    if (!ctx) return SG_ERR_ARG;
    if (!x_adv_dev || !y_dev || !lower_dev || !upper_dev || !p || !f) return an_fail(ctx, SG_ERR_ARG, "NULL argument");
    if (p->max_iter < 0) return an_fail(ctx, SG_ERR_ARG, "max_iter must be >= 0");
    if (p->loss.loss == SG_LOSS_LINEAR && !p->loss.coef_dev) return an_fail(ctx, SG_ERR_ARG, "SG_LOSS_LINEAR needs coef_dev");
    // defense/feature_level.py:33: with a single utterance the reference DROPS empty clusters (variable frame count);
    // that case stays on the host-chained path (model/defended_model.py)
    if (B < 2) return an_fail(ctx, SG_ERR_ARG, "the fused FeCo loop needs a batch of at least 2 utterances");
    const int eot_size = p->eot_size > 0 ? p->eot_size : 1, eot_bs = p->eot_batch_size > 0 ? p->eot_batch_size : 1;
    if (eot_size % eot_bs) return an_fail(ctx, SG_ERR_ARG, "EOT size should be divisible by EOT batch size");
    // Expectation over the defense's randomness (adaptive_attack/EOT.py:16-54): eot_size clusterings per gradient step,
    // each started from fresh random frames.  The reference repeats the batch (x_batch.repeat, EOT.py:24) and runs the
    // whole model on the copies; only the DEFENSE is random here, so the log-mel front-end runs once per step, the R
    // clusterings and the CNN run as one batch of R x B, and because the compression is linear in the features the
    // repeats' gradients are summed at the feature level (repeat order) and ONE log-mel adjoint + overlap-add takes the
    // sum to the waveform: sign(sum) == sign(mean).  (EOT_batch_size only says how the reference cuts the repeats into
    // model calls.)  The evenly started clustering is a deterministic function of its input: every repeat is the same
    // computation, one stands for all.  The final pass is a single forward.
    const int reps = f->random_init ? eot_size : 1;
    AnDims d;
    int rc = an_check(ctx, B * reps, T, 0, &d);  // workspace for the batch of R x B rows
    if (rc) return rc;
    d.B = B;
    const int k = f->k;
    int Tin[kAnConv], Tout[kAnConv];
    if (k < 1 || k > d.F || f->max_iter < 1 || !an_layer_frames(k, Tin, Tout))
        return an_fail(ctx, SG_ERR_ARG, "FeCo: need 1 <= k <= %d frames, enough of them for the AudioNet stack, max_iter >= 1", d.F);
    hipStream_t s = (hipStream_t)stream;
    AnWorkspace& w = ctx->an_ws;
    const int L = kAnConv - 1;
    for (int r = 0; r < reps; ++r)
        AN_HIP(hipMemcpyAsync(w.y_rep + (size_t)r * B, y_dev, (size_t)B * sizeof(int64_t), hipMemcpyDeviceToDevice, s));
    float* xc = x_adv_dev;  // waveform ping-pong of the fused overlap-add (sg_an_pgd_run)
    float* xn = an_step_target(ctx, d);
    if (!xn) xn = x_adv_dev;
    for (int it = 0; it <= p->max_iter; ++it) {
        const bool last = it == p->max_iter;
        const int R = last ? 1 : reps, rows = B * R;
        d.keep_scale = it > 0;  // iterates stay in [-1, 1]
        if ((rc = an_frontend_forward(ctx, xc, d, s))) return rc;
        const uint64_t key = f->seed + (uint64_t)it * 0x9E3779B97F4A7C15ull;  // repeat r: + r * 0xC2B2AE3D27D4EB4F
        trace_mark(ctx, SG_STAGE_AN_FECO_FWD, s, 0);
        rc = sg_feco_kmeans_compress(ctx, w.feats, B, d.F, kAnMel, k, f->max_iter, f->random_init, key, f->index_base, R,
                                     w.feco_ids, w.feco_out, w.feco_cnt, s);
        if (rc) return rc;
        trace_mark(ctx, SG_STAGE_AN_FECO_FWD, s, 1);
        // per-step records as the reference prints them (attack/FGSM.py:50-58): the loss averaged over the step's EOT
        // repeats, the decision voted over them
        const bool direct = R == 1, rec = loss_trace_dev || decision_trace_dev;
        float* ltr = !rec ? nullptr : (direct && loss_trace_dev ? loss_trace_dev + (size_t)it * B : w.trace_l);
        int64_t* dtr = !rec ? nullptr : (direct && decision_trace_dev ? decision_trace_dev + (size_t)it * B : w.trace_d);
        const bool head_in = !last && an_head_in_backward(ctx, rows, k);
        if (head_in) {  // the head inside the backward launch (or forward + head + backward as one launch)
            const AnHeadArgs head = an_head_args(ctx, w.y_rep, p->loss, B, nullptr, nullptr, nullptr, ltr, dtr, nullptr);
            if (an_one_launch(ctx, rows, k)) {
                if ((rc = an_net_forward_backward(ctx, w.feco_out, rows, k, head, w.dfeco, s))) return rc;
            } else {
                if ((rc = an_net_forward(ctx, w.feco_out, rows, k, s))) return rc;
                if ((rc = an_net_backward(ctx, rows, k, w.dfeco, s, &head))) return rc;
            }
        } else {
            if ((rc = an_net_forward(ctx, w.feco_out, rows, k, s))) return rc;
            AN_STAGE(SG_STAGE_AN_TAIL, launch_an_tail(w.act[L], rows, w.Tout[L], ctx->an.fc_w, ctx->an.fc_b, ctx->an.S, -INFINITY, w.y_rep, p->loss, !last,
                                  nullptr, last ? scores_dev : nullptr, last ? decisions_dev : nullptr, last ? loss_dev : nullptr,
                                  w.dact[L], ltr, dtr, last ? success_dev : nullptr, s, B));
        }
        if (rec && !direct)
            AN_HIP(launch_eot_trace_reduce(w.trace_l, w.trace_d, R, B, loss_trace_dev ? loss_trace_dev + (size_t)it * B : nullptr,
                                           decision_trace_dev ? decision_trace_dev + (size_t)it * B : nullptr, s));
        if (last) break;
        if (!head_in && (rc = an_net_backward(ctx, rows, k, w.dfeco, s))) return rc;
        trace_mark(ctx, SG_STAGE_AN_FECO_BWD, s, 0);
        if ((rc = sg_feco_compress_backward_reps(ctx, w.dfeco, w.feco_ids, w.feco_cnt, B, d.F, kAnMel, k, 1, R, w.dfeats, s))) return rc;
        trace_mark(ctx, SG_STAGE_AN_FECO_BWD, s, 1);
        rc = an_frontend_backward(ctx, xc, d, w.dfeats, nullptr, xc, xn, lower_dev, upper_dev, p->step_size, p->grad_sign, s);
        if (rc) return rc;
        if (xn != xc) std::swap(xc, xn);
    }
    if (xc != x_adv_dev) AN_HIP(hipMemcpyAsync(x_adv_dev, xc, (size_t)B * T * sizeof(float), hipMemcpyDeviceToDevice, s));
    return SG_OK;
}

}  // extern "C"
