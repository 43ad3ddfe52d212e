// Internal declarations shared by the HIP translation units (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdlib>
#include <stdint.h>
#include <string>
#include <vector>

#include "../../include/speakerguard_hip.h"

namespace sg {

// ---------------------------------------------------------------- front-end constants
// Kaldi MFCC as called at reference model/xv_plda.py:116-148.
constexpr int kShift = 160;     // frame_shift 10 ms
constexpr int kWin = 400;       // frame_length 25 ms
constexpr int kFft = 512;       // round_to_power_of_two
constexpr int kMel = 30;        // num_mel_bins
constexpr int kCep = 30;        // num_ceps
constexpr int kLdaLd = 516;     // row length of the device copy of the (D, 513) LDA matrix
constexpr int kFeatPad = 32;    // cepstra padded to the GEMM K granule
constexpr int kCmnWindow = 300; // iv_plda.py:310
constexpr float kEps = 1.1920928955078125e-07f;  // torch.finfo(float32).eps

// TDNN geometry, reference model/_xv_plda/xvecTDNN.py:16-33
constexpr int kLayers = 5;
constexpr int kCin[kLayers] = {30, 512, 512, 512, 512};
constexpr int kCout[kLayers] = {512, 512, 512, 512, 1500};
constexpr int kCinPad[kLayers] = {32, 512, 512, 512, 512};
constexpr int kCoutPad[kLayers] = {512, 512, 512, 512, 1536};
constexpr int kTaps[kLayers] = {5, 5, 7, 1, 1};
constexpr int kDil[kLayers] = {1, 2, 3, 1, 1};
constexpr int kEmb = 512;
constexpr int kPoolC = 1536;           // padded tdnn5 channels
constexpr int kStats = 2 * kPoolC;     // [mean | std], padded
constexpr int kFc1SplitK = 48;         // fc1 forward split-K (3072 / 64: two chunks per block)
constexpr int kFc1BwdSplitK = 8;          // (512 / 64: two chunks per block)
constexpr int kL1BwdSplitK = 10;       // tdnn1 data gradient: two K-slabs per tap (N = 32 gives only 19 tiles per 8 utterances: with one
                                       // slab per tap 95 blocks for 256 CUs, 19 -> 11 us at 8 utterances, 29 -> 24 at 32, same at 64).  A
                                       // constant of the arithmetic: the slabs are summed in order, whatever the batch

inline int num_frames(int T) { return (T + kShift / 2) / kShift; }

// ---------------------------------------------------------------- AudioNet CSI-NE (model/audionet_csine.py)
constexpr int kAnFft = 1024, kAnHop = 160, kAnWin = 800, kAnMel = 32, kAnBins = 513;  // Preprocessor.py:13-23
constexpr int kAnConv = 7;                                                              // conv2 .. conv8
constexpr int kAnCin[kAnConv] = {32, 64, 128, 128, 128, 128, 64};
constexpr int kAnCout[kAnConv] = {64, 128, 128, 128, 128, 64, 32};
constexpr int kAnPad[kAnConv] = {1, 1, 1, 1, 1, 1, 0};
constexpr bool kAnPool[kAnConv] = {true, false, false, true, false, true, false};
// torch.stft(center=True) on the pre-emphasised signal of T-1 samples: 1 + (T-1)/hop frames
inline int an_num_frames(int T) { return T < kAnWin ? 0 : 1 + (T - 1) / kAnHop; }

struct AnTables {            // device pointers
    float* window;           // [800] periodic hann
    float* mel_w;            // [32][513] slaney mel filterbank
    int* mel_lo;             // [32]
    int* mel_hi;             // [32]
    int* bin_m0;             // [513]
    float* bin_w0;           // [513]
    float* bin_w1;           // [513]
    double2* twiddle;        // [512] exp(-2 pi i k / 1024)
    uint16_t* bitrev;        // [1024]
    float* mel_cache;        // [B*F][32] forward -> backward hand-over within one pass (null: the backward recomputes)
    float2* spec_cache;      // [B*F][512] packed spectrum Z of every frame, same hand-over (null: the backward transforms again)
    // what a block's waves read lane by lane (k_audionet.hip AnLaneTab), laid out by the host: a block copies them to LDS
    float* lane_win;         // [16][64] window tap of FFT input 2 (lane + 64 i) + {0, 1} (0 outside the 800-tap window)
    float* lane_melw;        // [20][64] weights of the lane's run of a mel filter's bins, ascending, zero-padded
    int* lane_k0;            // [64] first bin of the lane's run
    int* mel_seg;            // [32] filter m: first lane | number of runs << 8 (an_build_tables)
    // the 512-point transform's twiddle tables as its lanes read them (fft512.h: tw1[(j - 1) * 64 + lane] = W512^(j lane),
    // tw2[9 b + c] = W64^(b c)), float64 values and the same rounded once to float32
    double2* tw1d;           // [448]
    double2* tw2d;           // [72]
    float2* tw1f;
    float2* tw2f;
};
// how the AudioNet front-end runs (sg_an_configure)
struct AnFrontCfg {
    // defaults = the fastest measured combination (profiles/r05_an_frontend_ab.txt)
    int fft32 = 1;       // transforms in float32 (the reference's precision) or float64
    int spec_cache = -1;  // the forward keeps every frame's packed spectrum for the backward of the same pass: 1 / 0, -1 = by size
    int ola = -1;        // overlap-add (+ update) inside the log-mel adjoint: 1 / 0, -1 = where it pays (an_ola_pays: large batches)
};
struct AnOlaArgs {
    const float* x;        // (B, T) waveform the frames are re-transformed from (spectrum cache: unused)
    const float* dfeats;   // (B, F, 32)
    float* dframes;        // (B, F, 800): only the edge frames are written (f < edge_lo or f >= edge_hi)
    float* grad_out;       // (B, T) or null
    const float* x_in;     // update: x_out = clamp(x_in + step * sign(g) * grad_sign); null: no update
    float* x_out;
    const float* lower;
    const float* upper;
    const float* scale_p;
    float step;
    int grad_sign;
    int B, T, F, S;        // S slices per utterance (chosen by the launcher)
    int edge_lo, edge_hi;  // frames the edge kernel needs (launcher)
    int t_lo, t_hi;        // d x[t] for t in [t_lo, t_hi] comes from the fused kernel, the rest from the edge kernel (launcher)
};

struct AnModel {
    bool loaded = false;
    int S = 0;
    float* w25 = nullptr;       // folded 5x5 pre-filter [mel offset][time offset]
    float pre_bias = 0.f;
    float* wf[kAnConv] = {};    // forward  [3*Cin][Cout]   (BatchNorm folded)
    float* wb[kAnConv] = {};    // backward [3*Cout][Cin]
    float* wfq[kAnConv] = {};   // k4-packed copies for the quad-fed tile kernel (layers whose N is a multiple of 128)
    float* wbq[kAnConv] = {};
    float* bias[kAnConv] = {};
    float* fc_w = nullptr;      // [S][32]
    float* fc_b = nullptr;      // [S]
};

struct AnWorkspace {
    int B = 0, T = 0, F = 0;
    int Tin[kAnConv] = {}, Tout[kAnConv] = {};  // frames entering / leaving each conv (before pooling)
    float* scale = nullptr;
    float* feats = nullptr;    // (B, F, 32) log-mel
    float* pre = nullptr;      // (B, F, 32) pre-filter output
    float* act[kAnConv] = {};  // ReLU outputs (B, Tout, Cout)
    float* pool[kAnConv] = {}; // pooled outputs where the layer has a MaxPool
    float* dact[kAnConv] = {}; // gradients wrt pre-activations
    float* dpool[kAnConv] = {};
    float* dpre = nullptr;     // (B, F, 32)
    float* dfeats = nullptr;   // (B, F, 32)
    float* dframes = nullptr;  // (B, F, 800)
    // FeCo inside the fused loop (sg_an_pgd_run_feco): cluster ids / sizes, compressed features and their gradient
    int* feco_ids = nullptr;     // (B, F)
    int* feco_cnt = nullptr;     // (B, F) (k <= F used)
    float* feco_out = nullptr;   // (B, k, 32)
    float* dfeco = nullptr;      // (B, k, 32)
    int64_t* y_rep = nullptr;    // (B) labels repeated for the EOT repeats batched into one pass
    float* trace_l = nullptr;    // (B * R) per-row loss / decision records of such a pass (reduced over the repeats afterwards)
    int64_t* trace_d = nullptr;
    float* mel_cache = nullptr;  // (B, F, 32) mel energies of the forward pass, kept for the backward of the same pass
    float2* spec_cache = nullptr;  // (B, F, 512) packed spectra of the forward pass (allocated on first use, grown on demand)
    size_t spec_cache_bytes = 0;
    bool spec_cache_refused = false;  // an allocation failed once: the passes run without the cache
    float* x_alt = nullptr;      // (B, T) the other half of the waveform ping-pong of the fused overlap-add update (on demand)
    // which input the mel cache belongs to (sg_an_logmel_backward(reuse_forward) checks pointer and shape, not contents)
    const float* cache_x = nullptr;
    int cache_B = 0, cache_T = 0;
    bool cache_spec = false;     // the spectrum cache holds that input's spectra too
    std::vector<void*> allocs;
};

// The MFCC kernels' constant tables exactly as they sit in a block's LDS (k_mfcc.hip), built once on the host per transform
// precision R and copied into LDS with 16-byte loads by every block (round 6; rounds 1-5 derived them per block from the raw
// tables -- dependent global look-ups in front of the first frame).
constexpr int kMfccTw1 = 7 * 64, kMfccTw2 = 72;  // = kFftTw1, kFftTw2 (fft512.h)
constexpr int kMelLaneBins = 24;  // >= bins per half mel filter (21 for 30 filters, 20-7600 Hz, 512-point FFT; host-checked)
template <typename R>
struct MfccLdsImage {
    R tw1[2 * kMfccTw1];                // (re, im) of W512^(j lane) at (j - 1) * 64 + lane, float64 values rounded once to R
    R tw2[2 * kMfccTw2];                // W64^(b c) at 9 b + c
    float window[kFft];                 // povey window, zero beyond sample 399
    float dct[kMel * kCep + 64];        // [m][c] (forward: lane c reads dct[m][c])
    float dct_t[kCep * 32];             // [c][m] (backward: lane m reads dct[m][c])
    float lifter[32];
    float melw_lane[kMelLaneBins * 64]; // [j][lane]: weight of bin mel_k0[lane] + j in the half filter of lane (0 beyond it)
    int mel_k0[64];                     // first bin of lane's half filter (two lanes per filter)
};
static_assert(sizeof(MfccLdsImage<float>) % 16 == 0 && sizeof(MfccLdsImage<double>) % 16 == 0, "copied as 16-byte words");

struct MfccTables {          // device pointers
    const MfccLdsImage<float>* lds_f32;
    const MfccLdsImage<double>* lds_f64;
    int* bin_m0;             // [256] lower mel index touching this bin (-1: none)   (the adjoint's per-bin mel membership)
    float* bin_w0;           // [256] weight into mel bin_m0
    float* bin_w1;           // [256] weight into mel bin_m0+1 (0 if none)
    int fft64;               // sg_xv_configure: 0 = float32 transforms (default, the reference's precision), 1 = float64
    // spectrum hand-over forward -> backward within one pass (null: the backward recomputes the forward):
    // bins 0..255 of every frame's FFT as float2, in the transform's register order (bin (l >> 3) + 8 (l & 7) + 64 d at
    // element 64 d + l), and the 30 mel energies
    float2* spec_cache;      // [B*F][256]
    float* mel_cache;        // [B*F][32]
    int rep_utts;            // > 0: rows are EOT repeats of rep_utts utterances (row = repeat * rep_utts + utterance):
                             // repeats share the waveform row, repeat r draws its dither from key seed + r * 0xC2B2AE3D27D4EB4F
};

struct XvModel {
    bool loaded = false;
    int D = 0, S = 0;
    float threshold = 0.f;
    float logdet_given = 0.f, logdet_without = 0.f;  // sum log(1 + psi/(psi+1)), sum log(psi+1)
    // folded TDNN weights, GEMM layouts
    float* wf[kLayers] = {};    // forward  [taps*CinPad][CoutPad]
    float* wb[kLayers] = {};    // backward [taps*CoutPad][CinPad]
    float* wfq[kLayers] = {};   // wf packed k4-major [taps*CinPad/4][CoutPad][4] (quad-fed GEMM)
    float* wbq[kLayers] = {};   // wb packed k4-major
    float* bias[kLayers] = {};  // [CoutPad] folded
    float* fc1_w = nullptr;     // [kStats][512] folded (K-major)
    float* fc1_wt = nullptr;    // [512][kStats] folded (for the backward GEMM)
    float* fc1_b = nullptr;     // [512] folded
    float* emb_mean = nullptr;  // [512]
    int Dp = 0;                 // D rounded up to a multiple of 4 (row stride of the transposed / PLDA matrices below)
    float* lda = nullptr;       // [D][kLdaLd]: the (D, 513) LDA matrix, rows padded to 516 floats (16-byte aligned rows)
    float* lda_t = nullptr;     // [513][Dp], zero-padded columns
    float* plda_mean = nullptr; // [D]
    float* plda_p = nullptr;    // [D][Dp], zero-padded columns
    float* plda_pt = nullptr;   // [D][Dp] transposed, zero-padded columns
    float* pa = nullptr;        // [D][kLdaLd]: (P A)[d][i], i < 512 -- the backward's P^T and LDA^T as one product (float64 product, rounded once)
    float* plda_psi = nullptr;  // [D]
    float* enroll = nullptr;    // [S][D]
    int enroll_cap = 0;         // speakers the enroll buffer holds (sg_xv_set_enroll reuses it)
    // per-call override of the enrolled set (iv_plda.py:155-165 enroll_embs=): a caller-owned device table that the
    // next passes score against; the model's own set is untouched
    const float* enroll_override = nullptr;
    int S_override = 0;
    std::vector<void*> allocs;  // device memory owned by this model (freed on reload / destroy)
};

struct Workspace {
    int B = 0, T = 0, F = 0;          // capacity the buffers were sized for
    int Fl[kLayers] = {};              // TDNN output frames per layer
    float* scale = nullptr;            // [1]
    float* feats_raw = nullptr;        // [B][F][30]
    float* feats = nullptr;            // [B][F][32] CMVN output, zero padded
    float* act[kLayers] = {};          // relu outputs [B][Fl][CoutPad]
    float* dact[kLayers] = {};         // d loss / d pre-activation [B][Fl][CoutPad]
    float* dfeats = nullptr;           // [kL1BwdSplitK][B][F][32] split-K slabs of the tdnn1 data gradient
    float* dfeats_raw = nullptr;       // [B][F][30]
    float* dframes = nullptr;          // [B][F][400]
    float2* spec_cache = nullptr;      // [B][F][256] forward spectrum kept for the backward (39 MB at B = 64)
    float* mel_cache = nullptr;        // [B][F][32]
    float* stats = nullptr;            // [B][kStats]
    float* fc1_part = nullptr;         // [kFc1SplitK][B][512]
    float* demb = nullptr;             // [B][512]
    float* dstats_part = nullptr;      // [kFc1BwdSplitK][B][kStats]
    float* tdnn_emb = nullptr;         // [B][512]
    float* emb = nullptr;              // [B][D]
    float* scores = nullptr;           // [B][S]
    float* loss = nullptr;             // [B]
    int64_t* decisions = nullptr;      // [B]
    float* grad = nullptr;             // [B][T]
    int64_t* y_rep = nullptr;          // [B] labels repeated for EOT repeats batched into one pass
    float* eot_loss_rows = nullptr;    // [reps * B] per-repeat records of a step whose repeats run as several passes
    int64_t* eot_dec_rows = nullptr;
    size_t eot_rows_cap = 0;
    std::vector<void*> allocs;
};

}  // namespace sg

struct sg_ctx {
    int device = 0;
    int num_cus = 256;               // compute units of the device: the stream-K launches are sized to it
    std::string err;
    sg::MfccTables tab{};
    bool tables_ready = false;
    float* range_scratch = nullptr;  // [512] partial max/min of check_input_range
    float* cw2_scratch = nullptr;    // [1024][32] loss2 partials
    float* sk_slabs = nullptr;       // stream-K scratch (k_conv_gemm.hip)
    unsigned* sk_flags = nullptr;
    sg::XvModel xv;
    sg::Workspace ws;
    // Health word: host-pinned, device-mapped.  A kernel that gives up on a stream-K hand-off (bounded spin) ORs a
    // bit into it; every pass entry point and sg_sync read the host side and fail loudly (no synchronisation needed).
    unsigned* err_host = nullptr;
    unsigned* err_dev = nullptr;
    bool use_streamk = true;         // sg_set_streamk
    int lose_handoffs = 0;           // sg_debug_lose_handoffs: stream-K launches left that publish no hand-off flags
    int lose_feco = 0;               // ... and paired k-means launches left whose second halves publish nothing (own budget)
    // FeCo k-means over two CUs per instance (k_feco.hip): exchange buffers + flags, the launch counter the flag values
    // are derived from, and the switch (sg_feco_set_two_cu: -1 where it fits, 0 never)
    unsigned long long* feco_xchg = nullptr;
    unsigned* feco_flags = nullptr;
    unsigned feco_epoch = 0;
    int feco_two_cu = -1;
    sg::AnTables an_tab{};
    sg::AnFrontCfg an_cfg{};
    bool an_tables_ready = false;
    sg::AnModel an;
    sg::AnWorkspace an_ws;
    std::vector<void*> model_allocs;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    // Stage trace (sg_trace_begin / sg_trace_end): while on, every launch of the x-vector pass sequences is bracketed by a
    // pair of HIP events on the launch stream; a record = (stage tag, event before, event after).
    bool trace_on = false;
    bool trace_open = false;   // the opening event of record trace_used was recorded, the closing one not yet
    int trace_dropped = 0;     // records given up because an event record failed (reported by sg_trace_end)
    int trace_used = 0;
    std::vector<hipEvent_t> trace_ev;  // 2 per record
    std::vector<int> trace_tag;
};

namespace sg {

// stage trace: event pair around one launch (no-op unless sg_trace_begin switched it on and records are left).  A
// record whose opening or closing event could not be recorded is dropped -- its slot is reused by the next launch --
// and counted; sg_trace_end reports the count instead of a later, unrelated-looking HIP error.  A launch error
// between the two marks leaves the record open; the next opening mark simply overwrites it.
inline void trace_mark(sg_ctx* ctx, int tag, hipStream_t s, int after) {
    if (!ctx->trace_on || ctx->trace_used >= (int)ctx->trace_tag.size()) return;
    if (!after) {
        ctx->trace_tag[ctx->trace_used] = tag;
        ctx->trace_open = hipEventRecord(ctx->trace_ev[2 * ctx->trace_used], s) == hipSuccess;
        if (!ctx->trace_open) ++ctx->trace_dropped;
    } else {
        if (!ctx->trace_open || ctx->trace_tag[ctx->trace_used] != tag) return;
        ctx->trace_open = false;
        if (hipEventRecord(ctx->trace_ev[2 * ctx->trace_used + 1], s) == hipSuccess) ++ctx->trace_used;
        else ++ctx->trace_dropped;
    }
}


// ---------------------------------------------------------------- kernel launchers (k_*.hip)
enum Epilogue { EPI_NONE = 0, EPI_BIAS_RELU = 1, EPI_RELU_MASK = 2 };

struct ConvGemmArgs {
    const float* A;     // activations [B*Ta][lda]
    const float* W;     // weights [taps*Kc][ldw]
    const float* Wq;    // same weights packed k4-major [taps*Kc/4][ldw][4] (null: b32-fed kernels only)
    unsigned a_bytes, w_bytes;  // extents of A and W for the buffer descriptors (filled by launch_conv_gemm)
    float* C;           // output [B*Tc][ldc] (+ z*split_stride for split-K partials)
    const float* bias;  // [N]            (EPI_BIAS_RELU)
    const float* mask;  // [B*Tc][ldc]    (EPI_RELU_MASK: keep where mask > 0)
    int M, N;           // M = B*Tc rows, N multiple of the tile width
    int Ta, Tc;         // rows per utterance in A / C
    int Kc;             // K per tap, multiple of 32
    int lda, ldw, ldc;
    int taps, tap_step; // A row offset of tap j = tap_base + j * tap_step
    int tap_base;
    int total_chunks, chunks_per_split;
    long long split_stride;
    int sk_xcd;         // stream-K: XCD-contiguous workers + n-tile-major tile order (see kernel)
    int num_cus;        // stream-K: persistent blocks = resident slots of THIS device (0: assume 256)
    unsigned long long* trace;  // tuning aid (SG_SK_TRACE): per-worker phase timestamps, 16 slots each, or null
    int force;          // 0 auto, 1 one b32-fed block per tile, 2 stream-K b32-fed 8-wave, 3 stream-K quad-fed 8-wave,
                        // 4 one quad-fed block per tile, 5 one 16 x 16 block per wave, 6 / 7 / 8 / 9 stream-K with the roles split
                        // between waves, 128- / 64- / 32- / 256-row tiles, 10 = 128-row tiles as four 64 x 64 computing waves
                        // (kind 9) (parity tests)
    int ablate;         // 8 = fault injection (sg_debug_lose_handoffs): stream-K hand-off flags are never published;
                        // builds with -DSG_EXP_ABLATE only (never the shipped library), from SG_ABLATE: 1 no global loads,
                        // 2 no LDS stores, 4 no barrier (results become wrong), 16 = A/B agent-scope fences around the hand-off
    int no_streamk;     // one block per tile whatever the shape (sg_set_streamk(ctx, 0): same bits, no co-residency needed)
    float* sk_slabs;    // stream-K: [768][64*128] parked partial tiles (may be null -> tile launch)
    unsigned* sk_flags; // stream-K: [768] hand-off flags
    unsigned* err_word; // stream-K: device-visible health word (bit 0 = a hand-off wait timed out), may be null
    int* lose_counter;  // HOST pointer (never dereferenced on the device): the context's fault-injection budget, may be null
};

// A process may hold one sg_ctx per GPU: whatever a launcher remembers between calls -- the > 64 KB dynamic-LDS opt-in of a
// kernel, a tuning aid's device buffer -- is remembered PER DEVICE (index = the current device, which every entry point
// sets from its context before launching).
// Tuning, tracing and test knobs are environment variables that count ONLY when SG_TUNE=1 is set as well (INTEGRATION.md,
// "Environment"): without it the library runs its own choices and a stray SG_* variable in a user's environment changes
// nothing.  The gate is read once per process; the knobs keep their own read-once / read-per-call behaviour.
inline const char* sg_tune_env(const char* name) {
    static const bool on = [] {
        const char* e = getenv("SG_TUNE");
        return e && atoi(e) != 0;
    }();
    return on ? getenv(name) : nullptr;
}

constexpr int kMaxDevices = 64;
inline int sg_device_slot() {
    int d = 0;
    (void)hipGetDevice(&d);
    return d & (kMaxDevices - 1);
}
// a tuning aid's scratch buffer on the current device (allocated on first use, grown on demand, never freed: tuning aids only)
struct PerDeviceScratch {
    void* ptr[kMaxDevices] = {};
    size_t cap[kMaxDevices] = {};
    void* get(size_t bytes) {
        const int d = sg_device_slot();
        if (cap[d] < bytes) {
            if (ptr[d]) (void)hipFree(ptr[d]);
            ptr[d] = nullptr;
            cap[d] = 0;
            if (hipMalloc(&ptr[d], bytes) == hipSuccess) cap[d] = bytes;
        }
        return ptr[d];
    }
};

// the launch-side bookkeeping every contraction of a context carries
inline void conv_ctx_args(sg_ctx* ctx, ConvGemmArgs& a) {
    a.sk_slabs = ctx->sk_slabs;
    a.sk_flags = ctx->sk_flags;
    a.err_word = ctx->err_dev;
    a.num_cus = ctx->num_cus;
    a.no_streamk = ctx->use_streamk ? 0 : 1;
    a.ablate = 0;
    // test hook (sg_debug_lose_handoffs): launch_streamk takes a launch off this counter when it actually launches a
    // stream-K kernel -- launches that end up on tile kernels (tdnn1, split-K, forced kinds) do not use the budget up
    a.lose_counter = ctx->use_streamk ? &ctx->lose_handoffs : nullptr;
}

// tile: 0 = auto (stream-K 128x128 8-wave blocks when the shape qualifies, else 64x128), 1 = 128x32 (4x1 waves),
//       2 = 64x128 (2x2 waves), one block per tile
hipError_t launch_conv_gemm(const ConvGemmArgs& a, int tile, int epi, int splits, hipStream_t s);
int conv_gemm_tile_rows(int M, int N);
// [K][N] row-major -> k4-major [K/4][N][4] on the device (K % 4 == 0)
hipError_t launch_pack_k4(const float* w, int K, int N, float* wq, hipStream_t s);  // tile height launch_conv_gemm(tile 0) picks

// mode 0: xv_plda ('origin': [-1,1] -> x32768), mode 1: AudioNet ('scale': int16 range -> /32768)
hipError_t launch_input_scale(const float* x, int64_t n, float* scratch512, float* scale, int mode, hipStream_t s);
hipError_t launch_mfcc_fwd(const MfccTables& t, const float* x, int B, int T, int F, const float* scale,
                           const sg_dither* dz, float* feats, hipStream_t s);
hipError_t launch_mfcc_bwd(const MfccTables& t, const float* x, int B, int T, int F, const float* scale,
                           const sg_dither* dz, const float* dfeats, float* dframes, hipStream_t s);
// overlap-add of dframes into d loss / d x (+ acc_in, the sum over earlier EOT repeats; may alias grad_out);
// optional fused PGD update of x (in place)
// dframes (R, B, F, 400): R batched EOT repeats, summed in repeat order
hipError_t launch_frames_to_wave(const float* dframes, int B, int T, int F, int R, const float* acc_in, float* grad_out,
                                 float* x_io, const float* lower, const float* upper, float step, int grad_sign,
                                 hipStream_t s);
hipError_t launch_pgd_update(float* x, const float* g, const float* lo, const float* hi, int64_t n,
                             float step, int grad_sign, hipStream_t s);
// out[r][0..ncol) = in[r][0..ncol), out[r][ncol..ld_out) = 0
hipError_t launch_copy_cols(const float* in, int ld_in, float* out, int ld_out, int64_t rows, int ncol, hipStream_t s);
hipError_t launch_sum_cols(const float* in, int ld_in, int nsplit, long long slab_stride, float* out, int ld_out,
                           int64_t rows, int ncol, hipStream_t s);
hipError_t launch_cmvn_fwd(const float* in, int ld_in, float* out, int ld_out, int B, int F, hipStream_t s);
hipError_t launch_cmvn_bwd(const float* dout, int ld_dout, int nsplit, long long slab_stride, float* din, int ld_din,
                           int B, int F, hipStream_t s);
hipError_t launch_pool_fwd(const float* act5, int B, int Tc, float* stats, hipStream_t s);
hipError_t launch_pool_bwd(const float* act5, const float* stats, const float* dstats_part, int nsplit,
                           int B, int Tc, float* dact5, hipStream_t s);

hipError_t launch_cw2_step(float* modifier, float* exp_avg, float* exp_avg_sq, const float* x, const float* input_cur,
                           const float* grad1, const float* const_c, int B, int T, float lr, int step_t,
                           float* input_next, float* loss2, float* scratch, hipStream_t s);
hipError_t launch_nes_queries(const float* x, int n, int T, int half, int with_clean, float sigma, uint64_t seed,
                              int64_t index_base, int pair_base, const float* noise_in, float* queries, float* noise_out,
                              hipStream_t s);
hipError_t launch_nes_grad(const float* loss, int n, int T, int half, int with_clean, uint64_t seed, int64_t index_base,
                           int pair_base, const float* noise_in, int accumulate, float final_sigma, int final_batches,
                           float* grad, hipStream_t s);
hipError_t launch_fakebob_step(float* x, float* grad, const float* prev_grad, const float* lr, const float* lower,
                               const float* upper, int n, int T, float momentum, float one_m_momentum, int grad_sign,
                               hipStream_t s);

hipError_t launch_loss_eval(const float* scores, const int64_t* y, int B, int S, float threshold, const sg_loss_spec& ls,
                            int64_t* dec, float* loss, float* dscores, hipStream_t s);
// per-step records of a step that ran R EOT repeats (rows r * B + b): loss_out[b] = mean over the repeats, dec_out[b] = the
// majority vote with first-seen tie-break (attack/FGSM.py:50-58, attack/utils.py:118-125 Counter.most_common)
hipError_t launch_eot_trace_reduce(const float* loss_rows, const int64_t* dec_rows, int R, int B, float* loss_out,
                                   int64_t* dec_out, hipStream_t s);

// The network's head (max over time, fc, decision, loss, d loss / d conv8: what an_tail_kernel does in a launch of its own)
// inside the fused backward launch: every block computes it for its utterance from act[6] -- a few microseconds of a
// block's ~80-160 -- and slice 0 writes the per-utterance outputs.  Same arithmetic in the same order as an_tail_kernel.
struct AnHeadArgs {
    int on;                    // 0: d loss / d conv8 comes from AnFusedArgs::dtop
    const float* fc_w;         // (S, 32)
    const float* fc_b;
    int S;
    float threshold;
    const int64_t* y;          // (rows)
    sg_loss_spec ls;
    int coef_rows;
    float* emb_out; float* scores_out; int64_t* dec_out; float* loss_out;   // optional, per row
    float* loss_trace; int64_t* dec_trace; uint8_t* success;                 // optional, per row
};

// the AudioNet CNN of one pass in one launch per direction (k_audionet_fused.hip), or -- whole utterances per block, S = 1 --
// forward, head and backward in ONE launch (launch_an_cnn_fwdbwd)
struct AnFusedArgs {
    // forward
    const float* feats;        // (rows, Fnet, 32)
    float* pre;                // (rows, Fnet, 32)
    float* act[kAnConv];       // (rows, Tout, Cout) ReLU outputs
    float* pool[kAnConv];      // (rows, Tout / 2, Cout) where the block has a MaxPool
    const float* wq[kAnConv];  // k4-packed weights of the direction: [3 K / 4][N][4]
    const float* wq_bwd[kAnConv];  // the data gradients' weights when one launch runs both directions
    const float* bias[kAnConv];
    const float* w25;
    float pre_bias;
    // backward
    const float* dtop;         // (rows, Tout[6], 32) d loss / d conv8 pre-activation
    float* dfeats;             // (rows, Fnet, 32)
    int Fnet, S, buf_floats;
    int Tin[kAnConv], Tout[kAnConv];
    unsigned long long* trace;  // tuning aid (SG_AN_TRACE): per block 16 timestamps (100 MHz) at the stage boundaries, or null
    AnHeadArgs head;
};
// false: the utterance is too long for the LDS-resident form even in its finest cut (the per-layer sequence runs)
bool an_fused_supported(const int* Tin, const int* Tout, int Fnet, int rows, int num_cus);
// force_slices > 0: that many time slices per utterance instead of the planner's choice (tests: same bits for any cut)
hipError_t launch_an_cnn_fused(AnFusedArgs a, int rows, int num_cus, bool backward, int force_slices, hipStream_t s);
// forward + head + backward of whole utterances in one launch; hipErrorNotSupported unless the planner's cut is S = 1
hipError_t launch_an_cnn_fwdbwd(AnFusedArgs a, int rows, int num_cus, int force_slices, hipStream_t s);
int an_fused_slices(const int* Tin, const int* Tout, int Fnet, int rows, int num_cus, int force_slices);  // the cut the fused launches use (0: not supported)
hipError_t launch_an_logmel_fwd(const AnTables& t, const float* x, int B, int T, int F, const float* scale, float* feats,
                                int fft32, hipStream_t s);
hipError_t launch_an_logmel_bwd(const AnTables& t, const float* x, int B, int T, int F, const float* scale,
                                const float* dfeats, float* dframes, int fft32, hipStream_t s);
// the adjoint with the overlap-add (+ update) inside: needs t.mel_cache of the same pass; x_out != x_in
hipError_t launch_an_logmel_bwd_ola(const AnTables& t, AnOlaArgs a, int fft32, int num_cus, hipStream_t s);
bool an_ola_pays(int B, int F, int fft32, int num_cus);
hipError_t launch_an_frames_to_wave(const float* dframes, int B, int T, int F, const float* scale, float* grad_out,
                                    float* x_io, const float* lower, const float* upper, float step, int grad_sign,
                                    hipStream_t s);
hipError_t launch_an_prefilter(const float* in, float* out, int B, int T, const float* w25, float bias, int transpose,
                               hipStream_t s);
hipError_t launch_an_pool_fwd(const float* in, float* out, int B, int Tin, int C, hipStream_t s);
hipError_t launch_an_pool_bwd(const float* act, const float* dpool, float* dact, int B, int Tin, int C, hipStream_t s);
hipError_t launch_an_tail(const float* act8, int B, int T8, const float* fc_w, const float* fc_b, int S, float threshold,
                          const int64_t* y, const sg_loss_spec& ls, int want_grad, float* emb, float* scores,
                          int64_t* decisions, float* loss, float* dact8, float* loss_trace, int64_t* dec_trace,
                          uint8_t* success, hipStream_t s, int coef_rows = 0);

struct TailArgs {
    const float* fc1_part; int nsplit; int B;
    const XvModel* m;  // host struct with device pointers (copied by value into the launch)
    const int64_t* y; sg_loss_spec loss; int want_grad;
    float* tdnn_emb; float* emb; float* scores; int64_t* decisions; float* loss_out; float* demb;
    // optional per-pass records for the fused loop
    float* loss_trace; int64_t* decision_trace; uint8_t* success;
    int coef_rows;  // SG_LOSS_LINEAR: rows of the caller's coef table (0: B); rows of batched EOT repeats wrap (row % coef_rows)
};
hipError_t launch_tail(const TailArgs& a, hipStream_t s);

}  // namespace sg
