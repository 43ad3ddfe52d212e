/*
 * speakerguard_hip.h -- C-ABI of the MI355X (gfx950) attack hot path.
 *
 * The reference (SpeakerGuard, pure Python/PyTorch) has no native boundary of its own; this is
 * the boundary a maintainer would bind with ctypes.  Every entry point
 * names the reference interface it stands in for (file:line under /root/reference).
 *
 * Conventions
 *   - plain C, no torch types; every pointer marked "dev" is a DEVICE pointer owned by the
 *     caller (e.g. tensor.data_ptr()); pointers marked "host" are host memory.
 *   - the library owns only the workspace inside sg_ctx (grown on demand, freed by sg_destroy).
 *   - every call returns 0 on success, non-zero on error; sg_last_error(ctx) gives the text.
 *     Nothing throws across the ABI.
 *   - one sg_ctx per device, not thread-safe.  Work is enqueued on `stream` (a hipStream_t passed
 *     as void*; NULL = the default stream) and is asynchronous unless stated otherwise.
 *   - all floating point is IEEE fp32 (f32-input MFMA); labels / decisions are int64.
 *
 * Tensor shapes use the reference's names: B utterances, T samples, F frames (snip_edges=False:
 * F = (T + 80) / 160), 30 cepstra, D = back-end dimension (rows of transform.txt), S = enrolled
 * speakers.
 */
#ifndef SPEAKERGUARD_HIP_H
#define SPEAKERGUARD_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct sg_ctx sg_ctx;

#define SG_OK 0
#define SG_ERR_ARG 1
#define SG_ERR_HIP 2
#define SG_ERR_STATE 3

/* ---- context ------------------------------------------------------------------------------ */
int sg_version(void);
int sg_create(int device, sg_ctx** out);
void sg_destroy(sg_ctx* ctx);
const char* sg_last_error(const sg_ctx* ctx);
/* blocks until everything enqueued on `stream` by this ctx has finished, then reports sg_health() */
int sg_sync(sg_ctx* ctx, void* stream);
/* SG_ERR_HIP if a kernel of an earlier launch raised the context's health word (a stream-K hand-off wait that timed
 * out: the contraction's output, and everything computed from it, is invalid); clears the word.  Does not
 * synchronise -- call it after the results were awaited (sg_sync does both).  Every pass entry point checks it too. */
int sg_health(sg_ctx* ctx);
/* The large contractions (TDNN layers 2-5, both directions) run as "stream-K" launches: one persistent block per compute
 * unit, a tile that is split between two blocks handed over through device memory.  A waiting block spins on its
 * predecessor, so ALL blocks of a launch must be resident at once: the context needs the GPU TO ITSELF while a pass runs
 * (no other process or context computing on the same device).  On a shared GPU a hand-off wait can time out; that is
 * reported, never silent (sg_health), and the caller's remedy is sg_set_streamk(ctx, 0): every contraction then runs as
 * one block per tile -- the same fused multiply-add chain per output, bit-identical results, a few percent slower at
 * batch 64 -- with no residency requirement.  speakerguard_amd's attack drivers do exactly that once, automatically
 * (attack/FGSM.py _run_batches), before they give up.  Default: enabled. */
int sg_set_streamk(sg_ctx* ctx, int32_t enable);
/* TEST HOOK (fault injection for the health path; no production use): the next `launches` stream-K launches of this
 * context publish no hand-off flags, so their waiting blocks time out (after a shortened bound) and raise the health word.
 * Only launches that really run a stream-K kernel count (a contraction that ends up on a tile kernel does not use the
 * budget up).  The two-unit k-means protocol below has a budget of its own, set to the same number by the same call. */
int sg_debug_lose_handoffs(sg_ctx* ctx, int32_t launches);
/* The FeCo k-means (sg_feco_kmeans*) runs an instance on TWO compute units when the batch leaves room for it (2 x instances
 * <= compute units): both blocks compute the same clustering and share only the assignment step's work through device
 * memory, every wait bounded (20 us) -- a block whose partner does not show up (GPU shared with something else) computes
 * everything itself: same bits, the one-unit speed.  mode -1: where it fits (default); 0: never.
 * (sg_debug_lose_handoffs also arms this protocol, with its own count: a paired launch it counts makes every second block
 * publish nothing.) */
int sg_feco_set_two_cu(sg_ctx* ctx, int32_t mode);
/* TEST HOOK (no production use): parks the launch counter of the two-unit k-means protocol, so that a test reaches the wrap
 * of the exchange words' 15-bit launch tag (and of the 32-bit counter itself) in a handful of launches instead of 2^15.  The
 * counter only moves the way real launches move it from there on -- the wipe of the exchange buffer at a tag wrap included. */
int sg_debug_feco_epoch(sg_ctx* ctx, uint32_t epoch);

/* How the x-vector front-end's 512-point transforms run (forward and adjoint): fft_bits 32 (default) = float32, the
 * reference's own precision (torchaudio 0.6's kaldi.mfcc is float32 end to end, model/xv_plda.py:114-148); 64 = float64
 * transforms around the same float32 stages (the form of rounds 1-5, kept as the counterpart).  Takes effect from the next
 * pass; results of the two differ by float32 round-off of the spectrum. */
int sg_xv_configure(sg_ctx* ctx, int32_t fft_bits);

/* ---- x-vector + PLDA model ----------------------------------------------------------------
 * Replaces the tensors the reference holds after model/xv_plda.py:17-47 ran: the xvecTDNN
 * state_dict (model/_xv_plda/xvecTDNN.py:16-44), emb_mean (model/utils.py:50-60), the LDA
 * matrix (model/utils.py:63-80), PLDA mean/transform/psi (model/_xv_plda/plda.py:27-49) and the
 * enrolled embeddings (model/utils.py:21-47).  All pointers are HOST fp32, PyTorch layouts.
 * The library folds the eval-mode BatchNorms (affine=False, xvecTDNN.py:17) into the following
 * layer's weights and re-lays everything out for the kernels. */
typedef struct sg_xv_weights {
    const float* tdnn_weight[5]; /* (cout, cin, k): (512,30,5) (512,512,5) (512,512,7) (512,512,1) (1500,512,1) */
    const float* tdnn_bias[5];   /* (cout) */
    const float* bn_mean[5];     /* running_mean (cout) */
    const float* bn_var[5];      /* running_var (cout)  */
    const float* fc1_weight;     /* (512, 3000) */
    const float* fc1_bias;       /* (512) */
    const float* emb_mean;       /* (512) */
    const float* lda;            /* (D, 513) last column = offset (iv_plda.py:423-435) */
    const float* plda_mean;      /* (D) */
    const float* plda_transform; /* (D, D) */
    const float* plda_psi;       /* (D) */
    const float* enroll;         /* (S, D) processed enrolment embeddings */
    int32_t D;
    int32_t S;
    float bn_eps;                /* 1e-5 */
    float threshold;             /* -INFINITY for CSI (xv_plda.py:41) */
} sg_xv_weights;

int sg_xv_load(sg_ctx* ctx, const sg_xv_weights* w);
/* replace the enrolled speakers / threshold only (model.enroll_embs, model.threshold) */
int sg_xv_set_enroll(sg_ctx* ctx, const float* enroll_host, int32_t S, float threshold);
/* The per-call `enroll_embs=` argument of forward / score / make_decision (model/iv_plda.py:155-165,172-194): score
 * the following passes against a caller-owned DEVICE table (S, D) instead of the model's enrolled set, which stays
 * untouched.  NULL ends the override.  No allocation, no synchronisation; the table must stay valid until the passes
 * enqueued while it was set have finished. */
int sg_xv_enroll_override(sg_ctx* ctx, const float* enroll_dev, int32_t S);

/* number of frames / TDNN output frames for T samples (0 if too short) */
int32_t sg_xv_num_frames(int32_t T);

/* loss selector: attack/utils.py:104-116 resolve_loss */
#define SG_LOSS_ENTROPY 0 /* SEC4SR_CrossEntropy, attack/utils.py:7-29 */
#define SG_LOSS_MARGIN 1  /* SEC4SR_MarginLoss,  attack/utils.py:31-102 */
#define SG_LOSS_LINEAR 2  /* loss[b] = sum_s coef[b][s] * score[b][s]: the vector-Jacobian product of the scores.  What
                           * autograd hands down at adaptive_attack/EOT.py:35 for ANY loss of the scores: a caller-defined
                           * loss L passes coef = dL/dscores and gets dL/dx; model/defended_model.py:67-75 ('average' order:
                           * the loss of the MEAN score of several defended branches) uses it per branch. */
#define SG_TASK_CSI 0
#define SG_TASK_SV 1
#define SG_TASK_OSI 2

typedef struct sg_loss_spec {
    int32_t loss;       /* SG_LOSS_* */
    int32_t task;       /* SG_TASK_* */
    int32_t targeted;   /* 0/1 */
    int32_t clip_max;   /* Margin: max(0, loss) (attack/utils.py:99-100) */
    float confidence;   /* Margin kappa */
    float threshold;    /* SV/OSI threshold used INSIDE the loss */
    const float* coef_dev; /* SG_LOSS_LINEAR: (B,S) device tensor, else NULL */
} sg_loss_spec;

/* Input levels, reference model/xv_plda.py:45-47 allowed_flags */
#define SG_FLAG_WAV 0  /* x: (B,1,T) waveform            */
#define SG_FLAG_RAW 1  /* x: (B,F,30) raw MFCC           */
#define SG_FLAG_CMVN 2 /* x: (B,F,30) CMVN-normalised    */

/* Dither policy for the MFCC front-end (xv_plda.py:119 hard-codes dither=1.0 from the global
 * RNG).  dither == 0 disables it.  Otherwise the noise is kaldi's sqrt(-2 ln u) cos(2 pi u) with
 * u from a counter-based generator keyed by (key, utterance, frame, sample), where for row b of the call
 *     g = row_base + b,  repeat = rep_rows > 0 ? g / rep_rows : 0,  key = seed + repeat * 0xC2B2AE3D27D4EB4F,
 *     utterance = index_base + (g - repeat * rep_rows)
 * i.e. index_base is the GLOBAL index of the first utterance (times the rows an utterance contributes to the call),
 * row_base the position of row 0 inside the full call this one is a slice of, and rep_rows > 0 says that the full
 * call's rows are EOT repeats of rep_rows rows (adaptive_attack/EOT.py:29 `x_batch.repeat(EOT_batch_size, 1, 1)`):
 * the noise an utterance sees does not depend on how a batch is chunked or sharded over GPUs.  noise_dev, when
 * non-NULL, is an explicit (B,F,400) tensor that is added instead (parity tests). */
typedef struct sg_dither {
    float dither;
    uint64_t seed;
    int64_t index_base;
    const float* noise_dev;
    int64_t row_base;
    int32_t rep_rows;
} sg_dither;

/* ---- per-stage entry points (parity tests; also what model.compute_feat etc. call) --------- */

/* model/utils.py:7-19 check_input_range(range_type='origin'): writes 32768.f or 1.f to
 * *scale_dev (device float) from the batch max/min; no host sync. */
int sg_input_scale(sg_ctx* ctx, const float* x_dev, int64_t n, float* scale_dev, void* stream);

/* model/xv_plda.py:107-156 raw() = torchaudio.compliance.kaldi.mfcc per utterance.
 * x (B,T) dev; scale_dev: device float multiplied into x (NULL = 1); feats (B,F,30) dev. */
int sg_xv_mfcc(sg_ctx* ctx, const float* x_dev, int32_t B, int32_t T, const float* scale_dev,
               const sg_dither* dither, float* feats_dev, void* stream);

/* model/iv_plda.py:296-377 cmvn(): sliding 300-frame centred mean subtraction. (B,F,30)->(B,F,30) */
int sg_xv_cmvn(sg_ctx* ctx, const float* feats_dev, int32_t B, int32_t F, float* out_dev, void* stream);

/* Backward of the two front-end stages alone (what autograd derives for xv_plda.py:107-156 and
 * iv_plda.py:296-377): needed when something sits BETWEEN the stages, i.e. a feature-level defense
 * (model/defended_model.py:46-65 process_sequential with a flag-1 or flag-2 defense).
 * sg_xv_mfcc_backward: d loss/d raw MFCC (B,F,30) -> d loss/d waveform (B,T); same scale / dither as the forward.
 * sg_xv_cmvn_backward: d loss/d CMVN features -> d loss/d raw features, (B,F,30) both. */
int sg_xv_mfcc_backward(sg_ctx* ctx, const float* x_dev, int32_t B, int32_t T, const float* scale_dev,
                        const sg_dither* dither, const float* dfeats_dev, float* grad_dev, void* stream);
int sg_xv_cmvn_backward(sg_ctx* ctx, const float* dout_dev, int32_t B, int32_t F, float* din_dev, void* stream);

/* Forward pass: model.make_decision / score / embedding (iv_plda.py:155-194, xv_plda.py:87-104).
 * Any output pointer may be NULL.  decisions (B) int64, scores (B,S), emb (B,D) processed
 * embedding (xv_plda.py:159-174), tdnn_emb (B,512) raw fc1 output (xvecTDNN.py:63). */
int sg_xv_forward(sg_ctx* ctx, const float* x_dev, int32_t B, int32_t T_or_F, int32_t flag,
                  const sg_dither* dither, int64_t* decisions_dev, float* scores_dev, float* emb_dev,
                  float* tdnn_emb_dev, void* stream);

/* TDNN layer activations of the LAST sg_xv_forward / sg_xv_loss_grad call (debug / parity):
 * layer 1..5 -> relu output (B, F_l, C_l) channel-last, C_5 padded to 1536; copies to out_dev. */
int sg_xv_debug_activation(sg_ctx* ctx, int32_t layer, float* out_dev, int64_t capacity_floats,
                           int32_t* rows_per_utt, int32_t* channels, void* stream);

/* The autograd call site the hand-coded backward replaces: adaptive_attack/EOT.py:32-35
 * (make_decision, loss, loss.backward(ones)).  grad has the shape of x (flag 0: (B,T); 1/2:
 * (B,F,30)); loss (B); outputs may be NULL. */
int sg_xv_loss_grad(sg_ctx* ctx, const float* x_dev, const int64_t* y_dev, int32_t B, int32_t T_or_F,
                    int32_t flag, const sg_loss_spec* loss, const sg_dither* dither,
                    int64_t* decisions_dev, float* scores_dev, float* loss_dev, float* grad_dev,
                    void* stream);

/* attack/FGSM.py:65,68: x += step*sign(grad)*grad_sign; x = min(max(x, lower), upper). In place. */
int sg_pgd_update(sg_ctx* ctx, float* x_dev, const float* grad_dev, const float* lower_dev,
                  const float* upper_dev, int64_t n, float step_size, int32_t grad_sign, void* stream);

/* attack/utils.py:7-102 on given scores (B,S): per-example loss, d loss/d scores and the decision (argmax, -1 unless
 * max > threshold) -- the tail kernels' loss stage alone.  Used where a loss is taken of scores that no single model
 * pass produced (model/defended_model.py 'average' order: mean score over the defended branches). */
int sg_loss_eval(sg_ctx* ctx, const float* scores_dev, const int64_t* y_dev, int32_t B, int32_t S, float threshold,
                 const sg_loss_spec* loss, int64_t* decisions_dev, float* loss_dev, float* dscores_dev, void* stream);

/* ---- attack-state updates around the model call (config 3: CW2, config 5: FAKEBOB/NES) -------
 * attack/CW2.py:72-82.  One pass over (B,T): if grad1 != NULL, the Adam update (torch.optim.Adam
 * defaults, step_t = 1-based step count) of `modifier` from grad1 = d loss1/d input_cur, where the full
 * objective is const[b]*loss1 + ||input_cur - x||^2 and input = tanh(modifier + atanh(0.999999 x));
 * then input_next = tanh(modifier + atanh(0.999999 x)) and loss2[b] = sum_t (input_next - x)^2.
 * Call with grad1 == NULL to produce the first input from a fresh modifier. */
int sg_cw2_step(sg_ctx* ctx, float* modifier_dev, float* exp_avg_dev, float* exp_avg_sq_dev,
                const float* x_dev, const float* input_cur_dev, const float* grad1_dev,
                const float* const_dev, int32_t B, int32_t T, float lr, int32_t step_t,
                float* input_next_dev, float* loss2_dev, void* stream);

/* adaptive_attack/NES.py:19-25: queries (n, 2*half + with_clean, T) = [x] , x + sigma z_p , x - sigma z_p.
 * z_p comes from noise_in (n, half, T) when given, else from a counter-based generator keyed by
 * (seed, example index + index_base, pair_base + p, t) -- the same noise is regenerated by
 * sg_nes_grad, so it is never stored.  noise_out (optional) receives the draws (parity tests). */
int sg_nes_queries(sg_ctx* ctx, const float* x_dev, int32_t n, int32_t T, int32_t half, int32_t with_clean,
                   float sigma, uint64_t seed, int64_t index_base, int32_t pair_base,
                   const float* noise_in_dev, float* queries_dev, float* noise_out_dev, void* stream);

/* adaptive_attack/NES.py:47,52,54: grad (n,T) (+)= mean over the 2*half noisy queries of loss * noise;
 * loss (n, 2*half + with_clean) in query order.  On the last chunk pass final_sigma = sigma (> 0) and
 * final_batches = number of chunks to apply NES.py:54's grad / sigma / num_batches. */
int sg_nes_grad(sg_ctx* ctx, const float* loss_dev, int32_t n, int32_t T, int32_t half, int32_t with_clean,
                uint64_t seed, int64_t index_base, int32_t pair_base, const float* noise_in_dev,
                int32_t accumulate, float final_sigma, int32_t final_batches, float* grad_dev, void* stream);

/* attack/FAKEBOB.py:93-104: grad <- momentum*prev + one_minus_momentum*grad (in place);
 * x <- min(max(x + grad_sign*lr[e]*sign(grad), lower), upper).  lr (n) per example. */
int sg_fakebob_step(sg_ctx* ctx, float* x_dev, float* grad_dev, const float* prev_grad_dev,
                    const float* lr_dev, const float* lower_dev, const float* upper_dev, int32_t n, int32_t T,
                    float momentum, float one_minus_momentum, int32_t grad_sign, void* stream);

/* ---- fused attack loop ----------------------------------------------------------------------
 * attack/FGSM.py:38-70 attack_batch for the xv_plda model: max_iter gradient steps plus the final
 * forward-only pass, entirely on the device (FGSM = max_iter 1, step_size epsilon). */
typedef struct sg_pgd_params {
    sg_loss_spec loss;
    float step_size;
    int32_t max_iter;
    int32_t grad_sign;       /* attack/utils.py:114 */
    int32_t eot_size;        /* EOT.py:16-54: passes per gradient step, each with fresh dither (key = dither.seed +
                              * step * 0x9E3779B97F4A7C15 + repeat * 0xC2B2AE3D27D4EB4F); their data gradients are summed
                              * in pass order before the sign step.  The repeats of a step are rows of ONE batch
                              * (eot_size x B rows, as many as fit one pass), with the bits of separate passes.
                              * With dither == 0 all repeats coincide: one pass. */
    int32_t eot_batch_size;  /* how the reference groups the repeats into model calls; must divide eot_size */
    sg_dither dither;
} sg_pgd_params;

/* x_adv (B,T) dev: in = start point, out = adversarial audio; lower/upper (B,T) dev.
 * success (B) uint8, decisions (B) int64, scores (B,S), loss (B): state at the final pass (a single forward,
 * FGSM.py:45-47).  loss_trace ((max_iter+1)*B) / decision_trace ((max_iter+1)*B): optional per-step records, as the
 * reference prints them (FGSM.py:50-58): the loss averaged over the step's EOT repeats, the decision voted over them
 * (attack/utils.py:118-125, first seen wins a tie).  A loss of kind SG_LOSS_LINEAR passes ONE (B,S) coef table; the
 * rows of the repeats of an utterance share its row. */
int sg_xv_pgd_run(sg_ctx* ctx, float* x_adv_dev, const int64_t* y_dev, const float* lower_dev,
                  const float* upper_dev, int32_t B, int32_t T, const sg_pgd_params* params,
                  uint8_t* success_dev, int64_t* decisions_dev, float* scores_dev, float* loss_dev,
                  float* loss_trace_dev, int64_t* decision_trace_dev, void* stream);

/* ---- AudioNet CSI-NE model (model/audionet_csine.py) ----------------------------------------
 * log-mel(32) front-end (model/_audionet/Preprocessor.py:85-112) -> 5x5 pre-filter -> seven
 * Conv1d/BatchNorm/ReLU(/MaxPool) blocks -> max over time -> Linear(32, num_class).  Input levels:
 * flag 0 = wav (B,1,T) in [-1,1] (int16-scaled input is divided by 32768, model/utils.py:15-16),
 * flag 1 = log-mel features (B,F,32).  All pointers in sg_an_weights are HOST fp32 tensors in the
 * reference's state_dict layout (keys convN.0.weight/bias, convN.1.weight/bias/running_mean/
 * running_var, fc.weight/bias); the BatchNorms are folded into the convolutions at load time. */
typedef struct sg_an_weights {
    const float* conv1_weight;   /* (1,1,5,5) */
    const float* conv1_bias;     /* (1) */
    const float* bn1[4];         /* conv1.1 weight, bias, running_mean, running_var (1 each) */
    const float* conv_weight[7]; /* conv2..conv8 .0.weight (cout, cin, 3) */
    const float* conv_bias[7];
    const float* bn_weight[7];
    const float* bn_bias[7];
    const float* bn_mean[7];
    const float* bn_var[7];
    const float* fc_weight;      /* (num_class, 32) */
    const float* fc_bias;        /* (num_class) */
    int32_t num_class;
    float bn_eps;                /* 1e-5 */
} sg_an_weights;

int sg_an_load(sg_ctx* ctx, const sg_an_weights* w);
/* frames of the centred STFT: 1 + (T-1)/160 (0 if T < 1024) */
int32_t sg_an_num_frames(int32_t T);
/* Preprocessor.forward: x (B,T) dev -> log-mel (B,F,32) dev (channel-last) */
int sg_an_logmel(sg_ctx* ctx, const float* x_dev, int32_t B, int32_t T, float* feats_dev, void* stream);
/* backward of Preprocessor.forward alone: d loss/d log-mel (B,F,32) -> d loss/d x (B,T); used when a
 * feature-level defense sits between the front-end and the CNN (defended_model.py:46-65).
 * reuse_forward != 0: the caller states that x_dev still holds exactly the samples of this context's last sg_an_logmel /
 * waveform-level pass (same pointer, B, T -- the library checks those, it cannot check the contents); the backward then
 * starts from that pass's mel energies instead of recomputing them (126 -> 82 us at 64 x 3 s).  With 0, or when pointer
 * or shape differ, the forward is recomputed. */
int sg_an_logmel_backward(sg_ctx* ctx, const float* x_dev, int32_t B, int32_t T, const float* dfeats_dev, float* grad_dev,
                          int32_t reuse_forward, void* stream);
/* How the log-mel front-end and its adjoint run on this context (defaults 32, -1, -1 = the library's choice by size):
 *   fft_bits 32 | 64: the scalar type of the STFT's transforms.  The reference computes its STFT in float32
 *     (model/_audionet/Preprocessor.py:100-105, torch.stft on a float32 signal); 64 is the form of rounds 1-4.
 *   spectrum_cache: the forward of a pass keeps every frame's packed spectrum (B x F x 4 KB) for the backward of the same
 *     pass instead of the backward transforming the frame again: 1 / 0, or -1 = decided per call from B x F (it pays
 *     everywhere except around 24-40 thousand frames).  An optional speed-up: when the buffer cannot be allocated the
 *     passes run without it.
 *   fused_overlap_add: the adjoint adds the frames' gradients up on chip (and applies the attack's update there) instead
 *     of writing B x F x 800 floats for a second kernel; same sums in the same order, same bits.  1 / 0, or -1: the
 *     library decides per call from the batch (the fused form cuts utterances into runs with 5 halo frames each and pays
 *     from ~200 utterances of 3 s; below that the separate pair is faster).
 * Takes effect from the next pass; results of the two transform precisions differ by float32 round-off. */
int sg_an_configure(sg_ctx* ctx, int32_t fft_bits, int32_t spectrum_cache, int32_t fused_overlap_add);
/* audionet_csine.make_decision / score / embedding (:149-257): decisions (B), scores (B,num_class), emb (B,32) */
int sg_an_forward(sg_ctx* ctx, const float* x_dev, int32_t B, int32_t T_or_F, int32_t flag,
                  int64_t* decisions_dev, float* scores_dev, float* emb_dev, void* stream);
/* activations of the last pass (parity): layer 1 = pre-filter output, 2..8 = conv2..conv8 block outputs
 * (after pooling where the block has one), channel-last (B, rows, C) */
int sg_an_debug_activation(sg_ctx* ctx, int32_t layer, float* out_dev, int64_t capacity_floats,
                           int32_t* rows_per_utt, int32_t* channels, void* stream);
/* adaptive_attack/EOT.py:32-35 for AudioNet: make_decision + loss + d loss/d x (hand-coded backward) */
int sg_an_loss_grad(sg_ctx* ctx, const float* x_dev, const int64_t* y_dev, int32_t B, int32_t T_or_F,
                    int32_t flag, const sg_loss_spec* loss, int64_t* decisions_dev, float* scores_dev,
                    float* loss_dev, float* grad_dev, void* stream);
/* attack/FGSM.py:38-70 attack_batch on AudioNet, whole loop on the device (params->dither ignored).
 * On a non-zero return x_adv_dev is UNSPECIFIED (the loop steps between two buffers when the overlap-add runs inside the
 * adjoint; an error in the middle leaves the caller's buffer on an earlier iterate): discard it with the error. */
int sg_an_pgd_run(sg_ctx* ctx, float* x_adv_dev, const int64_t* y_dev, const float* lower_dev,
                  const float* upper_dev, int32_t B, int32_t T, const sg_pgd_params* params,
                  uint8_t* success_dev, int64_t* decisions_dev, float* scores_dev, float* loss_dev,
                  float* loss_trace_dev, int64_t* decision_trace_dev, void* stream);

/* ---- measurement ---------------------------------------------------------------------------
 * Time `iters` launches of one TDNN contraction (layer 1..5 = forward, -1..-5 = data gradient) with HIP
 * events on `stream`; returns average milliseconds per launch in *ms_per_launch and the
 * algorithmic FLOPs of one launch in *flops, and the tile height (rows) the launcher picked in
 * *tile_rows.  Used by bench.py for the roofline block. */
int sg_xv_time_layer(sg_ctx* ctx, int32_t layer, int32_t B, int32_t T, int32_t iters,
                     float* ms_per_launch, double* flops, int32_t* tile_rows, void* stream);

/* Stage trace of the pass sequences (what rocprofv3 --kernel-trace shows, from inside the process).  The
 * reference has no counterpart: its only timing is a time.time() per training batch (adver_train.py:185,237); this
 * is the measurement hook SURVEY.md section 5 asks for "around the C-ABI step call".
 * Between sg_trace_begin and sg_trace_end every launch of sg_xv_forward / sg_xv_loss_grad / sg_xv_pgd_run and of
 * sg_an_forward / sg_an_loss_grad / sg_an_pgd_run / sg_an_pgd_run_feco (tags 30..) is bracketed by a pair of HIP events
 * on the launch stream, up to max_records launches (further launches are not recorded).  The per-stage entry points
 * (sg_xv_mfcc, sg_an_logmel, sg_feco_*, the attack-state updates) are not traced.  An event record that fails drops
 * its launch record, and sg_trace_end then returns SG_ERR_HIP with the count in sg_last_error.
 * sg_trace_end waits for the last recorded event, writes tag and elapsed milliseconds of each record in launch order
 * (at most `capacity`), the number of records to *n_out, and switches the trace off.  Tags: +l / -l = forward /
 * data-gradient contraction of TDNN layer l (1..5), others below.  Event records cost a few microseconds between
 * launches: trace a run of its own, not the run that is timed. */
#define SG_STAGE_MFCC_FWD 10
#define SG_STAGE_CMVN_FWD 11
#define SG_STAGE_POOL_FWD 12
#define SG_STAGE_FC1_FWD 13
#define SG_STAGE_TAIL 14
#define SG_STAGE_FC1_BWD 15
#define SG_STAGE_POOL_BWD 16
#define SG_STAGE_CMVN_BWD 17
#define SG_STAGE_MFCC_BWD 18
#define SG_STAGE_OVERLAP_ADD 19
/* AudioNet: 30 + l / 40 + l = forward / data-gradient contraction of conv block l (0..6 = conv2..conv8) */
#define SG_STAGE_AN_LOGMEL_FWD 20
#define SG_STAGE_AN_PREFILTER_FWD 21
#define SG_STAGE_AN_POOL_FWD 22
#define SG_STAGE_AN_TAIL 23
#define SG_STAGE_AN_POOL_BWD 24
#define SG_STAGE_AN_PREFILTER_BWD 25
#define SG_STAGE_AN_LOGMEL_BWD 26
#define SG_STAGE_AN_OVERLAP_ADD 27
#define SG_STAGE_AN_FECO_FWD 28
#define SG_STAGE_AN_FECO_BWD 29
#define SG_STAGE_AN_CONV_FWD 30
#define SG_STAGE_AN_CONV_BWD 40
#define SG_STAGE_AN_FUSED_FWD 50 /* the whole conv stack of a pass in one launch (round 4) */
#define SG_STAGE_AN_FUSED_BWD 51 /* (round 6: with the network's head inside, when a gradient follows) */
#define SG_STAGE_AN_FUSED_FWDBWD 52 /* forward + head + backward of whole utterances in one launch (round 6) */
int sg_trace_begin(sg_ctx* ctx, int32_t max_records);
int sg_trace_end(sg_ctx* ctx, int32_t* tags_out, float* ms_out, int32_t capacity, int32_t* n_out);

/* ---- the convolution primitive on caller buffers ------------------------------------------------
 * Dilated 1-D convolution / its data gradient on channel-last rows, the contraction behind
 * torch.nn.functional.conv1d in model/_xv_plda/xvecTDNN.py:16-33,49-53 and behind autograd's conv
 * input-gradient (adaptive_attack/EOT.py:35), exposed so the MFMA kernels can be checked in isolation:
 *
 *   C[b*Tc + t][n] = epi( sum_{j < taps} sum_{c < Kc} A[b*Ta + t + tap_base + j*tap_step][c] * W[j*Kc + c][n] )
 *
 * rows outside [0, Ta) of an utterance contribute zero.  A: (B*Ta, Kc) floats, W: (taps*Kc, N), C: (B*Tc, N);
 * Kc % 32 == 0, N % 128 == 0.  epi: 0 none, 1 relu(acc + bias[n]), 2 acc where mask[row][n] > 0 else 0
 * (mask shaped like C).  kernel: 0 = what the TDNN layers get (stream-K with 16-wave 256x128 quad-fed blocks when
 * the shape qualifies, one 16x16 block per wave on the 16x16x4 MFMA when there are at most 2800 such blocks, else one
 * quad-fed 64x128 block per tile), 1 = one b32-fed 64x128 block per tile, 2 = stream-K with the b32-fed 8-wave kernel,
 * (6 / 7 / 8 / 9 = stream-K with the roles split between waves, 128- / 64- / 32- / 256-row tiles, 10 = 128-row tiles as four
 * 64x64 computing waves; an error when the shape does not qualify),
 * 3 = stream-K with 8-wave 128x128 quad-fed blocks, 4 = one quad-fed 64x128 block per tile, 5 = one 16x16 block per
 * wave.  All choices give bit-identical results: the float32 fmaf chain restated in oracle/conv_chain.c. */
int sg_conv1d_rows(sg_ctx* ctx, const float* a_dev, const float* w_dev, float* c_dev, const float* bias_dev,
                   const float* mask_dev, int32_t B, int32_t Ta, int32_t Tc, int32_t Kc, int32_t N, int32_t taps,
                   int32_t tap_step, int32_t tap_base, int32_t epi, int32_t kernel, void* stream);

/* ---- data formats either side of the path ----------------------------
 * attackMain.py:154-166 save_audio + metric/metric.py:8-42, one pass over the batch:
 *   pcm (B,T) int16: per utterance, x * 2^15 if 0.9*max <= 1 and 0.9*min >= -1, then numpy astype(int16)
 *     (truncate toward zero, keep the low 16 bits: 1.0 -> -32768);
 *   metrics (B,5) double: L2, L0, L1, Linf of preprocess(adver) - preprocess(benign) and SNR in dB
 *     (+inf for a zero perturbation); preprocess divides by 2^15 unless -1 <= max <= 1 (metric.py:8-12).
 * pcm_dev or metrics_dev may be NULL; benign_dev is only needed for the metrics. */
int sg_wav_finalize(sg_ctx* ctx, const float* benign_dev, const float* adver_dev, int32_t B, int32_t T,
                    int16_t* pcm_dev, double* metrics_dev, void* stream);

/* set_threshold.py:22-47 set_threshold(score_target, score_untarget): scans the target scores in order and
 * keeps the first one minimising |FRR - FAR| (both in percent).  out3_dev: threshold, FRR, FAR (doubles).
 * Synchronises the stream. */
int sg_eer_threshold(sg_ctx* ctx, const float* target_dev, int32_t n_target, const float* untarget_dev,
                     int32_t n_untarget, double* out3_dev, void* stream);

/* ---- FeCo feature-level defense -------------------------------------------
 * defense/feature_level.py:168-217 kmeans(): cluster the F frames of each utterance into k = int(F * ratio)
 * groups and replace every group by the mean of its frames; feats (B,F,D), D <= 64.
 * sg_feco_kmeans: cluster ids (B,F) under this library's determinism contract (k_feco.hip header; the reference
 *   delegates to a randomly initialised third-party k-means, so its ids are not reproducible).
 * sg_feco_kmeans_seeded: the same clustering started from k distinct RANDOM frames (the reference's third-party k-means
 *   starts from a random draw of numpy's global generator, kmeans_pytorch initialize(): np.random.choice(F, k,
 *   replace=False)); here the draw is a function of (seed, index_base + utterance) only -- Philox4x32-10, k_feco.hip
 *   header -- so a pass is reproducible and independent of the shard layout, and fresh seeds per pass give the
 *   randomised defense that expectation-over-transformation attacks (adaptive_attack/EOT.py) average over.
 * sg_feco_compress: :204-216 given the ids: out (B,k,D) = cluster means, an empty cluster i takes frame i (the
 *   reference's `force` fallback; with B == 1 the reference drops such rows -- the host compacts using counts).
 * sg_feco_compress_backward: the gradient autograd derives for that step. */
int sg_feco_kmeans(sg_ctx* ctx, const float* feats_dev, int32_t B, int32_t F, int32_t D, int32_t k, int32_t max_iter,
                   int32_t* assign_dev, void* stream);
int sg_feco_kmeans_seeded(sg_ctx* ctx, const float* feats_dev, int32_t B, int32_t F, int32_t D, int32_t k, int32_t max_iter,
                          uint64_t seed, int64_t index_base, int32_t* assign_dev, void* stream);
int sg_feco_compress(sg_ctx* ctx, const float* feats_dev, const int32_t* assign_dev, int32_t B, int32_t F, int32_t D,
                     int32_t k, float* out_dev, int32_t* counts_dev, void* stream);
/* The whole forward of the defense in one launch: the clustering (random_init != 0: seeded form) and the cluster means
 * with the `force` fallback -- the means ARE the centroids of the clustering's last update (same ids, same ascending
 * sums), so out / counts equal sg_feco_compress of the returned ids bit for bit.
 * reps > 1 (seeded form only): the SAME features clustered reps times, repeat r from key seed + r * 0xC2B2AE3D27D4EB4F --
 * the EOT repeats of an adaptive attack (adaptive_attack/EOT.py:24-25 x_batch.repeat) without repeating the front-end;
 * assign (reps,B,F), out (reps,B,k,D), counts (reps,B,k).  sg_feco_compress_backward_reps is the matching gradient:
 * dout (reps,B,k,D) -> dfeats (B,F,D) = sum over the repeats, in repeat order (the compression is linear in the
 * features, so one front-end adjoint serves all repeats). */
int sg_feco_kmeans_compress(sg_ctx* ctx, const float* feats_dev, int32_t B, int32_t F, int32_t D, int32_t k, int32_t max_iter,
                            int32_t random_init, uint64_t seed, int64_t index_base, int32_t reps, int32_t* assign_dev,
                            float* out_dev, int32_t* counts_dev, void* stream);
int sg_feco_compress_backward_reps(sg_ctx* ctx, const float* dout_dev, const int32_t* assign_dev, const int32_t* counts_dev,
                                   int32_t B, int32_t F, int32_t D, int32_t k, int32_t force, int32_t reps,
                                   float* dfeats_dev, void* stream);
int sg_feco_compress_backward(sg_ctx* ctx, const float* dout_dev, const int32_t* assign_dev, const int32_t* counts_dev,
                              int32_t B, int32_t F, int32_t D, int32_t k, int32_t force, float* dfeats_dev, void* stream);

/* BASELINE.json configs[3]: PGD + EOT against the FeCo-defended AudioNet as ONE device-resident loop --
 * attack/FGSM.py:38-70 attack_batch with the model of model/defended_model.py:46-65 (FeCo at feature level 1:
 * waveform -> log-mel -> FeCo -> AudioNet CNN) and the gradient chained back through the defense by hand.
 * k = int(F * cl_r) is computed by the caller (Python float arithmetic, feature_level.py:184).  random_init != 0: every
 * gradient step clusters the step's log-mel features params->eot_size times from fresh random frames (repeat r of step
 * it: key = seed + it * 0x9E3779B97F4A7C15 + r * 0xC2B2AE3D27D4EB4F, sg_feco_kmeans_compress with reps), runs the CNN on
 * the eot_size x B compressed copies as one batch, sums the repeats' feature-level gradients in repeat order
 * (sg_feco_compress_backward_reps) and takes the sum through ONE log-mel adjoint: only the defense is random, so the
 * front-end is not repeated.  random_init == 0: deterministic defense, one pass per step.  Needs B >= 2 (with one utterance the reference drops
 * empty clusters, feature_level.py:209-212: host path).  Outputs as sg_an_pgd_run. */
typedef struct sg_feco_params {
    int32_t k;
    int32_t max_iter;     /* k-means assignment steps */
    int32_t random_init;
    uint64_t seed;
    int64_t index_base;   /* global index of utterance 0 (shard offset) */
} sg_feco_params;
int sg_an_pgd_run_feco(sg_ctx* ctx, float* x_adv_dev, const int64_t* y_dev, const float* lower_dev,
                       const float* upper_dev, int32_t B, int32_t T, const sg_pgd_params* params,
                       const sg_feco_params* feco, uint8_t* success_dev, int64_t* decisions_dev, float* scores_dev,
                       float* loss_dev, float* loss_trace_dev, int64_t* decision_trace_dev, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* SPEAKERGUARD_HIP_H */
