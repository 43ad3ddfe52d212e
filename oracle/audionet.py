"""AudioNet CSI-NE (log-mel front-end + 1-D CNN), restated on PyTorch-CPU fp32.  TEST INFRASTRUCTURE.

PINNED (round 2) against the reference's own code with two disclosed harness accommodations: tests/golden/an_ref.npz
holds outputs of reference model/audionet_csine.py + model/_audionet/Preprocessor.py + attack/utils.py executed
unmodified by tests/golden/make_golden_frontends.py, where (a) the uninstalled ``librosa.filters.mel`` (librosa==0.8.0,
README.md:56) is supplied by the third-party ``transformers.audio_utils.mel_filter_bank(norm='slaney',
mel_scale='slaney')`` and (b) ``torch.stft`` called without ``return_complex`` (Preprocessor.py:100-105, an error on
torch >= 2) gets the pre-1.8 real-view return.  This file reproduces that run exactly (max abs difference 0.0 on
log-mel, logits, loss and both gradients; tests/test_oracle_frontends.py).  What remains outside the reference's own
arithmetic is the mel basis itself, which restates the published librosa 0.8.0 algorithm (Slaney scale,
``norm='slaney'``, fmin 0, fmax sr/2) and agrees with the third-party one to 5e-10.
"""
import numpy as np
import torch
import torch.nn.functional as F

SR, N_MELS, N_FFT, HOP, WIN, PREEMPH = 16000, 32, 1024, 160, 800, 0.97
EPSILON = 1e-16
BN_EPS = 1e-5
# (name, cin, cout, kernel, padding, maxpool) -- audionet_csine.py:75-115
CONV_SPEC = (("conv2", 32, 64, 3, 1, True), ("conv3", 64, 128, 3, 1, False), ("conv4", 128, 128, 3, 1, False),
             ("conv5", 128, 128, 3, 1, True), ("conv6", 128, 128, 3, 1, False), ("conv7", 128, 64, 3, 1, True),
             ("conv8", 64, 32, 3, 0, False))


def hz_to_mel(f):
    f = np.asanyarray(f, dtype=np.float64)
    f_sp = 200.0 / 3
    mels = f / f_sp
    min_log_hz, min_log_mel, logstep = 1000.0, 1000.0 / f_sp, np.log(6.4) / 27.0
    return np.where(f >= min_log_hz, min_log_mel + np.log(np.maximum(f, 1e-10) / min_log_hz) / logstep, mels)


def mel_to_hz(m):
    m = np.asanyarray(m, dtype=np.float64)
    f_sp = 200.0 / 3
    min_log_hz, min_log_mel, logstep = 1000.0, 1000.0 / f_sp, np.log(6.4) / 27.0
    return np.where(m >= min_log_mel, min_log_hz * np.exp(logstep * (m - min_log_mel)), f_sp * m)


def mel_basis():
    """librosa.filters.mel(16000, 1024, 32, fmin=0, fmax=8000) of librosa 0.8.0 -> (32, 513) float32."""
    fftfreqs = np.linspace(0, SR / 2.0, 1 + N_FFT // 2)
    mel_f = mel_to_hz(np.linspace(hz_to_mel(0.0), hz_to_mel(SR / 2.0), N_MELS + 2))
    fdiff = np.diff(mel_f)
    ramps = np.subtract.outer(mel_f, fftfreqs)
    w = np.zeros((N_MELS, 1 + N_FFT // 2))
    for i in range(N_MELS):
        lower = -ramps[i] / fdiff[i]
        upper = ramps[i + 2] / fdiff[i + 1]
        w[i] = np.maximum(0, np.minimum(lower, upper))
    enorm = 2.0 / (mel_f[2:N_MELS + 2] - mel_f[:N_MELS])
    return (w * enorm[:, None]).astype(np.float32)


_MEL = None


def preprocess(wav):
    """Preprocessor.forward :88-112.  wav (B, T) in [-1, 1] -> log-mel (B, 32, frames)."""
    global _MEL
    if _MEL is None:
        _MEL = torch.from_numpy(mel_basis())
    wav = wav[:, 1:] - PREEMPH * wav[:, :-1]
    spec = torch.stft(wav, n_fft=N_FFT, hop_length=HOP, win_length=WIN,
                      window=torch.hann_window(WIN, dtype=wav.dtype), center=True, pad_mode="reflect",
                      return_complex=True)
    mag = spec.real.pow(2) + spec.imag.pow(2)  # _square of the (re, im) pair
    mel = torch.matmul(mag.transpose(2, 1), _MEL.t().to(wav.dtype)).transpose(2, 1)
    return 10 * torch.clamp(mel, EPSILON).log10()


def check_input_range(x, range_type="scale", bits=16):
    ori_type = "scale" if 0.9 * x.max() <= 1 and 0.9 * x.min() >= -1 else "origin"
    if range_type != ori_type:
        return x * (2 ** (bits - 1)) if ori_type == "scale" else x / (2 ** (bits - 1))
    return x


class AudioNet:
    allowed_flags = [0, 1]
    range_type = "scale"

    def __init__(self, state_dict):
        self.p = {k: torch.as_tensor(np.asarray(v)).float().clone() for k, v in state_dict.items()
                  if "num_batches_tracked" not in k}
        self.num_spks = self.p["fc.bias"].shape[0]
        self.threshold = -np.inf

    def double(self):
        self.p = {k: v.double() for k, v in self.p.items()}
        return self

    def _bn(self, x, name):
        p = self.p
        return F.batch_norm(x, p[name + ".running_mean"], p[name + ".running_var"], p[name + ".weight"], p[name + ".bias"],
                            False, 0.1, BN_EPS)

    def raw(self, x):
        return preprocess(x.squeeze(1)).transpose(1, 2)  # (B, T, F)

    def compute_feat(self, x, flag=1):
        assert flag == 1
        return self.raw(check_input_range(x, self.range_type))

    def layers(self, feats):
        """extract_emb :176-207; returns the list of layer outputs (after pooling where present)."""
        p = self.p
        x = feats.transpose(1, 2).unsqueeze(1)
        x = self._bn(F.conv2d(x, p["conv1.0.weight"], p["conv1.0.bias"], padding=2), "conv1.1").squeeze(1)
        outs = [x]
        for name, _, _, _, pad, pool in CONV_SPEC:
            if name == "conv8" and x.shape[2] < 3:  # repeat-pad, :195-203
                n = -(-3 // x.shape[2])
                x = x.repeat(1, 1, n)
            x = F.relu(self._bn(F.conv1d(x, p[name + ".0.weight"], p[name + ".0.bias"], padding=pad), name + ".1"))
            if pool:
                x = F.max_pool1d(x, 2, stride=2)
            outs.append(x)
        return outs

    def extract_emb(self, feats):
        return self.layers(feats)[-1].max(2)[0]

    def embedding(self, x, flag=0):
        return self.extract_emb(self.compute_feat(x, 1) if flag == 0 else x)

    def forward(self, x, flag=0, return_emb=False, enroll_embs=None):
        emb = self.embedding(x, flag)
        logits = F.linear(emb, self.p["fc.weight"], self.p["fc.bias"])
        return (logits, emb) if return_emb else logits

    __call__ = forward

    def score(self, x, flag=0, enroll_embs=None):
        return self.forward(x, flag)

    def make_decision(self, x, flag=0, enroll_embs=None):
        scores = self.score(x, flag)
        return torch.argmax(scores, dim=1), scores
