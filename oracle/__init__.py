"""CPU oracle for the SpeakerGuard attack hot path.  TEST INFRASTRUCTURE -- NOT PRODUCT CODE.

Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may
import anything from this package, and there only as the checker / the reported CPU baseline.
Nothing under ``speakerguard_amd/`` imports it; the product path raises when the HIP library
is missing instead of falling back to this code.

What it is: a plain PyTorch-CPU fp32 restatement of the reference algorithm, function by
function, each citing the reference file:line it follows.

Parity status (see DESIGN.md "Oracle pinning"):

* from MFCC features onward (CMVN, TDNN, pooling, LDA/PLDA back-end, scoring, decisions,
  losses, d loss / d features) and all attack logic (FGSM/PGD/CWinf/CW2/FAKEBOB/EOT/NES):
  PINNED -- checked in ``tests/test_oracle_golden.py`` against fixtures under ``tests/golden/``
  that ``tests/golden/make_golden.py`` produced by importing the reference itself in the build
  container.
* waveform -> MFCC (``oracle.kaldi_mfcc``): PARITY UNPINNED.  The arithmetic lives in
  torchaudio==0.6.0 (reference README.md:54; call site model/xv_plda.py:114-148), which is
  neither vendored in the reference nor installed; the module restates the published Kaldi
  algorithm as torchaudio.compliance.kaldi implements it.
* AudioNet front-end and CNN (``oracle.audionet``): PARITY UNPINNED (librosa==0.8.0 and the
  old torch.stft API are not runnable here; reference model/_audionet/Preprocessor.py:57,100).
* dilated conv contraction (``oracle.conv_rows``): PINNED against torch.nn.functional.conv1d and its autograd.
* int16 PCM rounding, perturbation metrics, EER threshold (``oracle.post``) and FAKEBOB.estimate_threshold
  (``oracle.attacks``): PINNED by fixtures the reference's own functions produced (tests/golden/make_golden.py).
* FeCo (``oracle.feco``): the k-means ids restate THIS repository's determinism contract (the reference's clustering
  is a randomly initialised third-party k-means) and the means-and-fallback step restates
  defense/feature_level.py:204-216 -- PARITY UNPINNED.
"""
