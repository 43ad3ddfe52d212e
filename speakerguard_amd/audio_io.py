"""On-disk format of the adversarial audio; mirrors reference attackMain.py:154-166 save_audio.

The int16 quantisation (x * 2^15 when the utterance is in the float domain, then numpy's truncating, wrapping
astype(int16)) runs on the device in the same pass as the perturbation metrics (C-ABI ``sg_wav_finalize``);
only the resulting PCM crosses PCIe (96 KB per 3-s utterance instead of 192 KB).
"""
import os

import torch
from scipy.io.wavfile import write

from . import _native as N
from .metric.metric import _context


def quantize_pcm(advers):
    """(N,1,T) or (N,T) float tensor -> (N,T) int16 CPU numpy array, reference save_audio rounding."""
    a = advers.detach().to(torch.float32)
    a = a.reshape(a.shape[0], -1)
    if not a.is_cuda:
        a = a.to("cuda:0")
    a = a.contiguous()
    pcm = torch.empty(a.shape, device=a.device, dtype=torch.int16)
    _context(a.device).call("sg_wav_finalize", None, N._ptr(a), a.shape[0], a.shape[1], N._ptr(pcm), None,
                            N.current_stream_ptr(a.device))
    return pcm.cpu().numpy()


def save_audio(advers, names, root, fs=16000):
    """Writes root/<spk_id>/<name>.wav for every utterance (spk_id = name.split('-')[0], :161-166)."""
    pcm = quantize_pcm(advers)
    for adver, name in zip(pcm, names):
        spk_dir = os.path.join(root, name.split("-")[0])
        os.makedirs(spk_dir, exist_ok=True)
        write(os.path.join(spk_dir, name + ".wav"), fs, adver)
