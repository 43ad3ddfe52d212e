"""Speaker enrollment on the native forward path; mirrors the per-speaker body of reference enroll.py:40-111
(the surrounding directory walking / argparse driver is out of scope).

    emb          = mean over the speaker's enrollment utterances of model.embedding(audio)      (:49-62)
    z-norm stats = mean / std of model.score(test_audio, enroll_embs=emb) over other speakers'   (:68-92)
                   test utterances
    speaker_model line "id path znorm_mean znorm_std"                                            (:94-108)
"""
import numpy as np
import torch


def _as_batch(audio, device):
    a = audio if isinstance(audio, torch.Tensor) else torch.as_tensor(np.asarray(audio, dtype=np.float32))
    a = a.to(device=device, dtype=torch.float32)
    if a.dim() == 1:
        a = a.view(1, 1, -1)
    elif a.dim() == 2:
        a = a.unsqueeze(0)  # (1, T) as torchaudio.load returns -> (1, 1, T), enroll.py:53
    return a


def enroll_speaker(model, utterances):
    """Mean embedding (1, dim) over the enrollment utterances (each (1,T) / (T,) in [-1,1] or int16 scale;
    the reference multiplies by 2^15 first (:52), which check_input_range makes equivalent)."""
    emb, n = None, 0
    for audio in utterances:
        e = model.embedding(_as_batch(audio, model.device))  # (1, dim)
        emb = e.clone() if emb is None else emb + e
        n += 1
    if n == 0:
        raise ValueError("no enrollment utterances")
    return emb / n


def znorm_stats(model, emb, test_utterances):
    """(mean, std) of the scores of other speakers' test utterances against `emb` (:68-92; np.std, ddof 0).
    ``enroll_embs=`` is a per-call override (iv_plda.py:155-165): the model's own enrolled set is not touched."""
    scores = []
    for audio in test_utterances:
        s = model.score(_as_batch(audio, model.device), enroll_embs=emb)
        scores.append(float(s.flatten()[0].item()))
    return float(np.mean(scores)), float(np.std(scores))


def speaker_model_line(spk_id, emb_path, z_norm_mean, z_norm_std):
    """One row of the speaker_model file model/utils.py:21-47 parses (enroll.py:94)."""
    return "{} {} {} {}".format(spk_id, emb_path, z_norm_mean, z_norm_std)
