"""Natural-evolution-strategy gradient estimate; mirrors reference adaptive_attack/NES.py:15-56.

The antithetic queries and the loss-weighted noise average are native passes (``nes_queries`` /
``nes_grad``).  By default the Gaussian noise comes from the engine's counter-based generator and is
regenerated inside ``nes_grad`` instead of being stored (the reference draws ``torch.randn`` from the
global RNG, NES.py:19 -- a stream no other implementation can reproduce).  Set ``noise_fn(shape)`` to
feed explicit noise (tests).
"""
import torch

from ..attack.utils import resolve_prediction


class NES:
    _draws = 0  # distinct noise for every forward() in the process

    def __init__(self, samples_per_draw, samples_per_draw_batch, sigma, EOT_wrapper, noise_fn=None, seed=0):
        self.samples_per_draw = samples_per_draw
        self.samples_per_draw_batch_size = samples_per_draw_batch
        self.sigma = sigma
        self.EOT_wrapper = EOT_wrapper  # EOT wraps the model
        self.noise_fn = noise_fn
        self.seed = seed

    def forward(self, x, y):
        n_audios, n_channels, N = x.shape
        num_batches = self.samples_per_draw // self.samples_per_draw_batch_size
        half = self.samples_per_draw_batch_size // 2
        model = self.EOT_wrapper.model
        base = getattr(model, 'base_model', model)
        x = x.contiguous()
        index_base = 0
        if hasattr(base, 'noise_seed'):
            # keyed by (seed, attack call, restart, NES call inside the chunk) + global example index in the kernel:
            # independent of the chunking and of the shard layout (model/_engine_ops.py)
            base._nes_draw += 1
            seed = base.noise_seed(self.seed ^ 0x4E4553, base._nes_draw)
            index_base = base._index_base
        else:
            NES._draws += 1
            seed = (self.seed * 0x9E3779B97F4A7C15 + NES._draws) & 0xFFFFFFFFFFFFFFFF
        grad = torch.empty_like(x)
        EOT_num_batches = int(self.EOT_wrapper.EOT_size // self.EOT_wrapper.EOT_batch_size)
        for i in range(num_batches):
            with_clean = i == 0  # the clean sample rides along in the first chunk (NES.py:22-23)
            noise_in = None
            if self.noise_fn is not None:
                noise_in = self.noise_fn([n_audios, half, n_channels, N]).to(x.device, torch.float32).contiguous()
            eval_input, _ = base.nes_queries(x, half, with_clean, self.sigma, seed, i * half, noise_in, index_base=index_base)
            per = 2 * half + int(with_clean)
            eval_y = y.repeat_interleave(per)
            if hasattr(base, 'row_keys'):
                base._row_scale = per  # example e of the chunk owns rows [e * per, (e + 1) * per) of the model call
            try:
                scores, loss, _, decisions = self.EOT_wrapper(eval_input, eval_y)
            finally:
                if hasattr(base, 'row_keys'):
                    base._row_scale = 1
            loss = (loss / EOT_num_batches).view(n_audios, -1).contiguous()
            scores = (scores / EOT_num_batches).view(n_audios, -1, scores.shape[1])
            last = i == num_batches - 1
            base.nes_grad(loss, grad, n_audios, N, half, with_clean, seed, i * half, noise_in, i > 0,
                          self.sigma if last else 0.0, num_batches, index_base=index_base)
            if i == 0:
                adver_loss = loss[..., 0]
                adver_score = scores[:, 0, :]
                mean_loss = loss[..., 1:].mean(1)
                predicts = resolve_prediction(decisions).reshape(n_audios, -1)
                predict = predicts[:, 0]
            else:
                mean_loss = mean_loss + loss.mean(1)
        mean_loss = mean_loss / num_batches
        return mean_loss, grad, adver_loss, adver_score, predict

    __call__ = forward
