"""Natural-evolution-strategy gradient estimate; mirrors reference adaptive_attack/NES.py:15-56."""
import torch

from ..attack.utils import resolve_prediction


class NES:

    def __init__(self, samples_per_draw, samples_per_draw_batch, sigma, EOT_wrapper):
        self.samples_per_draw = samples_per_draw
        self.samples_per_draw_batch_size = samples_per_draw_batch
        self.sigma = sigma
        self.EOT_wrapper = EOT_wrapper  # EOT wraps the model

    def forward(self, x, y):
        n_audios, n_channels, N = x.shape
        num_batches = self.samples_per_draw // self.samples_per_draw_batch_size
        for i in range(num_batches):
            # antithetic pairs; the clean sample rides along in the first chunk (NES.py:19-23)
            noise = torch.randn([n_audios, self.samples_per_draw_batch_size // 2, n_channels, N], device=x.device)
            noise = torch.cat((noise, -noise), 1)
            if i == 0:
                noise = torch.cat((torch.zeros_like(x).unsqueeze(1), noise), 1)
            eval_input = (noise * self.sigma + x.unsqueeze(1)).view(-1, n_channels, N)
            per = self.samples_per_draw_batch_size + 1 if i == 0 else self.samples_per_draw_batch_size
            eval_y = y.repeat_interleave(per)
            scores, loss, _, decisions = self.EOT_wrapper(eval_input, eval_y)
            EOT_num_batches = int(self.EOT_wrapper.EOT_size // self.EOT_wrapper.EOT_batch_size)
            loss = (loss / EOT_num_batches).view(n_audios, -1)
            scores = (scores / EOT_num_batches).view(n_audios, -1, scores.shape[1])
            if i == 0:
                adver_loss = loss[..., 0]
                loss = loss[..., 1:]
                adver_score = scores[:, 0, :]
                noise = noise[:, 1:, :, :]
                grad = torch.mean(loss.unsqueeze(2).unsqueeze(3) * noise, 1)
                mean_loss = loss.mean(1)
                predicts = resolve_prediction(decisions).reshape(n_audios, -1)
                predict = predicts[:, 0]
            else:
                grad = grad + torch.mean(loss.unsqueeze(2).unsqueeze(3) * noise, 1)
                mean_loss = mean_loss + loss.mean(1)
        grad = grad / self.sigma / num_batches
        mean_loss = mean_loss / num_batches
        return mean_loss, grad, adver_loss, adver_score, predict

    __call__ = forward
