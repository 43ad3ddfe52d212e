"""PGD; mirrors reference attack/PGD.py (per-example epsilon ball, random restarts, best-of-inits)."""
import numpy as np
import torch

from .FGSM import FGSM


class PGD(FGSM):

    def __init__(self, model, task='CSI', epsilon=0.002, step_size=0.0004, max_iter=10, num_random_init=0,
                 loss='Entropy', targeted=False,
                 batch_size=1, EOT_size=1, EOT_batch_size=1,
                 verbose=1):
        self.model = model
        self.task = task
        self.epsilon = epsilon
        self.step_size = step_size
        self.max_iter = max_iter
        self.num_random_init = num_random_init
        self.loss_name = loss
        self.targeted = targeted
        self.batch_size = batch_size
        self._init_common(EOT_size, EOT_batch_size, verbose)

    def attack(self, x, y):
        self._check_inputs(x, y)
        self._begin_attack()
        upper = torch.clamp(x + self.epsilon, max=1)   # PGD.py:48-49
        lower = torch.clamp(x - self.epsilon, min=-1)
        x_ori = x.clone() if self.num_random_init > 0 else x  # only the restarts overwrite x (PGD.py:58-61)
        best_success_rate = -1
        best_success = None
        best_adver_x = None
        for init in range(max(1, self.num_random_init)):
            if self.num_random_init > 0:  # host RNG, same call as PGD.py:60
                x = x_ori + torch.tensor(np.random.uniform(-self.epsilon, self.epsilon, tuple(x_ori.shape)),
                                         device=x.device, dtype=x.dtype)
            adver_x, success = self._run_batches(x, y, lower, upper, tag=init)
            if sum(success) / len(success) > best_success_rate:   # whole-batch criterion, PGD.py:74-77
                best_success_rate = sum(success) / len(success)
                best_success = success
                best_adver_x = adver_x
        return best_adver_x, best_success
