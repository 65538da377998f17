"""Restatement of diffusers 0.29.2 PNDMScheduler (skip_prk_steps) and DDIMScheduler (eta = 0) as used with the
SD1.5 scheduler_config.json — the schedulers the reference obtains at pipeline_audio_cond_animation.py:511 and
drives at :325-327 (set_timesteps), :337 (scale_model_input, identity), :364 (step).  Tensor-level, stateful,
written directly from the published algorithm; deliberately independent of asva_amd/schedulers.py (which plans
scalar coefficients for the fused kernel) so the two can be checked against each other.
Third-party source absent -> PARITY UNPINNED by reference vectors; pinned by known answers in tests/test_oracle.py
(51-entry PLMS timestep list 981, 961, 961, 941 ... 1; alpha-bar endpoints; one-step closed forms).
"""
from __future__ import annotations

import numpy as np
import torch


def make_alphas_cumprod(n=1000, beta_start=0.00085, beta_end=0.012):
    betas = torch.linspace(beta_start ** 0.5, beta_end ** 0.5, n, dtype=torch.float32) ** 2   # "scaled_linear"
    return torch.cumprod(1.0 - betas, dim=0)


class RefPNDM:
    def __init__(self, n_train=1000, steps_offset=1):
        self.n_train, self.offset = n_train, steps_offset
        self.acp = make_alphas_cumprod(n_train)
        self.final_acp = self.acp[0]                      # set_alpha_to_one = False
        self.init_noise_sigma = 1.0

    def set_timesteps(self, n):
        self.n = n
        ratio = self.n_train // n
        ts = (np.arange(0, n) * ratio).round() + self.offset
        plms = np.concatenate([ts[:-1], ts[-2:-1], ts[-1:]])[::-1].copy()
        self.timesteps = torch.from_numpy(plms.astype(np.int64))
        self.ets, self.counter, self.cur_sample = [], 0, None

    def _prev_sample(self, sample, t, prev_t, eps):
        a_t = self.acp[t]
        a_p = self.acp[prev_t] if prev_t >= 0 else self.final_acp
        b_t, b_p = 1 - a_t, 1 - a_p
        sample_coeff = (a_p / a_t) ** 0.5
        denom = a_t * b_p ** 0.5 + (a_t * b_t * a_p) ** 0.5
        return sample_coeff * sample - (a_p - a_t) * eps / denom

    def step(self, eps, t, sample):
        t = int(t)
        prev_t = t - self.n_train // self.n
        if self.counter != 1:
            self.ets = self.ets[-3:]
            self.ets.append(eps)
        else:
            prev_t = t
            t = t + self.n_train // self.n
        if len(self.ets) == 1 and self.counter == 0:
            self.cur_sample = sample
        elif len(self.ets) == 1 and self.counter == 1:
            eps = (eps + self.ets[-1]) / 2
            sample = self.cur_sample
            self.cur_sample = None
        elif len(self.ets) == 2:
            eps = (3 * self.ets[-1] - self.ets[-2]) / 2
        elif len(self.ets) == 3:
            eps = (23 * self.ets[-1] - 16 * self.ets[-2] + 5 * self.ets[-3]) / 12
        else:
            eps = (55 * self.ets[-1] - 59 * self.ets[-2] + 37 * self.ets[-3] - 9 * self.ets[-4]) / 24
        out = self._prev_sample(sample, t, prev_t, eps)
        self.counter += 1
        return out


class RefDDIM:
    def __init__(self, n_train=1000, steps_offset=1):
        self.n_train, self.offset = n_train, steps_offset
        self.acp = make_alphas_cumprod(n_train)
        self.final_acp = self.acp[0]
        self.init_noise_sigma = 1.0

    def set_timesteps(self, n):
        self.n = n
        ratio = self.n_train // n
        ts = (np.arange(0, n) * ratio).round()[::-1].copy().astype(np.int64) + self.offset
        self.timesteps = torch.from_numpy(ts)

    def step(self, eps, t, sample):
        t = int(t)
        prev_t = t - self.n_train // self.n
        a_t = self.acp[t]
        a_p = self.acp[prev_t] if prev_t >= 0 else self.final_acp
        x0 = (sample - (1 - a_t) ** 0.5 * eps) / a_t ** 0.5
        direction = (1 - a_p) ** 0.5 * eps            # eta = 0: no variance term
        return a_p ** 0.5 * x0 + direction
