"""PNDM (PLMS) and DDIM schedulers for the denoising loop — host-side scalar schedule + one fused device
kernel per step (avsd_guided_step: guidance combine, multistep blend, latent update, frame-0 pinning).

The reference takes its scheduler from diffusers (PNDMScheduler.from_pretrained(sd15, "scheduler"),
pipeline_audio_cond_animation.py:511; called at :325-327,337,364).  diffusers 0.29.2 is not vendored in the
reference nor installed here, so the update rules are restated from its published algorithm
(`PNDMScheduler.set_timesteps/step_plms/_get_prev_sample`, `DDIMScheduler.set_timesteps/step`) for the SD1.5
scheduler_config.json: scaled_linear betas 0.00085..0.012 over 1000 steps, steps_offset 1, skip_prk_steps
true, set_alpha_to_one false, epsilon prediction, "leading" timestep spacing.

Both classes also expose the object protocol the reference pipeline uses (`set_timesteps`, `timesteps`,
`init_noise_sigma`, `scale_model_input`, `step(...).prev_sample`) so they drop into
AudioCondAnimationPipeline; the fast path (`plan_step` + `ops.guided_step`) folds guidance and the update
into one launch and keeps the eps history on the device.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import List, Optional, Tuple

import numpy as np
import torch


def alphas_cumprod(num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012, schedule="scaled_linear") -> np.ndarray:
    if schedule == "scaled_linear":
        # diffusers computes this in float32 torch; keep f32 so table entries are bit-identical
        betas = torch.linspace(beta_start ** 0.5, beta_end ** 0.5, num_train_timesteps, dtype=torch.float32) ** 2
    elif schedule == "linear":
        betas = torch.linspace(beta_start, beta_end, num_train_timesteps, dtype=torch.float32)
    else:
        raise NotImplementedError(schedule)
    return torch.cumprod(1.0 - betas, dim=0).numpy().astype(np.float64)


@dataclass
class StepPlan:
    """Everything avsd_guided_step needs for one scheduler step."""
    ca: float                 # coefficient of the (possibly saved) sample
    cb: float                 # coefficient of the blended epsilon
    w_cur: float = 1.0        # weight of this step's epsilon
    store_slot: int = -1      # ring slot to store this step's epsilon in (-1: do not store)
    hist_idx: Tuple[int, ...] = ()
    hist_w: Tuple[float, ...] = ()
    use_saved_sample: bool = False    # PLMS second step restarts from the sample saved at step 0
    save_sample: bool = False         # PLMS first step saves its input sample


class _Output:
    def __init__(self, prev_sample):
        self.prev_sample = prev_sample


class _Base:
    order = 1
    init_noise_sigma = 1.0

    def __init__(self, num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear",
                 steps_offset=1, set_alpha_to_one=False, prediction_type="epsilon", timestep_spacing="leading",
                 trained_betas=None, clip_sample=False, thresholding=False, rescale_betas_zero_snr=False, **unknown):
        if prediction_type != "epsilon" or timestep_spacing != "leading":
            raise NotImplementedError("only the SD1.5 scheduler configuration (epsilon, leading) is restated")
        # options that change the samples must not be dropped silently (diffusers' DDIM default is clip_sample=True;
        # the SD1.5 scheduler_config.json sets it to false)
        if clip_sample or thresholding or rescale_betas_zero_snr or trained_betas is not None:
            raise NotImplementedError("clip_sample / thresholding / rescale_betas_zero_snr / trained_betas are not restated: "
                                      "the SD1.5 scheduler configuration sets none of them")
        known_inert = {"clip_sample_range", "dynamic_thresholding_ratio", "sample_max_value", "skip_prk_steps"}
        bad = sorted(k for k in unknown if not k.startswith("_") and k not in known_inert)
        if bad:
            raise NotImplementedError(f"unknown scheduler options {bad}")
        self.num_train_timesteps = num_train_timesteps
        self.steps_offset = steps_offset
        self.acp = alphas_cumprod(num_train_timesteps, beta_start, beta_end, beta_schedule)
        self.final_alpha_cumprod = 1.0 if set_alpha_to_one else float(self.acp[0])
        self.num_inference_steps: Optional[int] = None
        self.timesteps: Optional[torch.Tensor] = None
        self.config = dict(num_train_timesteps=num_train_timesteps, beta_start=beta_start, beta_end=beta_end,
                           beta_schedule=beta_schedule, steps_offset=steps_offset, set_alpha_to_one=set_alpha_to_one,
                           prediction_type=prediction_type, timestep_spacing=timestep_spacing)

    def scale_model_input(self, sample, timestep=None):
        return sample

    @classmethod
    def from_pretrained(cls, path: str, subfolder: Optional[str] = None):
        import json
        import os

        p = os.path.join(path, subfolder) if subfolder else path
        with open(os.path.join(p, "scheduler_config.json")) as f:
            cfg = json.load(f)
        return cls(**{k: v for k, v in cfg.items() if not k.startswith("_")})

    def _acp(self, t: int) -> float:
        return float(self.acp[t]) if t >= 0 else self.final_alpha_cumprod


class DDIMScheduler(_Base):
    """eta = 0, no clipping / thresholding: x' = sqrt(a'/a) x + (sqrt(1-a') - sqrt(a'(1-a)/a)) eps."""

    def set_timesteps(self, num_inference_steps: int, device=None):
        self.num_inference_steps = num_inference_steps
        ratio = self.num_train_timesteps // num_inference_steps
        ts = (np.arange(0, num_inference_steps) * ratio).round()[::-1].copy().astype(np.int64) + self.steps_offset
        self._ts = [int(t) for t in ts]
        self.timesteps = torch.from_numpy(ts).to(device) if device is not None else torch.from_numpy(ts)

    def num_forwards(self) -> int:
        return len(self._ts)

    def plan_step(self, i: int) -> StepPlan:
        t = self._ts[i]
        prev = t - self.num_train_timesteps // self.num_inference_steps
        a, ap = self._acp(t), self._acp(prev)
        ca = (ap / a) ** 0.5
        cb = (1.0 - ap) ** 0.5 - (ap * (1.0 - a) / a) ** 0.5
        return StepPlan(ca=ca, cb=cb)

    def step(self, model_output, timestep, sample, eta: float = 0.0, generator=None, return_dict=True, **_):
        if eta != 0.0:
            raise NotImplementedError("DDIM eta != 0")
        i = self._ts.index(int(timestep))
        p = self.plan_step(i)
        prev = p.ca * sample + p.cb * model_output
        return _Output(prev) if return_dict else (prev,)


class PNDMScheduler(_Base):
    """PLMS (skip_prk_steps=True): 4th-order linear multistep on the stored epsilons, with the doubled second
    timestep that bootstraps the history (diffusers step_plms)."""

    ring_slots = 4

    def __init__(self, skip_prk_steps=True, cur_sample_aliases_latents: bool = True, **kw):
        """cur_sample_aliases_latents (default True = what the reference pipeline actually computes): diffusers'
        step_plms keeps `self.cur_sample = sample` WITHOUT cloning, and the reference passes a view,
        `video_latents[:, :, 1:]`, then writes the result back into that same storage
        (pipeline_audio_cond_animation.py:364; `.contiguous()` on :365 is a no-op).  The sample "restored" at the
        repeated second timestep is therefore the already-updated latents x1, not x0.  False gives textbook PLMS
        (what diffusers does when the caller rebinds `latents = step(...).prev_sample`)."""
        super().__init__(**kw)
        if not skip_prk_steps:
            raise NotImplementedError("PRK warm-up steps (skip_prk_steps=False)")
        self.config["skip_prk_steps"] = True
        self.cur_sample_aliases_latents = cur_sample_aliases_latents

    def set_timesteps(self, num_inference_steps: int, device=None):
        self.num_inference_steps = num_inference_steps
        ratio = self.num_train_timesteps // num_inference_steps
        base = (np.arange(0, num_inference_steps) * ratio).round().astype(np.int64) + self.steps_offset
        plms = np.concatenate([base[:-1], base[-2:-1], base[-1:]])[::-1].copy()
        self._ts = [int(t) for t in plms]
        self.timesteps = torch.from_numpy(plms).to(device) if device is not None else torch.from_numpy(plms)
        # object-protocol state
        self.ets: List[torch.Tensor] = []
        self.counter = 0
        self.cur_sample = None

    def num_forwards(self) -> int:
        return len(self._ts)

    def _coeffs(self, t: int, prev: int) -> Tuple[float, float]:
        a, ap = self._acp(t), self._acp(prev)
        b, bp = 1.0 - a, 1.0 - ap
        ca = (ap / a) ** 0.5
        denom = a * bp ** 0.5 + (a * b * ap) ** 0.5
        return ca, -(ap - a) / denom

    def plan_step(self, i: int) -> StepPlan:
        """Step i of the loop (i = diffusers' `counter`).  History ring: epsilon of the k-th APPENDING step
        lives in slot k % 4; step 1 (the repeated timestep) does not append."""
        ratio = self.num_train_timesteps // self.num_inference_steps
        t = self._ts[i]
        prev = t - ratio
        if i == 1:
            prev, t = t, t + ratio
        ca, cb = self._coeffs(t, prev)
        textbook = not self.cur_sample_aliases_latents
        if i == 0:
            return StepPlan(ca, cb, w_cur=1.0, store_slot=0, save_sample=textbook)
        if i == 1:
            return StepPlan(ca, cb, w_cur=0.5, store_slot=-1, hist_idx=(0,), hist_w=(0.5,), use_saved_sample=textbook)
        n_app = i            # appended epsilons after this step's append: steps 0,2,3,... -> i of them (i >= 2)
        cur = (n_app - 1) % self.ring_slots
        s = lambda back: (n_app - 1 - back) % self.ring_slots  # noqa: E731
        if n_app == 2:
            return StepPlan(ca, cb, 0.0, cur, (s(0), s(1)), (1.5, -0.5))
        if n_app == 3:
            return StepPlan(ca, cb, 0.0, cur, (s(0), s(1), s(2)), (23 / 12, -16 / 12, 5 / 12))
        return StepPlan(ca, cb, 0.0, cur, (s(0), s(1), s(2), s(3)), (55 / 24, -59 / 24, 37 / 24, -9 / 24))

    # object protocol (tensor-level, any device) — the same arithmetic, used when the scheduler is driven
    # through `.step()` by the reference-style loop.  Like diffusers it keeps `sample` itself (no clone), so a
    # caller that passes a view and writes back in place gets the aliasing described in __init__.
    def step(self, model_output, timestep, sample, return_dict=True, **_):
        ratio = self.num_train_timesteps // self.num_inference_steps
        t = int(timestep)
        prev = t - ratio
        if self.counter != 1:
            self.ets = self.ets[-3:]
            self.ets.append(model_output)
        else:
            prev, t = t, t + ratio
        if len(self.ets) == 1 and self.counter == 0:
            # diffusers keeps the caller's tensor itself (no clone): see __init__ for what that means for a caller that
            # writes the result back into the same storage.  Textbook PLMS keeps a copy.
            self.cur_sample = sample if self.cur_sample_aliases_latents else sample.clone()
        elif len(self.ets) == 1 and self.counter == 1:
            model_output = (model_output + self.ets[-1]) / 2
            sample = self.cur_sample
            self.cur_sample = None
        elif len(self.ets) == 2:
            model_output = (3 * self.ets[-1] - self.ets[-2]) / 2
        elif len(self.ets) == 3:
            model_output = (23 * self.ets[-1] - 16 * self.ets[-2] + 5 * self.ets[-3]) / 12
        else:
            model_output = (1 / 24) * (55 * self.ets[-1] - 59 * self.ets[-2] + 37 * self.ets[-3] - 9 * self.ets[-4])
        ca, cb = self._coeffs(t, prev)
        prev_sample = ca * sample + cb * model_output
        self.counter += 1
        return _Output(prev_sample) if return_dict else (prev_sample,)
