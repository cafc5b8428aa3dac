"""DDIMSampler / PLMSSampler with the reference's constructor and `.sample(...)` signature
(frido/models/diffusion/ddim.py:11-114, plms.py:11-114), driving the HIP SamplerEngine.

Noise: the reference draws x_T and one torch.randn per update from torch's global generator of the device it runs on.
With noise="torch" (default) the same draws are made from the HOST generator in the same order and uploaded, so a run after
torch.manual_seed(s) consumes exactly the noise stream of a reference run ON CPU with that seed (a reference run on a GPU
draws from that device's generator, a different stream); noise="philox" uses the device counter RNG keyed by
(seed, global sample index) instead (no host involvement, shard-invariant).
"""
import collections

import numpy as np
import torch

ENGINE_CACHE_SIZE = 4      # compiled SamplerEngines kept per denoiser (each owns per-stage plans, buffers and graphs)

from . import schedules
from ._lib import FridoHipError


class _SamplerBase:
    KIND = "ddim"

    def __init__(self, model, schedule="linear", **kwargs):
        self.model = model
        self.ddpm_num_timesteps = model.num_timesteps
        self.schedule = schedule

    def register_buffer(self, name, attr):
        setattr(self, name, attr)

    def make_schedule(self, ddim_num_steps, ddim_discretize="uniform", ddim_eta=0., verbose=True):
        if self.KIND == "plms" and ddim_eta != 0:
            raise ValueError("ddim_eta must be 0 for PLMS")
        ac = self.model.alphas_cumprod.detach().float().cpu().numpy()
        assert ac.shape[0] == self.ddpm_num_timesteps, "alphas have to be defined for each timestep"
        self.ddim_timesteps = schedules.make_ddim_timesteps(ddim_discretize, ddim_num_steps, self.ddpm_num_timesteps)
        sig, al, alp = schedules.make_ddim_sampling_parameters(ac, self.ddim_timesteps, ddim_eta)
        self.ddim_sigmas, self.ddim_alphas, self.ddim_alphas_prev = sig, al, alp
        self.ddim_sqrt_one_minus_alphas = np.sqrt((np.float32(1.0) - al).astype(np.float32))
        self.alphas_cumprod = ac

    def _engine(self, B, shape, nctx, S, eta, scale, num_stage, temperature, replica=0):
        from .runtime import SamplerEngine
        unet = self.model.model.diffusion_model
        rt = unet.runtime()
        C, H, W = shape
        key = (self.KIND, B, C, H, W, nctx, S, float(eta), scale != 1.0, num_stage, float(temperature), replica)
        cache = rt.__dict__.setdefault("_sampler_engines", collections.OrderedDict())
        if key in cache:
            cache.move_to_end(key)
        else:
            while len(cache) >= ENGINE_CACHE_SIZE:      # least-recently-used engine goes; it owns its plans' persistent buffers
                # (Builder.persist_scope) and graphs, so its HBM is released with it -- the activation pool is shared and reused
                cache.popitem(last=False)
            cache[key] = SamplerEngine(rt.builder_for(replica), unet.cfg, B=B, C=C, H=H, W=W, nctx=nctx, S=S, eta=eta, kind=self.KIND,
                                       alphas_cumprod=self.model.alphas_cumprod.detach().float().cpu().numpy(),
                                       embed_dim=self.model.embed_dim_list, cfg_scale=scale, num_stage=num_stage,
                                       temperature=temperature)
        eng = cache[key]
        eng.cfg_scale = float(scale)      # read from a device scalar by the captured step bodies: one graph, any scale
        return eng

    @torch.no_grad()
    def sample(self, S, batch_size, shape, conditioning=None, num_stage=1, callback=None, normals_sequence=None,
               img_callback=None, quantize_x0=False, eta=0., mask=None, x0=None, temperature=1., noise_dropout=0.,
               score_corrector=None, corrector_kwargs=None, verbose=True, x_T=None, log_every_t=100,
               unconditional_guidance_scale=1., unconditional_conditioning=None, noise="torch", seed=0, sample0=0,
               replica=0, **kwargs):
        if mask is not None or x0 is not None:
            # ddim.py:158-161 / plms.py blend `q_sample(x0, ts) * mask + (1 - mask) * img`, but img carries only the channels of the
            # stages reached so far: with num_stage > 1 -- every Frido model -- the reference itself raises a RuntimeError (size
            # mismatch) in stage 0 (full-channel x0) or in stage 1 (stage-0-channel x0); verified against the reference on CPU
            raise NotImplementedError("mask / x0 (inpainting): the reference's blend fails for multi-stage models (shape mismatch between "
                                      "x0 and the per-stage latent, ddim.py:158-161); not provided on the HIP path")
        if quantize_x0:
            raise NotImplementedError("quantize_x0: the reference calls exit() on this option (ddim.py:251-253)")
        if conditioning is None or isinstance(conditioning, dict):
            raise NotImplementedError("cross-attention conditioning tensor required")
        if conditioning.shape[0] != batch_size:
            # the reference only prints a warning here (ddim.py:87-93) and then fails (or silently broadcasts) inside the
            # denoiser; a mismatched batch is never what the caller meant
            raise ValueError(f"Got {conditioning.shape[0]} conditionings but batch-size is {batch_size}")
        if unconditional_conditioning is not None and unconditional_conditioning.shape != conditioning.shape:
            raise ValueError(f"unconditional_conditioning {tuple(unconditional_conditioning.shape)} must match "
                             f"conditioning {tuple(conditioning.shape)}")
        if not conditioning.is_cuda:
            raise FridoHipError("sample(): conditioning must live on the MI355X (there is no CPU path)")
        self.make_schedule(ddim_num_steps=S, ddim_eta=eta, verbose=verbose)
        if unconditional_guidance_scale != 1.:
            assert unconditional_conditioning is not None
        if verbose:
            print(f"Data shape for {self.KIND.upper()} sampling is {(batch_size, *shape)}, eta {eta}")
        self.num_stage = num_stage

        def go(noise_src):
            # the engine is looked up per attempt: a repeated run (autoplanes: the default plane format saturated) belongs to the
            # denoiser's NEW runtime on the bf16-pair build, with its own plans, buffers and graphs
            eng = self._engine(batch_size, tuple(shape), conditioning.shape[1], S, eta, unconditional_guidance_scale, num_stage,
                               temperature, replica)
            return eng.run(conditioning, unconditional_conditioning, x_T=x_T, noise=noise_src, seed=seed, sample0=sample0,
                           log_every_t=log_every_t, callback=callback, img_callback=img_callback, noise_dropout=noise_dropout,
                           score_corrector=score_corrector, corrector_kwargs=corrector_kwargs, model=self.model)
        from . import autoplanes
        return autoplanes.run(self.model.model.diffusion_model, go, f"{type(self).__name__}.sample", noise=noise)


class DDIMSampler(_SamplerBase):
    KIND = "ddim"


class PLMSSampler(_SamplerBase):
    KIND = "plms"
