"""ORACLE (test infrastructure only): CPU restatement of the reference's schedules and of its
multi-stage DDIM / PLMS sampling loops, plus decode_first_stage.  Pinned against
tests/golden/schedules.npz and tests/golden/sampler_*.npz (captured from the reference's own
DDIMSampler / PLMSSampler with torch.manual_seed(23) and the recorded noise stream).
"""
import numpy as np
import torch
import torch.nn.functional as F


# ---- schedules (host, float64 numpy like the reference) -----------------------------------------
def make_betas(n=1000, linear_start=0.0015, linear_end=0.0155):
    """util.py:21-26 'linear': linspace(sqrt(s), sqrt(e), n, float64) ** 2."""
    return torch.linspace(linear_start ** 0.5, linear_end ** 0.5, n, dtype=torch.float64).numpy() ** 2


def alphas_cumprod_f32(betas):
    """frido.py:133-155: cumprod in float64, registered as float32 buffers."""
    return torch.tensor(np.cumprod(1.0 - betas, axis=0), dtype=torch.float32)


def ddim_timesteps(S, T=1000):
    """util.py:46-60 'uniform': range(0, T, T // S) + 1."""
    return np.asarray(list(range(0, T, T // S))) + 1


def ddim_params(ac32, ts, eta):
    """util.py:63-74 evaluated on the fp32 alphas_cumprod tensor exactly as ddim.py:43-51 does."""
    alphas = ac32[ts]
    alphas_prev = np.asarray([ac32[0]] + ac32[ts[:-1]].tolist())
    sigmas = eta * np.sqrt((1 - alphas_prev) / (1 - alphas) * (1 - alphas / alphas_prev))
    return np.asarray(sigmas), np.asarray(alphas), alphas_prev


class NoiseSource:
    """Either fresh torch.randn draws (seeded by the caller) or a replayed flat tape."""

    def __init__(self, tape=None):
        self.tape = None if tape is None else torch.as_tensor(tape, dtype=torch.float32).reshape(-1)
        self.pos = 0

    def __call__(self, shape):
        if self.tape is None:
            return torch.randn(shape)
        n = int(np.prod(shape))
        out = self.tape[self.pos:self.pos + n].reshape(shape).clone()
        assert out.numel() == n, "noise tape exhausted"
        self.pos += n
        return out


def _model_eps(apply_model, x, t, c, s, start, scale, uc):
    """ddim.py:194-226: eps of stage s zero-padded on the frozen channels, optional CFG mix."""
    def one(cond):
        e = apply_model(x, t, cond, s)
        return torch.cat((torch.zeros(e.size(0), start, e.size(2), e.size(3)), e), dim=1)
    e_t = one(c)
    if scale != 1.0:
        e_u = one(uc)
        e_t = e_u + scale * (e_t - e_u)
    return e_t


def _x_prev(x, e_t, a_t, a_prev, sigma_t, sqrt1m, start, noise, temperature=1.0, noise_dropout=0.0):
    """ddim.py:237-268: all coefficients are fp32 (B,1,1,1) tensors built with torch.full; noise_dropout: ddim.py:260-262
    (F.dropout of the scaled noise: draws its keep mask from torch's global generator right after the randn)."""
    b = x.shape[0]
    full = lambda v: torch.full((b, 1, 1, 1), float(v))
    a_t, a_prev, sigma_t, sqrt1m = full(a_t), full(a_prev), full(sigma_t), full(sqrt1m)
    pred_x0 = (x - sqrt1m * e_t) / a_t.sqrt()
    pred_x0[:, :start] = x[:, :start]
    dir_xt = (1. - a_prev - sigma_t ** 2).sqrt() * e_t
    nz = sigma_t * noise * temperature
    if noise_dropout > 0.:
        nz = F.dropout(nz, p=noise_dropout)
    x_prev = a_prev.sqrt() * pred_x0 + dir_xt + nz
    x_prev[:, :start] = pred_x0[:, :start]
    return x_prev, pred_x0


def _handoff(img, s, num_stage, embed):
    """ddim.py:177-185: block-mean the finished channels (num_stage-s-1) times and expand back."""
    c0, c1 = sum(embed[:s]), sum(embed[:s + 1])
    tmp = img[:, c0:c1].clone()
    for _ in range(num_stage - s - 1):
        tmp = F.avg_pool2d(tmp, 2, 2)
    for _ in range(num_stage - s - 1):
        tmp = F.interpolate(tmp, scale_factor=2, mode="nearest")
    img[:, c0:c1] = tmp
    return img


@torch.no_grad()
def ddim_sample(apply_model, ac32, S, shape, cond, splits, embed, num_stage, eta=0.0, scale=1.0, uc=None,
                noise=None, log_every_t=100, temperature=1.0, x_T=None, noise_dropout=0.0, score_corrector=None):
    """ddim.py:116-186.  apply_model(x, t, cond, stage) -> eps of that stage.  A supplied x_T is adopted as the finished
    stage-0 result (ddim.py:150-152: stage 0 and its hand-off are skipped).  score_corrector: callable (e_t, x, t, cond) -> e_t
    applied after the CFG mix (ddim.py:228-230: `score_corrector.modify_score(model, e_t, x, t, c, **kwargs)`)."""
    noise = noise or NoiseSource()
    ts = ddim_timesteps(S)
    sig, al, alp = ddim_params(ac32, ts, eta)
    sq1m = np.sqrt(1.0 - al)
    b = shape[0]
    img = noise(shape) if x_T is None else x_T.clone()
    img_tmp = img.clone()
    inter = {"x_inter": [img], "pred_x0": [img]}
    total = ts.shape[0]
    for s in range(num_stage):
        if s == 0:
            img = img[:, :sum(splits[:1])]
        else:
            img = torch.cat((img, img_tmp[:, sum(splits[:s]):sum(splits[:s + 1])]), dim=1)
        if x_T is not None and s == 0:
            continue
        start = sum(embed[:s])
        for i, step in enumerate(np.flip(ts)):
            index = total - i - 1
            t = torch.full((b,), int(step), dtype=torch.long)
            e_t = _model_eps(apply_model, img, t, cond, s, start, scale, uc)
            if score_corrector is not None:
                e_t = score_corrector(e_t, img, t, cond)
            img, pred_x0 = _x_prev(img, e_t, al[index], alp[index], sig[index], sq1m[index], start,
                                   noise(img.shape), temperature, noise_dropout)
            if index % log_every_t == 0 or index == total - 1:
                inter["x_inter"].append(img)
                inter["pred_x0"].append(pred_x0)
        if num_stage != 1:
            img = _handoff(img, s, num_stage, embed)
    return img, inter


@torch.no_grad()
def plms_sample(apply_model, ac32, S, shape, cond, splits, embed, num_stage, scale=1.0, uc=None, noise=None,
                log_every_t=100, x_T=None, score_corrector=None):
    """plms.py:116-303 (eta must be 0, plms.py:25-26; noise is still drawn every update; x_T: plms.py:150-152).
    score_corrector: callable (e_t, x, t, cond) -> e_t applied inside EVERY model evaluation (plms.py:236-238)."""
    noise = noise or NoiseSource()
    ts = ddim_timesteps(S)
    sig, al, alp = ddim_params(ac32, ts, 0.0)
    sq1m = np.sqrt(1.0 - al)
    b = shape[0]
    img = noise(shape) if x_T is None else x_T.clone()
    img_tmp = img.clone()
    inter = {"x_inter": [img], "pred_x0": [img]}
    total = ts.shape[0]
    time_range = np.flip(ts)
    for s in range(num_stage):
        if s == 0:
            img = img[:, :sum(splits[:1])]
        else:
            img = torch.cat((img, img_tmp[:, sum(splits[:s]):sum(splits[:s + 1])]), dim=1)
        if x_T is not None and s == 0:
            continue
        start = sum(embed[:s])
        old = []
        for i, step in enumerate(time_range):
            index = total - i - 1
            t = torch.full((b,), int(step), dtype=torch.long)
            t_next = torch.full((b,), int(time_range[min(i + 1, total - 1)]), dtype=torch.long)
            upd = lambda e: _x_prev(img, e, al[index], alp[index], sig[index], sq1m[index], start, noise(img.shape))

            def model_out(xx, tt):
                e = _model_eps(apply_model, xx, tt, cond, s, start, scale, uc)
                return score_corrector(e, xx, tt, cond) if score_corrector is not None else e
            e_t = model_out(img, t)
            if len(old) == 0:
                x_p, _ = upd(e_t)
                e_next = model_out(x_p, t_next)
                e_p = (e_t + e_next) / 2
            elif len(old) == 1:
                e_p = (3 * e_t - old[-1]) / 2
            elif len(old) == 2:
                e_p = (23 * e_t - 16 * old[-1] + 5 * old[-2]) / 12
            else:
                e_p = (55 * e_t - 59 * old[-1] + 37 * old[-2] - 9 * old[-3]) / 24
            img, pred_x0 = upd(e_p)
            old.append(e_t)
            if len(old) >= 4:
                old.pop(0)
            if index % log_every_t == 0 or index == total - 1:
                inter["x_inter"].append(img)
                inter["pred_x0"].append(pred_x0)
        if num_stage != 1:
            img = _handoff(img, s, num_stage, embed)
    return img, inter


@torch.no_grad()
def decode_first_stage(vq_decode_fn, z_in, scale_factor, embed):
    """frido.py:823-891 (adopted_scale_factor branch): per-scale 1/scale_factor then VQ decode."""
    z = z_in.clone()
    start = 0
    for i, e in enumerate(embed):
        z[:, start:start + e] *= 1. / scale_factor[i]
        start += e
    return vq_decode_fn(z)
