"""Host-side noise schedules (numpy).  Scalars only — nothing here touches image data.

Restates, with the reference's exact float32/float64 mix so tables are bit-identical
(pinned by tests/golden/schedules.npz):
  frido/modules/diffusionmodules/util.py:21-26   make_beta_schedule('linear')
  frido/models/diffusion/frido.py:127-155        register_schedule (float64 cumprod -> float32 buffers)
  frido/modules/diffusionmodules/util.py:46-74   make_ddim_timesteps / make_ddim_sampling_parameters
  frido/models/diffusion/ddim.py:25-54           DDIMSampler.make_schedule
"""
import numpy as np

COEF_ROW = 12   # must match csrc/misc.hip: a_t, a_prev, sigma, sqrt(1-a_t), ab0..ab3, den, pad


def _linspace(start, end, steps):
    """torch.linspace's float64 evaluation (what the reference's schedule is built with): symmetric — first half
    fma(step, i, start), second half fma(-step, steps-1-i, end) — with FUSED multiply-adds, emulated here through
    80-bit long doubles (the product step*i is exact in 64 mantissa bits for i < 2^11).  np.linspace differs in
    the last bit for ~15 % of the entries."""
    L = np.longdouble
    start, end = np.float64(start), np.float64(end)
    step = (end - start) / np.float64(steps - 1)
    i = np.arange(steps)
    lo = (L(step) * i.astype(L) + L(start)).astype(np.float64)
    hi = (L(end) - L(step) * (steps - 1 - i).astype(L)).astype(np.float64)
    return np.where(i < steps // 2, lo, hi)


def make_beta_schedule(schedule, n_timestep, linear_start=1e-4, linear_end=2e-2, cosine_s=8e-3):
    if schedule == "linear":
        return _linspace(linear_start ** 0.5, linear_end ** 0.5, n_timestep) ** 2
    if schedule == "sqrt_linear":
        return _linspace(linear_start, linear_end, n_timestep)
    if schedule == "sqrt":
        return _linspace(linear_start, linear_end, n_timestep) ** 0.5
    if schedule == "cosine":
        ts = np.arange(n_timestep + 1, dtype=np.float64) / n_timestep + cosine_s
        al = np.cos(ts / (1 + cosine_s) * np.pi / 2) ** 2
        al = al / al[0]
        return np.clip(1 - al[1:] / al[:-1], 0, 0.999)
    raise ValueError(f"schedule '{schedule}' unknown.")


def ddpm_tables(betas):
    """The float32 buffers DDPM.register_schedule registers (inference subset)."""
    betas = np.asarray(betas, dtype=np.float64)
    alphas = 1.0 - betas
    ac = np.cumprod(alphas, axis=0)
    ac_prev = np.append(1.0, ac[:-1])
    f32 = lambda a: np.asarray(a, dtype=np.float32)
    return dict(betas=f32(betas), alphas_cumprod=f32(ac), alphas_cumprod_prev=f32(ac_prev),
                sqrt_alphas_cumprod=f32(np.sqrt(ac)), sqrt_one_minus_alphas_cumprod=f32(np.sqrt(1.0 - ac)),
                log_one_minus_alphas_cumprod=f32(np.log(1.0 - ac)), sqrt_recip_alphas_cumprod=f32(np.sqrt(1.0 / ac)),
                sqrt_recipm1_alphas_cumprod=f32(np.sqrt(1.0 / ac - 1)))


def make_ddim_timesteps(method, num_ddim, num_ddpm):
    if method == "uniform":
        ts = np.asarray(list(range(0, num_ddpm, num_ddpm // num_ddim)))
    elif method == "quad":
        ts = (np.linspace(0, np.sqrt(num_ddpm * .8), num_ddim) ** 2).astype(int)
    else:
        raise NotImplementedError(f'There is no ddim discretization method called "{method}"')
    return ts + 1


def make_ddim_sampling_parameters(alphacums32, ddim_timesteps, eta):
    """alphacums32: float32 alphas_cumprod.  Returns (sigmas f64, alphas f32, alphas_prev f64) with the reference's
    exact float mix (util.py:63-74 evaluated on a float32 torch tensor and a float64 ndarray, as ddim.py:43-46 does):
    `ndarray / tensor` dispatches to Tensor.__rtruediv__ = reciprocal(tensor) * ndarray, so 1/(1 - alphas) is formed
    in FLOAT32 and only then widened; everything else is float64."""
    ac = np.asarray(alphacums32, dtype=np.float32)
    alphas = ac[ddim_timesteps]
    alphas_prev = np.asarray([ac[0]] + ac[ddim_timesteps[:-1]].tolist(), dtype=np.float64)
    recip = (np.float32(1.0) / (np.float32(1.0) - alphas)).astype(np.float32).astype(np.float64)
    sigmas = eta * np.sqrt(recip * (1 - alphas_prev) * (1 - alphas.astype(np.float64) / alphas_prev))
    return sigmas, alphas, alphas_prev


def sampler_coef_table(alphacums32, S, eta, plms=False):
    """float32 table [n_steps][COEF_ROW] in LOOP order (row i = i-th executed step, index = n-1-i) plus
    the DDPM timestep of every row.  PLMS rows carry the Adams-Bashforth weights of plms.py:285-301."""
    ts = make_ddim_timesteps("uniform", S, len(alphacums32))
    sig, al, alp = make_ddim_sampling_parameters(alphacums32, ts, eta)
    sq1m = np.sqrt((np.float32(1.0) - al).astype(np.float32))
    n = ts.shape[0]
    tab = np.zeros((n, COEF_ROW), dtype=np.float32)
    for i in range(n):
        idx = n - 1 - i
        tab[i, 0:4] = [np.float32(al[idx]), np.float32(alp[idx]), np.float32(sig[idx]), sq1m[idx]]
        if plms:
            k = min(i, 3)
            ab = {0: ([1, 1, 0, 0], 2), 1: ([3, -1, 0, 0], 2), 2: ([23, -16, 5, 0], 12), 3: ([55, -59, 37, -9], 24)}[k]
            tab[i, 4:8] = ab[0]
            tab[i, 8] = ab[1]
        else:
            tab[i, 4:9] = [1, 0, 0, 0, 1]
    return tab, np.flip(ts).copy()
