"""Batch-sharded sampling over the GPUs of one node.

Every image is independent end to end (GroupNorm / LayerNorm / attention are per sample), so the
global batch is partitioned contiguously over ranks, each rank runs the identical captured program on
its shard with noise keyed by (seed, GLOBAL sample index), and the decoded images are joined by ONE
RCCL all-gather over xGMI (`torch.distributed` backend "nccl" is RCCL on ROCm).  The reference's
equivalent is N share-nothing processes writing PNGs (scripts/sample_diffusion.py:88-100,435-448).
"""
import torch
import torch.distributed as dist


def shard_range(total, rank, world):
    """Contiguous [lo, hi) slice of `total` samples owned by `rank` (sizes differ by at most 1)."""
    base, rem = divmod(total, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def all_gather_images(local, total=None, group=None):
    """One collective joining per-rank image shards [b_r, ...] -> [sum b_r, ...] on every rank.
    Equal shards use all_gather_into_tensor (a single RCCL all-gather); ragged shards are padded to the
    largest shard first."""
    if not (dist.is_available() and dist.is_initialized()):
        return local
    world = dist.get_world_size(group)
    total = total if total is not None else local.shape[0] * world
    sizes = [shard_range(total, r, world) for r in range(world)]
    bmax = max(hi - lo for lo, hi in sizes)
    buf = local
    if local.shape[0] != bmax:
        buf = torch.zeros((bmax,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
        buf[: local.shape[0]] = local
    if local.is_cuda and dist.get_backend(group) == "gloo":
        # (r05) device tensors under the CPU backend -- two ranks sharing ONE GPU in tests/test_model_gpu.py, where RCCL has no second
        # device to talk to: the shards are staged through host memory; the RCCL path below is untouched
        host = torch.empty((world * bmax,) + tuple(local.shape[1:]), dtype=local.dtype)
        dist.all_gather_into_tensor(host, buf.contiguous().cpu(), group=group)
        out = host.to(local.device)
    else:
        out = torch.empty((world * bmax,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
        dist.all_gather_into_tensor(out, buf.contiguous(), group=group)
    if all(hi - lo == bmax for lo, hi in sizes):
        return out
    return torch.cat([out[r * bmax: r * bmax + (hi - lo)] for r, (lo, hi) in enumerate(sizes)], dim=0)


@torch.no_grad()
def sample_images(model, cond, *, S, eta=1.0, sampler="ddim", scale=1.0, uncond=None, seed=0, sample0=0, noise="philox",
                  num_stage=None, gather=True, total=None, log_every_t=10 ** 9, gather_dtype="float32", check_status=True):
    """cond: this rank's conditioning shard [b, nctx, cd] on the GPU.  Returns decoded images (gathered).
    gather_dtype "float32": (N, 3, H, W) f32 in [-1, 1] (what decode_first_stage returns; 25 MB / rank at 32 images);
    "uint8" / "uint8_pil": the (N, H, W, 3) uint8 images of scripts/sample_diffusion.py custom_to_np (:115-121) / custom_to_pil
    (:103-113), produced by the decoder's last epilogue -- the one all-gather then moves 6.3 MB / rank.
    check_status (r05): after the decode, read the library's sticky numerics word (frido_status_flags: one device sync per pass) and
    turn a saturated fp16 operand plane or a non-finite normalisation statistic into a FridoNumericsWarning."""
    from .samplers import DDIMSampler, PLMSSampler
    unet = model.model.diffusion_model
    cls = PLMSSampler if sampler == "plms" else DDIMSampler
    b = cond.shape[0]
    shape = (unet.in_channels, unet.image_size, unet.image_size)
    z, _ = cls(model).sample(S=S, batch_size=b, shape=shape, conditioning=cond, num_stage=num_stage or unet.num_stage,
                             eta=eta, verbose=False, unconditional_guidance_scale=scale, unconditional_conditioning=uncond,
                             noise=noise, seed=seed, sample0=sample0, log_every_t=log_every_t)
    if gather_dtype == "float32":
        img = model.decode_first_stage(z)
    elif gather_dtype in ("uint8", "uint8_pil"):
        img = model.decode_first_stage(z, to_uint8="pil" if gather_dtype == "uint8_pil" else "np")
    else:
        raise ValueError(f"gather_dtype {gather_dtype!r}: 'float32', 'uint8' or 'uint8_pil'")
    if check_status and img.is_cuda:
        from . import _lib
        _lib.warn_on_status("sample_images")
    return all_gather_images(img, total=total) if gather else img
