"""Deterministic synthetic weights / inputs (no checkpoints or datasets are reachable offline).

Every parameter is filled from a pure-numpy generator keyed by the parameter's NAME and shape, so
the reference modules (in tests/golden/make_golden.py), the oracle and the HIP path all see
bit-identical weights without any weight file being committed.  Layers that the
reference zero-initialises (pyunet.py:236-238,801; attention.py:276-280) are filled like any other
layer, otherwise the U-Net output would be identically zero.
"""
import zlib

import numpy as np

__all__ = ["fill_tensor", "fill_state_dict", "seeded_normal"]


def _rng(name):
    return np.random.default_rng(zlib.crc32(name.encode("utf-8")))


def fill_tensor(name, shape):
    """float32 array for parameter `name` (a state_dict key) of `shape`."""
    shape = tuple(int(s) for s in shape)
    rng = _rng(name)
    leaf = name.rsplit(".", 1)[-1]
    if len(shape) == 0:
        return np.asarray(1.0, dtype=np.float32)
    if "embedding.weight" in name or name.endswith("emb.weight") or "stage_emb" in name:
        # VQ codebooks / token / stage embeddings: O(1) entries so that codes are well separated
        return (rng.standard_normal(shape) * (0.5 if "stage_emb" in name else 1.0)).astype(np.float32)
    if leaf in ("in_proj_weight", "text_projection") and len(shape) == 2:      # OpenAI CLIP's bare matrices: fan-in scaling too
        return (rng.standard_normal(shape) / np.sqrt(shape[1] if leaf == "in_proj_weight" else shape[0])).astype(np.float32)
    if leaf == "weight" and len(shape) >= 2:
        fan_in = int(np.prod(shape[1:]))
        return (rng.standard_normal(shape) / np.sqrt(fan_in)).astype(np.float32)
    if leaf == "weight":  # norm scales
        return (1.0 + 0.1 * rng.standard_normal(shape)).astype(np.float32)
    if leaf == "bias":
        return (0.1 * rng.standard_normal(shape)).astype(np.float32)
    if leaf in ("g", "scale"):
        return (1.0 + 0.1 * rng.standard_normal(shape)).astype(np.float32)
    return (0.1 * rng.standard_normal(shape)).astype(np.float32)


def fill_state_dict(shapes, skip=()):
    """shapes: mapping name -> shape.  Returns name -> float32 ndarray."""
    out = {}
    for name, shape in shapes.items():
        if any(name.startswith(s) or s in name for s in skip):
            continue
        out[name] = fill_tensor(name, shape)
    return out


def seeded_normal(tag, shape):
    """Standard-normal float32 input tensor keyed by a string tag (test/bench inputs)."""
    return _rng("input:" + tag).standard_normal(tuple(shape)).astype(np.float32)


def fill_module(module, prefix=""):
    """Fill every parameter of an nn.Module in place from the deterministic filler (keyed by prefix+name)."""
    import torch
    with torch.no_grad():
        for name, p in module.named_parameters():
            p.copy_(torch.from_numpy(fill_tensor(prefix + name, p.shape)).to(p.device))
    if hasattr(module, "invalidate"):
        module.invalidate()
    return module
