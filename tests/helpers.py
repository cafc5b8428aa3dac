"""Shared test helpers: golden loading and synthetic state_dicts."""
import os

import numpy as np
import torch
import torch.nn as nn

from frido_amd import holders
from frido_amd.synth import fill_tensor

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def golden(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)


def synth_sd(module: nn.Module, prefix, profile=None):
    """{prefix+name: filler tensor} for every parameter of a holder tree (profile: frido_amd.synth filler profile)."""
    return {prefix + k: torch.from_numpy(fill_tensor(prefix + k, v.shape, profile)) for k, v in module.state_dict().items()}


def unet_holder(cfg):
    root = nn.Module()
    holders.build_unet_params(root, cfg)
    return root


def vq_holder(cfg):
    root = nn.Module()
    holders.build_msvqgan_params(root, cfg["edconfig"], cfg["ddconfig"], cfg["n_embed"], cfg["embed_dim"])
    return root
