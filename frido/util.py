"""`frido.util` import path of the reference (frido/util.py): the config factory (-> frido_amd) plus the small host
helpers that scripts/sample_diffusion.py:16-19 and the model code import from here.  Own implementations written from the
signatures / documented behaviour (frido/util.py:10-72); none of them touches the device hot path."""
import inspect

import numpy as np
import torch

from frido_amd.models import instantiate_from_config, instantiate_from_config_main, get_obj_from_str  # noqa: F401


def exists(x):
    return x is not None


def default(val, d):
    """val unless it is None; a default given as a plain function is called (lazily built tensors)."""
    if val is not None:
        return val
    return d() if inspect.isfunction(d) else d


def ismap(x):
    """4-D tensor with more than three channels (a label / feature map rather than an image)."""
    return isinstance(x, torch.Tensor) and x.dim() == 4 and x.shape[1] > 3


def isimage(x):
    """4-D tensor with one or three channels."""
    return isinstance(x, torch.Tensor) and x.dim() == 4 and x.shape[1] in (1, 3)


def mean_flat(tensor):
    """Mean over every non-batch dimension -> shape [B]."""
    return tensor.reshape(tensor.shape[0], -1).mean(dim=1) if tensor.dim() > 1 else tensor


def count_params(model, verbose=False):
    n = sum(int(p.numel()) for p in model.parameters())
    if verbose:
        print(f"{type(model).__name__} has {n * 1.e-6:.2f} M params.")
    return n


def log_txt_as_img(wh, xc, size=10):
    """Render the captions `xc` as white (W, H) = wh RGB panels, returned as a float tensor [B, 3, H, W] in [-1, 1]
    (the reference's logging helper; PIL's built-in font is used when data/DejaVuSans.ttf is not next to the caller)."""
    from PIL import Image, ImageDraw, ImageFont
    try:
        font = ImageFont.truetype("data/DejaVuSans.ttf", size=size)
    except OSError:
        font = ImageFont.load_default()
    per_line = max(1, int(40 * (wh[0] / 256)))
    panels = []
    for cap in xc:
        if isinstance(cap, (list, tuple)):
            cap = ", ".join(repr(c) for c in cap)
        cap = str(cap)
        canvas = Image.new("RGB", tuple(wh), color="white")
        text = "\n".join(cap[i:i + per_line] for i in range(0, len(cap), per_line))
        try:
            ImageDraw.Draw(canvas).text((0, 0), text, fill="black", font=font)
        except UnicodeEncodeError:
            print("Cant encode string for logging. Skipping.")
        panels.append(np.asarray(canvas, dtype=np.float64).transpose(2, 0, 1) / 127.5 - 1.0)
    return torch.tensor(np.stack(panels))
