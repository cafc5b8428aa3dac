"""Reference-import harness (TEST INFRASTRUCTURE, this container only).

Imports the upstream reference from /root/reference on CPU so that golden vectors can be
captured from the reference's OWN modules.  Nothing from the
reference is copied: we only inject stub modules for packages the image lacks and patch the
hard-coded `.cuda()` calls (frido/models/diffusion/ddim.py:19-23,201; plms.py:18-22;
pyunet.py:892) so the code runs on CPU.  /root/reference does not exist on the GPU box, so
this file is only ever imported by tests/golden/make_golden.py.
"""
import sys
import types
import importlib

import torch
import torch.nn as nn

REF_ROOT = "/root/reference"


def _stub(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def install():
    """Make `import frido...` / `import taming...` resolve to /root/reference with stubs."""
    if getattr(install, "_done", False):
        return
    # our own repo ships `frido/` and `taming/` alias packages: make sure the reference wins here
    for k in [k for k in sys.modules if k.split(".")[0] in ("frido", "taming", "ldm")]:
        del sys.modules[k]
    sys.path = [REF_ROOT] + [p for p in sys.path if p not in ("", "/root/repo", REF_ROOT)]

    class LightningModule(nn.Module):
        @property
        def device(self):
            try:
                return next(self.parameters()).device
            except StopIteration:
                return torch.device("cpu")

        def log(self, *a, **k):
            pass

        def log_dict(self, *a, **k):
            pass

    class LightningDataModule:
        pass

    pl = _stub("pytorch_lightning", LightningModule=LightningModule,
               LightningDataModule=LightningDataModule, Callback=object, Trainer=object,
               seed_everything=lambda *a, **k: None)
    _stub("pytorch_lightning.utilities")
    _stub("pytorch_lightning.utilities.distributed", rank_zero_only=lambda f: f)
    _stub("pytorch_lightning.callbacks", ModelCheckpoint=object, Callback=object, LearningRateMonitor=object)
    _stub("pytorch_lightning.trainer")

    class ListConfig(list):
        pass

    class _OC:
        @staticmethod
        def load(*a, **k):
            raise RuntimeError("omegaconf stub")

    _stub("omegaconf", OmegaConf=_OC, ListConfig=ListConfig, DictConfig=dict)
    _stub("omegaconf.listconfig", ListConfig=ListConfig)
    tv = _stub("torchvision")
    tv.utils = _stub("torchvision.utils", make_grid=lambda *a, **k: None)
    tv.transforms = _stub("torchvision.transforms")
    _stub("kornia")
    torch.Tensor.cuda = lambda self, *a, **k: self
    nn.Module.cuda = lambda self, *a, **k: self
    install._done = True


def import_ref(modname):
    install()
    return importlib.import_module(modname)


def patch_samplers():
    ddim = import_ref("frido.models.diffusion.ddim")
    plms = import_ref("frido.models.diffusion.plms")
    for cls in (ddim.DDIMSampler, plms.PLMSSampler):
        cls.register_buffer = lambda s, n, a: setattr(s, n, a)
    return ddim.DDIMSampler, plms.PLMSSampler
