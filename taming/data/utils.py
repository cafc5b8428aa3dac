"""`taming.data.utils.custom_collate` (scripts/sample_diffusion.py:21; reference: taming/data/utils.py:162-202): the
DataLoader collate function of the layout datasets.  It is torch's default collation with ONE exception -- a per-sample
list of `Annotation` records (variable length, taming/data/helper_types.py) is passed through as a list of lists instead of
being transposed/stacked.  Own implementation on top of torch.utils.data.default_collate."""
import collections.abc

import numpy as np
import torch
from torch.utils.data import default_collate


def _is_annotation(obj):
    return isinstance(obj, tuple) and hasattr(obj, "_fields") and type(obj).__name__ == "Annotation"


def custom_collate(batch):
    elem = batch[0]
    if isinstance(elem, (str, bytes)):
        return batch
    if isinstance(elem, collections.abc.Mapping):
        return {key: custom_collate([sample[key] for sample in batch]) for key in elem}
    if isinstance(elem, tuple) and hasattr(elem, "_fields"):          # a namedtuple per sample: collate field-wise
        return type(elem)(*(custom_collate(list(field)) for field in zip(*batch)))
    if isinstance(elem, collections.abc.Sequence):
        if len(elem) > 0 and _is_annotation(elem[0]):
            return batch                                                # ragged annotation lists stay as they are
        if any(len(sample) != len(elem) for sample in batch):
            if any(len(s) > 0 and _is_annotation(s[0]) for s in batch):
                return batch
            raise RuntimeError("each element in list of batch should be of equal size")
        return [custom_collate(list(samples)) for samples in zip(*batch)]
    if isinstance(elem, np.ndarray) and elem.dtype.kind in "SUO":
        raise TypeError(f"custom_collate: cannot collate numpy arrays of dtype {elem.dtype}")
    return default_collate(batch)
