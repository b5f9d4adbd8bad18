"""Checkpoint I/O with mmcv's on-disk layout, so the reference's released `.pth` files (README.md:81-93) and its
`tools/test.py:172` flow (`load_checkpoint(model, path, map_location='cpu')`) carry over: a checkpoint is a dict
`{'meta': {...}, 'state_dict': OrderedDict, 'optimizer': ...}`; keys may carry DDP's `module.` prefix.  Parameter names
and shapes are the reference's (tests/test_host_logic.py), so no key translation is needed."""
import re
import time
from collections import OrderedDict

import torch


def _state_dict_of(checkpoint):
    if not isinstance(checkpoint, dict):
        raise RuntimeError(f'No state_dict found in checkpoint object of type {type(checkpoint)}')
    sd = checkpoint['state_dict'] if 'state_dict' in checkpoint else checkpoint
    return sd


def load_state_dict(module, state_dict, strict=False, logger=None):
    """mmcv.runner.load_state_dict: copy what matches, REPORT (not raise, unless strict) the rest."""
    own = module.state_dict()
    unexpected = [k for k in state_dict if k not in own]
    missing = [k for k in own if k not in state_dict and not k.endswith('num_batches_tracked')]
    mismatched = [(k, tuple(state_dict[k].shape), tuple(own[k].shape)) for k in state_dict
                  if k in own and tuple(state_dict[k].shape) != tuple(own[k].shape)]
    good = OrderedDict((k, v) for k, v in state_dict.items()
                       if k in own and tuple(v.shape) == tuple(own[k].shape))
    module.load_state_dict(good, strict=False)
    msgs = []
    if unexpected:
        msgs.append('unexpected key in source state_dict: ' + ', '.join(unexpected))
    if missing:
        msgs.append('missing keys in source state_dict: ' + ', '.join(missing))
    for k, a, b in mismatched:
        msgs.append(f'size mismatch for {k}: copying a param with shape {a} from checkpoint, the shape in current model is {b}')
    if msgs:
        text = 'The model and loaded state dict do not match exactly\n' + '\n'.join(msgs)
        if strict:
            raise RuntimeError(text)
        (logger.warning if logger is not None else print)(text)
    return dict(unexpected=unexpected, missing=missing, mismatched=mismatched)


def load_checkpoint(model, filename, map_location=None, strict=False, logger=None,
                    revise_keys=((r'^module\.', ''),)):
    """mmcv.runner.load_checkpoint(model, filename, map_location, strict, logger, revise_keys) -> checkpoint dict."""
    checkpoint = torch.load(filename, map_location=map_location, weights_only=False)
    sd = _state_dict_of(checkpoint)
    for pat, rep in revise_keys:
        sd = OrderedDict((re.sub(pat, rep, k), v) for k, v in sd.items())
    load_state_dict(model, sd, strict, logger)
    return checkpoint


def save_checkpoint(model, filename, optimizer=None, meta=None):
    """mmcv.runner.save_checkpoint layout (weights moved to the CPU, DDP prefix stripped)."""
    meta = dict(meta or {})
    meta.setdefault('time', time.asctime())
    meta.setdefault('fcaf3d_amd_version', __import__('fcaf3d_amd').__version__)
    module = model.module if hasattr(model, 'module') else model
    sd = OrderedDict((k, v.detach().cpu()) for k, v in module.state_dict().items())
    ckpt = {'meta': meta, 'state_dict': sd}
    if optimizer is not None:
        ckpt['optimizer'] = optimizer.state_dict()
    torch.save(ckpt, filename)
    return ckpt
