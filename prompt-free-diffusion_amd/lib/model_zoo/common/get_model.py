"""Model registry: `@register(name)` on a class, `get_model()(cfg)` to build it.

Same contract as the reference's lib/model_zoo/common/get_model.py:54-124: `get_model` is a
process-wide singleton; `__call__(cfg, verbose=True)` reads `cfg.type` / `cfg.args`, imports the
module that registers that type (chosen by type-name prefix), instantiates `cls(**args)`, then
optionally loads `cfg.pretrained` (or legacy `cfg.pth`) with `cfg.map_location` (default cpu)
and `cfg.strict_sd` (default True) from .pth / .ckpt / .safetensors.  Host-side Python only.
"""
import copy
import importlib
import os.path as osp

import torch

from ...log_service import print_log

# type-name prefix -> module (relative to lib.model_zoo) whose import registers the class
_PREFIX_TO_MODULE = (
    ("pfd", "pfd"),
    ("autoencoderkl", "autokl"),
    ("openai_unet", "openaimodel"),
    ("controlnet", "controlnet"),
    ("seecoder", "seecoder"),
    ("swin", "swin"),
)


def get_total_param(model):
    return sum(p.numel() for p in model.parameters())


def get_total_param_sum(model):
    with torch.no_grad():
        return float(sum(p.double().abs().sum().item() for p in model.parameters()))


class _GetModel:
    _instance = None

    def __init__(self):
        self.model = {}

    def register(self, model, name):
        self.model[name] = model

    def __call__(self, cfg, verbose=True):
        if cfg is None:
            return None
        t = cfg.type
        if t not in self.model:
            for prefix, modname in _PREFIX_TO_MODULE:
                if t.startswith(prefix):
                    importlib.import_module("." + modname, package=__package__.rsplit(".", 1)[0])
                    break
        if t not in self.model:
            raise KeyError(f"model type '{t}' is not registered")
        args = copy.deepcopy(cfg.args)
        if "backbone" in args:
            args.backbone = self(args.backbone)
        net = self.model[t](**args)

        pretrained = cfg.get("pretrained", None)
        if pretrained is None:  # legacy field name
            pretrained = cfg.get("pth", None)
        map_location = cfg.get("map_location", "cpu")
        strict_sd = cfg.get("strict_sd", True)
        if pretrained is not None:
            ext = osp.splitext(pretrained)[1]
            if ext == ".pth":
                sd = torch.load(pretrained, map_location=map_location)
            elif ext == ".ckpt":
                sd = torch.load(pretrained, map_location=map_location)["state_dict"]
            elif ext == ".safetensors":
                from safetensors.torch import load_file
                sd = dict(load_file(pretrained, map_location))
            else:
                raise ValueError(f"unknown checkpoint extension '{ext}' ({pretrained})")
            net.load_state_dict(sd, strict=strict_sd)
            if verbose:
                print_log("Load model from [{}] strict [{}].".format(pretrained, strict_sd))
        if verbose:
            print_log("Load {} with total {} parameters,{:.3f} parameter sum.".format(
                t, get_total_param(net), get_total_param_sum(net)))
        return net


def get_model():
    if _GetModel._instance is None:
        _GetModel._instance = _GetModel()
    return _GetModel._instance


def register(name):
    def wrapper(class_):
        get_model().register(class_, name)
        return class_
    return wrapper
