"""TEST INFRASTRUCTURE (not product code): deterministic synthetic weights.

There are no pretrained checkpoints in the build container or on the GPU box, so the parity
harness fills every floating-point parameter of a model from a seeded generator, keyed by the
state-dict key (so results do not depend on key order or on which sub-model is built).  Zero-
initialised parameters of the reference (`zero_module`: ResBlock out conv, SpatialTransformer
proj_out, UNet head, ControlNet zero-convs / hint tail, PPE_MLP last layer) are re-randomised
too, otherwise half of the network is multiplied by 0 and parity is vacuous.

Used by oracle/make_golden.py (fills the *reference* modules), by tests/ (fills the oracle's
state dict and the HIP modules) and by bench.py / __graft_entry__.smoke().
"""
import zlib

import torch

_EMBED_TOKENS = ("init_query", "query_pos_embedding", "level_embed", "relative_position_bias_table")


def seeded_tensor(key, shape, seed=0):
    """fp32 CPU tensor for parameter `key` of the given shape"""
    g = torch.Generator().manual_seed((zlib.crc32(key.encode()) * 31 + 7919 * seed) & 0x7FFFFFFF)
    shape = tuple(shape)
    r = torch.randn(shape, generator=g, dtype=torch.float32)
    if any(tok in key for tok in _EMBED_TOKENS):
        return r * 0.5
    if len(shape) <= 1:
        if key.endswith("weight"):      # every 1-D weight on the path is a norm scale
            return 1.0 + 0.1 * r
        return 0.05 * r                  # biases
    fan_in = 1
    for s in shape[1:]:
        fan_in *= s
    return r * (fan_in ** -0.5)


def seeded_state_dict(spec, seed=0, prefix=""):
    """spec: {key: shape} of floating-point parameters -> {key: fp32 tensor}; keys are looked up
    with `prefix` stripped so a sub-model gets the same values it has inside the composite."""
    out = {}
    for k, shape in spec.items():
        if prefix and not k.startswith(prefix):
            continue
        out[k[len(prefix):]] = seeded_tensor(k, shape, seed)
    return out


def fill_module_(module, seed=0, prefix=""):
    """overwrite every floating-point nn.Parameter of `module` in place (buffers untouched)"""
    with torch.no_grad():
        for name, p in module.named_parameters():
            if p.is_floating_point():
                p.copy_(seeded_tensor(prefix + name, p.shape, seed).to(p.dtype))
    return module


def param_spec(module, prefix=""):
    return {prefix + n: list(p.shape) for n, p in module.named_parameters() if p.is_floating_point()}
