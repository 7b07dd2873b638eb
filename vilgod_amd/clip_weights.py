"""Weights of the CLIP ViT image tower as a flat name -> float32 tensor dict.

Two sources:
  * `load_state_dict(path)`: a real OpenAI checkpoint (`ViT-B-16.pt`, the file the reference's
    `ClipWrapper` loads, src/utils/clip_utils.py:19).  Keys `visual.*` are kept, prefix dropped.
  * `synthetic_vit_weights(seed, ...)`: seeded random weights of the same architecture
    (no checkpoint can be downloaded in the build/bench environment).  Scales follow the
    reference's own initialisers (third_party/CLIP/clip/model.py:213-221, 316-323).  The
    generator is numpy Philox, i.e. bit-reproducible on every host, so the GPU box and the
    build container derive identical tensors from (seed, shape).

Names are exactly the reference state_dict names (model.py:206-221, 171-183) so a real
checkpoint drops in.
"""
import numpy as np
import torch

VIT_B16 = dict(width=768, layers=12, heads=12, patch=16, resolution=224, output_dim=512)


def _gen(seed):
    return np.random.Generator(np.random.Philox(key=int(seed)))


def synthetic_vit_weights(seed=0, width=768, layers=12, heads=12, patch=16, resolution=224, output_dim=512):
    g = _gen(seed)

    def normal(shape, std):
        return torch.from_numpy((g.standard_normal(shape, dtype=np.float32) * np.float32(std)))

    scale = width ** -0.5
    tokens = (resolution // patch) ** 2 + 1
    wd = {
        'conv1.weight': normal((width, 3, patch, patch), (3 * patch * patch) ** -0.5),
        'class_embedding': normal((width,), scale),
        'positional_embedding': normal((tokens, width), scale),
        'ln_pre.weight': 1.0 + normal((width,), 0.05),
        'ln_pre.bias': normal((width,), 0.02),
        'ln_post.weight': 1.0 + normal((width,), 0.05),
        'ln_post.bias': normal((width,), 0.02),
        'proj': normal((width, output_dim), scale),
    }
    proj_std = scale * (2 * layers) ** -0.5
    fc_std = (2 * width) ** -0.5
    for l in range(layers):
        p = f'transformer.resblocks.{l}.'
        wd[p + 'ln_1.weight'] = 1.0 + normal((width,), 0.05)
        wd[p + 'ln_1.bias'] = normal((width,), 0.02)
        wd[p + 'attn.in_proj_weight'] = normal((3 * width, width), scale)
        wd[p + 'attn.in_proj_bias'] = normal((3 * width,), 0.02)
        wd[p + 'attn.out_proj.weight'] = normal((width, width), proj_std)
        wd[p + 'attn.out_proj.bias'] = normal((width,), 0.02)
        wd[p + 'ln_2.weight'] = 1.0 + normal((width,), 0.05)
        wd[p + 'ln_2.bias'] = normal((width,), 0.02)
        wd[p + 'mlp.c_fc.weight'] = normal((4 * width, width), fc_std)
        wd[p + 'mlp.c_fc.bias'] = normal((4 * width,), 0.02)
        wd[p + 'mlp.c_proj.weight'] = normal((width, 4 * width), proj_std)
        wd[p + 'mlp.c_proj.bias'] = normal((width,), 0.02)
    return wd


def synthetic_text_features(seed=0, n_classes=24, dim=512):
    """Stand-in for ClipWrapper's normalised text features (clip_utils.py:22-26): seeded unit
    vectors [n_classes, dim] float32."""
    g = _gen(seed + 7919)
    t = torch.from_numpy(g.standard_normal((n_classes, dim), dtype=np.float32))
    return t / t.norm(dim=-1, keepdim=True)


def synthetic_text_weights(seed=0, width=64, layers=2, embed=32, vocab=49408, ctx=77):
    """Seeded state dict of a (small) CLIP text tower with the key names of third_party/CLIP/clip/model.py:288-300 -- test
    vector generator for vilgod_amd/clip_text.py (the same tensors are loaded into the reference model when the golden
    features are made)."""
    g = _gen(seed + 104729)
    n = lambda *shape, std=1.0: torch.from_numpy((g.standard_normal(shape, dtype=np.float32) * std).astype(np.float32))
    sd = {'token_embedding.weight': n(vocab, width, std=0.02), 'positional_embedding': n(ctx, width, std=0.01),
          'ln_final.weight': 1 + n(width, std=0.1), 'ln_final.bias': n(width, std=0.1), 'text_projection': n(width, embed, std=width ** -0.5)}
    for i in range(layers):
        p = f'transformer.resblocks.{i}.'
        sd[p + 'ln_1.weight'], sd[p + 'ln_1.bias'] = 1 + n(width, std=0.1), n(width, std=0.1)
        sd[p + 'ln_2.weight'], sd[p + 'ln_2.bias'] = 1 + n(width, std=0.1), n(width, std=0.1)
        sd[p + 'attn.in_proj_weight'], sd[p + 'attn.in_proj_bias'] = n(3 * width, width, std=width ** -0.5), n(3 * width, std=0.02)
        sd[p + 'attn.out_proj.weight'], sd[p + 'attn.out_proj.bias'] = n(width, width, std=width ** -0.5), n(width, std=0.02)
        sd[p + 'mlp.c_fc.weight'], sd[p + 'mlp.c_fc.bias'] = n(4 * width, width, std=width ** -0.5), n(4 * width, std=0.02)
        sd[p + 'mlp.c_proj.weight'], sd[p + 'mlp.c_proj.bias'] = n(width, 4 * width, std=(4 * width) ** -0.5), n(width, std=0.02)
    return sd


def load_state_dict(path):
    """Real checkpoint: accepts a TorchScript archive (the published ViT-B-16.pt) or a plain
    state_dict; returns the `visual.*` tensors as float32 with the prefix dropped."""
    try:
        sd = torch.jit.load(path, map_location='cpu').state_dict()
    except RuntimeError:
        sd = torch.load(path, map_location='cpu')
    out = {k[len('visual.'):]: v.float() for k, v in sd.items() if k.startswith('visual.')}
    if 'proj' not in out:
        raise ValueError(f'{path}: not a CLIP ViT checkpoint (no visual.proj)')
    return out


def infer_config(wd):
    width = wd['conv1.weight'].shape[0]
    patch = wd['conv1.weight'].shape[-1]
    layers = len([k for k in wd if k.endswith('attn.in_proj_weight')])
    tokens = wd['positional_embedding'].shape[0]
    grid = int(round((tokens - 1) ** 0.5))
    return dict(width=width, layers=layers, heads=width // 64, patch=patch, resolution=grid * patch,
                output_dim=wd['proj'].shape[1])
