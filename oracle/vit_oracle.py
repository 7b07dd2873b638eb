"""CPU oracle for CLIP ViT image encoding + zero-shot scoring + view voting
(SURVEY §8a rows D7, D9, D10).

TEST INFRASTRUCTURE ONLY (tests/, smoke(), bench.py cpu_baseline).  Plain torch fp32 on CPU,
functional (no nn.Module), operating on the same flat weight dict the product uploads to the
GPU, so the two sides share nothing but data.

Pinned against the reference: tests/golden/make_golden.py loads the reference's
third_party/CLIP/clip/model.py (standalone), copies a seeded weight dict into its
`VisionTransformer`, runs it, and freezes input/output; tests/test_oracle_vit.py checks this
restatement against those vectors.

Reference lines restated (relative to /root/reference):
  D7  third_party/CLIP/clip/model.py:157-163 (LayerNorm in fp32), :166-168 (QuickGELU),
      :171-192 (ResidualAttentionBlock; nn.MultiheadAttention packed in_proj, heads of 64),
      :206-240 (VisionTransformer.forward), :340-341 (encode_image)
  D9  src/utils/clip_utils.py:39-61  (L2 norm, 100*cos, softmax, top-1 via argpartition)
  D10 src/vilgod/lidar_frame.py:260-291 (majority vote, tie -> best mean score)
      src/vilgod/zero_shot_detector.py:412-415 (24 fine classes -> 4 names)
"""
import numpy as np
import torch
import torch.nn.functional as F


def layer_norm(x, w, b):
    # model.py:157-163: computed in float32, eps = nn.LayerNorm default 1e-5
    return F.layer_norm(x.float(), (x.shape[-1],), w, b, 1e-5)


def quick_gelu(x):
    return x * torch.sigmoid(1.702 * x)  # model.py:166-168


def vit_forward(wd, x, heads):
    """wd: flat dict (names as in the reference state_dict, prefix 'visual.' dropped).
    x: [n,3,H,W] float32.  Returns [n,output_dim]."""
    width = wd['conv1.weight'].shape[0]
    patch = wd['conv1.weight'].shape[-1]
    layers = len([k for k in wd if k.endswith('attn.in_proj_weight')])
    x = F.conv2d(x, wd['conv1.weight'], stride=patch)               # model.py:224
    n = x.shape[0]
    x = x.reshape(n, width, -1).permute(0, 2, 1)                    # :225-226
    cls = wd['class_embedding'].reshape(1, 1, width).expand(n, 1, width)
    x = torch.cat([cls, x], dim=1) + wd['positional_embedding']     # :227-228
    x = layer_norm(x, wd['ln_pre.weight'], wd['ln_pre.bias'])       # :229
    T = x.shape[1]
    dh = width // heads
    for l in range(layers):
        p = f'transformer.resblocks.{l}.'
        h = layer_norm(x, wd[p + 'ln_1.weight'], wd[p + 'ln_1.bias'])
        qkv = h @ wd[p + 'attn.in_proj_weight'].t() + wd[p + 'attn.in_proj_bias']
        q, k, v = qkv.split(width, dim=-1)
        q = q.reshape(n, T, heads, dh).transpose(1, 2) * (dh ** -0.5)
        k = k.reshape(n, T, heads, dh).transpose(1, 2)
        v = v.reshape(n, T, heads, dh).transpose(1, 2)
        a = torch.softmax(q @ k.transpose(-1, -2), dim=-1) @ v
        a = a.transpose(1, 2).reshape(n, T, width)
        x = x + (a @ wd[p + 'attn.out_proj.weight'].t() + wd[p + 'attn.out_proj.bias'])   # :190
        h = layer_norm(x, wd[p + 'ln_2.weight'], wd[p + 'ln_2.bias'])
        h = quick_gelu(h @ wd[p + 'mlp.c_fc.weight'].t() + wd[p + 'mlp.c_fc.bias'])
        x = x + (h @ wd[p + 'mlp.c_proj.weight'].t() + wd[p + 'mlp.c_proj.bias'])        # :191
    x = layer_norm(x[:, 0, :], wd['ln_post.weight'], wd['ln_post.bias'])                   # :235
    return x @ wd['proj']                                                                  # :238


def encode_in_chunks(wd, x, heads, split_size=50):
    """clip_utils.py:37-44 -- the reference encodes in chunks of `split_size`."""
    outs = [vit_forward(wd, c, heads) for c in torch.split(x, split_size)]
    return torch.cat(outs, dim=0) if outs else torch.zeros(0, wd['proj'].shape[1])


def clip_probabilities(features, text_features):
    """clip_utils.py:42-43.  text_features are already L2-normalised (clip_utils.py:26)."""
    f = features / features.norm(dim=-1, keepdim=True)
    return (100.0 * f @ text_features.T).softmax(dim=-1)


def top1(probs):
    """clip_utils.py:51-61 with top_k=1: np.argpartition(score,-1)[-1:] == index of the max
    (first-occurrence ties are not specified by argpartition; we report argmax and the tests
    avoid exact ties)."""
    p = probs.numpy() if torch.is_tensor(probs) else probs
    idx = np.array([np.argpartition(r, -1)[-1:][0] for r in p], dtype=np.int64) if len(p) else np.zeros(0, np.int64)
    return idx, p[np.arange(len(p)), idx]


def vote(class_names, class_scores):
    """lidar_frame.py:269-285 for ONE detection: class_names (V,) str array, class_scores (V,)
    -> (name, score)."""
    names, counts = np.unique(class_names, return_counts=True)
    if sum((counts[np.argmax(counts)]) == counts) > 1:
        name_max, max_score = None, 0
        for name in names:
            score = np.mean(class_scores[class_names == name])
            if score > max_score:
                max_score, name_max = score, name
        return name_max, max_score
    name = names[np.argmax(counts)]
    return name, np.mean(class_scores[class_names == name])
