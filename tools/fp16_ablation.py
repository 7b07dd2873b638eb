"""Which rounding step of the fp16 tower carries its probability error against the fp32 tower?  (VERDICT r4 task 3.)

The fp16 tower (csrc/vit.hip, dtype 1) differs from the fp32 tower in SIX places where a value is rounded to fp16:
    W     the GEMM weights (gamma-folded for in_proj / c_fc, rounded once from the fp32 product)
    x16   the fp16 copy of the fp32 residual stream that the folded-LayerNorm GEMMs (in_proj, c_fc) multiply
    qkv   in_proj's output
    P     the softmax probabilities handed to the P.V matrix product
    att   the attention output (out_proj's operand)
    fc    c_fc's output after QuickGELU (c_proj's operand)
Everything else -- accumulation, LayerNorm statistics, the residual adds, softmax -- is fp32 in both towers.  This tool restates the
tower in torch fp32 with each rounding as a switch (same formulas as the kernels: raw residual times gamma-scaled weights, statistics
applied afterwards, c1 summed over the ROUNDED weights), runs it on crops the renderer produced from 150k-point frames, and prints

    all six roundings on            (should reproduce the real fp16 tower's error -- printed beside it)
    all on EXCEPT one, for each     (how much of the error that step carries)
    only one on, for each           (the step's error alone)

as max |p - p_fp32| over all crops x 24 prompts, and the relative L2 error of the features.  Measurement aid (torch matmuls on the GPU);
nothing here is product code.      python tools/fp16_ablation.py [n_frames=3] [max_crops_per_frame=400] [layers]

With the third argument `layers` (VERDICT r5 task 3b) the table is per LAYER instead of per step: all six roundings on in every block
EXCEPT block l (and: only in block l), and with the last k = 1..4 blocks unrounded -- which blocks' roundings carry the error, i.e. what
a higher-precision treatment of a few late blocks could buy."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import torch.nn.functional as F

STEPS = ('W', 'x16', 'qkv', 'P', 'att', 'fc')


def r16(t, on):
    return t.half().float() if on else t


def tower(wd, x, heads, on, blocks=None):
    """x: [n,3,224,224] fp32 normalised crops.  on: set of STEPS whose rounding is applied.  fp32 math otherwise.
    blocks: the residual blocks in which `on` applies (None = all, and the patch embedding; a set = those blocks only, embedding unrounded)."""
    on_all = set(on)
    W_ = 'W' in on and blocks is None
    width = wd['conv1.weight'].shape[0]
    patch = wd['conv1.weight'].shape[-1]
    layers = len([k for k in wd if k.endswith('attn.in_proj_weight')])
    x = F.conv2d(x, r16(wd['conv1.weight'], W_), stride=patch)
    n = x.shape[0]
    x = x.reshape(n, width, -1).permute(0, 2, 1)
    cls = wd['class_embedding'].reshape(1, 1, width).expand(n, 1, width)
    x = torch.cat([cls, x], dim=1) + wd['positional_embedding']
    x = F.layer_norm(x, (width,), wd['ln_pre.weight'], wd['ln_pre.bias'], 1e-5)
    T = x.shape[1]
    dh = width // heads

    def folded(x, g, b, Wt, bias):
        """ln(x) Wt^T + bias the way k_gemm_f16_pp64<LN = 1> computes it: rstd * (x16 (g.W)^T - mean c1) + c2."""
        Wf = r16(g[None, :] * Wt, W_)                                  # rounded once, from the fp32 product
        c1 = Wf.sum(dim=1)
        c2 = bias + Wt @ b
        mean = x.mean(dim=-1, keepdim=True)
        rstd = torch.rsqrt(((x - mean) ** 2).mean(dim=-1, keepdim=True) + 1e-5)
        return rstd * (r16(x, 'x16' in on) @ Wf.t() - mean * c1) + c2

    for l in range(layers):
        p = f'transformer.resblocks.{l}.'
        on = on_all if (blocks is None or l in blocks) else set()
        W_ = 'W' in on
        if l == 0:      # block 0: ln_1 is computed by the embedding kernel and stored as fp16 (an 'x16'-kind rounding), plain weights
            h = r16(F.layer_norm(x, (width,), wd[p + 'ln_1.weight'], wd[p + 'ln_1.bias'], 1e-5), 'x16' in on)
            qkv = h @ r16(wd[p + 'attn.in_proj_weight'], W_).t() + wd[p + 'attn.in_proj_bias']
        else:
            qkv = folded(x, wd[p + 'ln_1.weight'], wd[p + 'ln_1.bias'], wd[p + 'attn.in_proj_weight'], wd[p + 'attn.in_proj_bias'])
        qkv = r16(qkv, 'qkv' in on)
        q, k, v = qkv.split(width, dim=-1)
        q = q.reshape(n, T, heads, dh).transpose(1, 2) * (dh ** -0.5)
        k = k.reshape(n, T, heads, dh).transpose(1, 2)
        v = v.reshape(n, T, heads, dh).transpose(1, 2)
        s = q @ k.transpose(-1, -2)
        e = torch.exp(s - s.max(dim=-1, keepdim=True).values)          # the kernel: exp of the shifted scores, P.V, then 1 / sum in fp32
        a = (r16(e, 'P' in on) @ v) / e.sum(dim=-1, keepdim=True)
        a = r16(a.transpose(1, 2).reshape(n, T, width), 'att' in on)
        x = x + (a @ r16(wd[p + 'attn.out_proj.weight'], W_).t() + wd[p + 'attn.out_proj.bias'])
        h = folded(x, wd[p + 'ln_2.weight'], wd[p + 'ln_2.bias'], wd[p + 'mlp.c_fc.weight'], wd[p + 'mlp.c_fc.bias'])
        h = r16(h * torch.sigmoid(1.702 * h), 'fc' in on)
        x = x + (h @ r16(wd[p + 'mlp.c_proj.weight'], W_).t() + wd[p + 'mlp.c_proj.bias'])
    x = F.layer_norm(x[:, 0, :], (width,), wd['ln_post.weight'], wd['ln_post.bias'], 1e-5)
    return x @ wd['proj']


def probs_of(f, text):
    f = f / f.norm(dim=-1, keepdim=True)
    return (100.0 * f @ text.T).softmax(dim=-1)


def main():
    n_frames = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    cap = int(sys.argv[2]) if len(sys.argv) > 2 else 400
    per_layer = len(sys.argv) > 3 and sys.argv[3] == 'layers'
    from vilgod_amd import synthetic, clip_weights as cw
    from vilgod_amd.pipeline import PseudoLabelPipeline
    from vilgod_amd.clip_wrapper import VitEncoder, clip_scores
    dev = torch.device('cuda:0')
    torch.backends.cuda.matmul.allow_tf32 = False
    wd = cw.synthetic_vit_weights(0, **cw.VIT_B16)
    text = cw.synthetic_text_features(0, 24, 512).to(dev)
    wdd = {k: v.to(dev).float() for k, v in wd.items()}
    pipe = PseudoLabelPipeline(device=dev, vit_dtype='f16', max_points=151_024, clip_model_path='/nonexistent')
    enc16 = VitEncoder(wd, dtype='f16', device=dev)
    poses = synthetic.make_poses(n_frames + 2)
    rows = {}

    def note(name, f, p, f32, p32):
        r = rows.setdefault(name, [0.0, 0.0, 0.0])
        r[0] = max(r[0], float((p - p32).abs().max()))
        r[1] += float(((f - f32) ** 2).sum())
        r[2] += float((f32 ** 2).sum())
    n_crops = 0
    for fi in range(n_frames):
        pts = pipe.upload(synthetic.make_frame(300 + fi, 150_000, n_objects=60))
        fs, d_ref, d_X, gidx = pipe.prepare(pts, poses[fi + 1], poses[0], fnr=fi)
        labels, pr = pipe.cluster(d_X)
        from vilgod_amd.frame_state import pack_clusters
        ids, index, seg = pack_clusters(labels, pr, pipe.prob_threshold)
        d_index, d_seg = torch.from_numpy(index).to(dev), torch.from_numpy(seg).to(dev)
        plane = pipe.ground_plane(d_ref, gidx)
        valid, _ = pipe.filter(d_X, d_index, d_seg, plane)
        vrows = np.flatnonzero(valid.cpu().numpy())
        parts = [index[seg[c]:seg[c + 1]] for c in vrows]
        d_vi = torch.from_numpy(np.concatenate(parts)).to(dev)
        d_vs = torch.from_numpy(np.r_[0, np.cumsum([len(p) for p in parts])].astype(np.int32)).to(dev)
        crops = pipe.projection.render_frame(d_X, d_vi, d_vs, fs.transform_to_ego, out='f32')[:cap]
        patches = pipe.projection.render_frame(d_X, d_vi, d_vs, fs.transform_to_ego, out='patch16c1')[:crops.shape[0] * 196]
        n = crops.shape[0]
        n_crops += n
        with torch.no_grad():
            def run(on, blocks=None):
                return torch.cat([tower(wdd, c, 12, on, blocks) for c in torch.split(crops, 64)])
            f32 = run(set())
            p32 = probs_of(f32, text)
            f_real = enc16.encode_patches(patches, n).float()
            note('REAL fp16 tower (csrc/vit.hip)', f_real, probs_of(f_real, text), f32, p32)
            f_all = run(set(STEPS))
            note('emulation, all six roundings', f_all, probs_of(f_all, text), f32, p32)
            note('  (real tower vs emulation)', f_real, probs_of(f_real, text), f_all, probs_of(f_all, text))
            if per_layer:
                allb = set(range(12))
                for l in range(12):
                    f = run(set(STEPS), allb - {l})
                    note(f'all blocks but {l} (embedding unrounded)', f, probs_of(f, text), f32, p32)
                for l in range(12):
                    f = run(set(STEPS), {l})
                    note(f'only block {l}', f, probs_of(f, text), f32, p32)
                for k in (1, 2, 3, 4, 6):
                    f = run(set(STEPS), set(range(12 - k)))
                    note(f'last {k} blocks unrounded', f, probs_of(f, text), f32, p32)
                for k in (1, 2, 3, 4):
                    f = run(set(STEPS), set(range(k, 12)))
                    note(f'first {k} blocks unrounded', f, probs_of(f, text), f32, p32)
                f = run(set(STEPS), allb)
                note('all blocks, embedding unrounded', f, probs_of(f, text), f32, p32)
            else:
                for s in STEPS:
                    f = run(set(STEPS) - {s})
                    note(f'all but {s}', f, probs_of(f, text), f32, p32)
                for s in STEPS:
                    f = run({s})
                    note(f'only {s}', f, probs_of(f, text), f32, p32)
        print(f'frame {fi}: {n} crops', flush=True)
    print(f'\n{n_crops} crops of {n_frames} synthetic 150k-point frames, 24 prompts; reference = the same tower with no rounding (fp32)')
    print(f'{"variant":44s} {"max |dp|":>10s} {"features rel L2":>16s}')
    for name, (mp, num, den) in rows.items():
        print(f'{name:44s} {mp:10.2e} {np.sqrt(num / den):16.2e}')


if __name__ == '__main__':
    main()
