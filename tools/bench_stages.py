"""Per-stage timing of the classification half of the hot path (render + ViT) on synthetic clusters.
Development aid; the contract benchmark is /bench.py.

    python tools/bench_stages.py [--crops 240] [--views 4] [--iters 10]
"""
import argparse
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

VIT_FLOP_PER_CROP = 2 * 17_563_453_440


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--clusters', type=int, default=60)
    ap.add_argument('--views', type=int, default=4)
    ap.add_argument('--iters', type=int, default=10)
    ap.add_argument('--dtype', default='f16')
    args = ap.parse_args()
    from vilgod_amd.projection import RealisticProjection, VIEWS_4, VIEWS_6
    from vilgod_amd.clip_wrapper import VitEncoder, clip_scores
    from vilgod_amd import clip_weights as cw
    dev = torch.device('cuda:0')
    rng = np.random.default_rng(0)
    clusters = []
    for i in range(args.clusters):
        P = int(np.clip(rng.lognormal(6.0, 1.0), 15, 20000))
        az, rg = rng.uniform(0, 6.28), rng.uniform(5, 60)
        c = np.array([rg * np.cos(az), rg * np.sin(az), 0.8])
        clusters.append((rng.normal(size=(P, 3)) * rng.uniform([0.3, 0.3, 0.5], [2.5, 1.2, 1.0]) + c).astype(np.float32))
    pts = torch.from_numpy(np.concatenate(clusters)).to(dev)
    seg = torch.from_numpy(np.concatenate([[0], np.cumsum([len(c) for c in clusters])]).astype(np.int32)).to(dev)
    proj = RealisticProjection({}, device=dev, views=VIEWS_4 if args.views == 4 else VIEWS_6)
    wd = cw.synthetic_vit_weights(0, **cw.VIT_B16)
    enc = VitEncoder(wd, dtype=args.dtype, device=dev)
    text = cw.synthetic_text_features(0, 24, 512).to(dev)
    n = args.clusters * args.views
    out_kind = 'f16' if args.dtype == 'f16' else 'f32'

    def step():
        crops = proj.render_frame(pts, None, seg, np.eye(4), out=out_kind)
        f = enc.encode(crops)
        return clip_scores(f, text)

    for _ in range(3):
        step()
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    t_r = t_v = 0.0
    for _ in range(args.iters):
        ev[0].record()
        crops = proj.render_frame(pts, None, seg, np.eye(4), out=out_kind)
        ev[1].record()
        f = enc.encode(crops)
        clip_scores(f, text)
        ev[2].record()
        torch.cuda.synchronize()
        t_r += ev[0].elapsed_time(ev[1])
        t_v += ev[1].elapsed_time(ev[2])
    t_r /= args.iters
    t_v /= args.iters
    print(f'points {pts.shape[0]}  crops {n}  dtype {args.dtype}')
    print(f'render : {t_r:8.3f} ms  ({n * 3 * 224 * 224 * 2 / t_r / 1e6:.1f} GB/s crop writes)')
    print(f'vit    : {t_v:8.3f} ms  ({n * VIT_FLOP_PER_CROP / t_v / 1e9:.1f} TFLOP/s)')
    print(f'frames/s (classification half only): {1000.0 / (t_r + t_v):.1f}')


if __name__ == '__main__':
    main()
