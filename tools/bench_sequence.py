"""Throughput of the reference's DEFAULT stage order (entropy scores + two-frame clustering) on a coherent
synthetic sequence of 150k-point frames, stage by stage (sequential, one stream)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vilgod_amd import synthetic
from vilgod_amd.pipeline import PseudoLabelPipeline, default_preprocessor_cfg
from vilgod_amd.entropy import EntropyScorer, TwoFrameClusterer, full_scores

n_frames = int(os.environ.get('FRAMES', '24'))
pts_n = int(os.environ.get('POINTS', '150000'))
frames, poses = synthetic.make_sequence(seed=0, n_frames=n_frames, n_points=pts_n, n_objects=60)
pipe = PseudoLabelPipeline(default_preprocessor_cfg(), device='cuda:0', vit_dtype='f16', clip_model_path='/nonexistent')
d_frames = [pipe.upload(f) for f in frames]
pipe.process_frame(d_frames[0], poses[0], poses[0])          # warm-up (allocations, first launches)
sync = torch.cuda.synchronize
for rep in range(2):
    pipe.new_sequence()
    sync(); t0 = time.perf_counter()
    prepared = [pipe.prepare(f, poses[i], poses[0], fnr=i) for i, f in enumerate(d_frames)]
    sync(); t1 = time.perf_counter()
    X = [p[2] for p in prepared]
    scorer = EntropyScorer(pipe.cluster_model)
    H = scorer.score_sequence(X)
    sync(); t2 = time.perf_counter()
    ent = []
    for (fs, _, d_X, _), h in zip(prepared, H):
        fs.entropy_scores, fs.entropy_indices = scorer.reduce(h)
        ent.append(full_scores(d_X.shape[0], fs.entropy_scores, fs.entropy_indices, device=pipe.device))
    sync(); t3 = time.perf_counter()
    two = TwoFrameClusterer(pipe.cluster_model)
    lab = [two.labels(i, X, ent) for i in range(n_frames)]
    sync(); t4 = time.perf_counter()
    out = [pipe.label(p[0], p[1], p[2], p[3], l[0], l[1], entropy=e.cpu().numpy()) for p, l, e in zip(prepared, lab, ent)]
    sync(); t5 = time.perf_counter()
    ms = lambda a, b: 1000.0 * (b - a) / n_frames
    print(f'rep {rep}: {n_frames} frames x {pts_n} pts, non-ground {sum(x.shape[0] for x in X) / n_frames:.0f}/frame | per frame ms: '
          f'ground+to_ref {ms(t0, t1):.2f}  entropy {ms(t1, t2):.2f}  reduce {ms(t2, t3):.2f}  two-frame clustering {ms(t3, t4):.2f}  '
          f'filter+classify+boxes {ms(t4, t5):.2f}  total {ms(t0, t5):.2f}  -> {1000.0 / ms(t0, t5):.1f} frames/s; '
          f'labels/frame {sum(len(o[1]["name"]) for o in out) / n_frames:.1f}, moving clusters/frame {sum(int((~o[0].static).sum()) for o in out) / n_frames:.1f}')

# ---- the same through PseudoLabelPipeline.process_sequence, sequential and with frames in flight ----
for nw in (1, 3):
    pipe.process_sequence(d_frames[:4], poses[:4], poses[0], n_workers=nw)          # warm-up (worker handles)
    sync(); t0 = time.perf_counter()
    out = pipe.process_sequence(d_frames, poses, poses[0], n_workers=nw)
    sync(); t1 = time.perf_counter()
    print(f'process_sequence n_workers={nw}: {1000.0 * (t1 - t0) / n_frames:.2f} ms per frame -> {n_frames / (t1 - t0):.1f} frames/s, '
          f'labels/frame {sum(len(o[1]["name"]) for o in out) / n_frames:.1f}')
