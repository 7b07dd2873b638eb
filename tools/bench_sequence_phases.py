"""Wall time of the phases of PseudoLabelPipeline.process_sequence (default configuration, N frames in flight)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vilgod_amd import synthetic
from vilgod_amd.pipeline import PseudoLabelPipeline, default_preprocessor_cfg
from vilgod_amd.entropy import EntropyScorer, TwoFrameClusterer, full_scores
n_frames = int(os.environ.get('FRAMES', '48')); nw = int(os.environ.get('WORKERS', '6'))
frames, poses = synthetic.make_sequence(seed=0, n_frames=n_frames, n_points=150000, n_objects=60)
pipe = PseudoLabelPipeline(default_preprocessor_cfg(), device='cuda:0', vit_dtype='f16', clip_model_path='/nonexistent')
d_frames = [pipe.upload(f) for f in frames]
pipe.process_sequence(d_frames[:8], poses[:8], poses[0], n_workers=nw)
sync = torch.cuda.synchronize
for rep in range(2):
    sync(); t0 = time.perf_counter()
    pipe.new_sequence()
    prepared = [pipe.prepare(p, poses[i], poses[0], fnr=i) for i, p in enumerate(d_frames)]
    sync(); t1 = time.perf_counter()
    X = [p[2] for p in prepared]
    scorer = EntropyScorer(pipe.cluster_model)
    mapper = lambda items, fn: pipe.map_workers(items, lambda w, it: fn(w.cluster_model, it), nw)
    H = scorer.score_sequence(X, mapper=mapper)
    sync(); t2 = time.perf_counter()
    ent = []
    for (fs, _, d_X, _), h in zip(prepared, H):
        fs.entropy_scores, fs.entropy_indices = scorer.reduce(h)
        ent.append(full_scores(d_X.shape[0], fs.entropy_scores, fs.entropy_indices, device=pipe.device))
    ent_host = [e.cpu().numpy() for e in ent]
    sync(); t3 = time.perf_counter()
    parts = TwoFrameClusterer(pipe.cluster_model, n_frames=2, seed=0).precompute_parts(X, ent, mapper=mapper)
    sync(); t4 = time.perf_counter()
    def job(w, i):
        fs, d_ref, d_X, gidx = prepared[i]
        labels, probs = TwoFrameClusterer(w.cluster_model, n_frames=2, seed=0, parts=parts).labels(i, X, ent)
        return w.label(fs, d_ref, d_X, gidx, labels, probs, entropy=ent_host[i])
    out = pipe.map_workers(range(n_frames), job, nw)
    sync(); t5 = time.perf_counter()
    ms = lambda a, b: 1000 * (b - a) / n_frames
    print(f'rep {rep}: {n_frames} frames, {nw} workers | ms per frame: prepare {ms(t0,t1):.2f}  entropy {ms(t1,t2):.2f}  reduce {ms(t2,t3):.2f}  '
          f'parts {ms(t3,t4):.2f}  cluster+label {ms(t4,t5):.2f}  total {ms(t0,t5):.2f} -> {n_frames/(t5-t0):.1f} frames/s')
