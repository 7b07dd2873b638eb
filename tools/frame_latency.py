"""Latency of ONE 150k-point frame on an otherwise idle GPU, stage by stage (pipeline.process_frame(timing=True): a device
synchronisation after every stage, so the figures add up to the frame's latency and are NOT what a stage costs inside the stream).
    python tools/frame_latency.py [n_frames=6]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault('GPU_MAX_HW_QUEUES', '16')
import numpy as np, torch
from vilgod_amd import synthetic
from vilgod_amd.pipeline import PseudoLabelPipeline


def frame_latency(pipe, frames, poses, warm=2):
    """-> ({stage: median ms}, total ms) over frames[warm:]; frames = pinned host clouds."""
    rows = []
    pipe.new_sequence()
    for i, f in enumerate(frames):
        torch.cuda.synchronize()
        pipe.process_frame(f, poses[1 + i], poses[0], fnr=i, timing=True)
        if i >= warm:
            rows.append(dict(pipe.latency))
    keys = list(rows[0].keys())
    med = {k: round(1e3 * float(np.median([r.get(k, 0.0) for r in rows])), 3) for k in keys}
    return med, round(sum(med.values()), 3)


if __name__ == '__main__':
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 6
    dev = torch.device('cuda:0')
    pipe = PseudoLabelPipeline(device=dev, max_points=151_024, clip_model_path='/nonexistent')
    poses = synthetic.make_poses(n + 4)
    frames = [torch.from_numpy(synthetic.make_frame(500 + i, 150_000, n_objects=60)).pin_memory() for i in range(n)]
    med, tot = frame_latency(pipe, frames, poses)
    for k, v in med.items():
        print(f'{k:28s} {v:8.3f} ms')
    print(f'{"total":28s} {tot:8.3f} ms   (front stage = everything before encode+scores: '
          f'{sum(v for k, v in med.items() if k not in ("encode+scores", "scores_d2h+box_wait", "vote+results")):.3f} ms)')
