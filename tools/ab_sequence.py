"""Interleaved A/B of PseudoLabelPipeline.process_sequence (the reference's default stage order: entropy scores + two-frame 5-D clustering,
bench.py `default_config_mode`) in ONE process:    python tools/ab_sequence.py [frames=48] [rounds=4] name1:ENV=V name2:ENV=V ...
Switches read per call (development library: VG_CLUSTER_SITOUT, VG_CLUSTER_FIRST_BATCH) or at handle creation (VG_ATT_STAGGER ...)."""
import os, statistics, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault('GPU_MAX_HW_QUEUES', '16')
import torch
from vilgod_amd import synthetic
from vilgod_amd.pipeline import PseudoLabelPipeline
args = sys.argv[1:]
nums = [a for a in args if a.isdigit()]
n_seq = int(nums[0]) if nums else 48
rounds = int(nums[1]) if len(nums) > 1 else 4
variants = [(sp.partition(':')[0], dict(e.split('=', 1) for e in sp.partition(':')[2].split(',') if e)) for sp in args if not sp.isdigit()] or [('default', {})]


class Env:
    def __init__(self, env): self.env, self.old = env, {}
    def __enter__(self):
        for k, v in self.env.items(): self.old[k] = os.environ.get(k); os.environ[k] = v
    def __exit__(self, *a):
        for k, v in self.old.items():
            if v is None: os.environ.pop(k, None)
            else: os.environ[k] = v


dev = torch.device('cuda:0')
pipes = {}
for name, env in variants:
    with Env(env):
        pipes[name] = PseudoLabelPipeline(device=dev, max_points=151_024, clip_model_path='/nonexistent')
frames, poses = synthetic.make_sequence(seed=0, n_frames=n_seq, n_points=150_000, n_objects=60)
frames = [pipes[variants[0][0]].upload(f) for f in frames]
res = {name: [] for name, _ in variants}
for r in range(rounds + 1):
    for name, env in variants:
        with Env(env):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            out = pipes[name].process_sequence(frames, poses, poses[0], n_workers=6)
            torch.cuda.synchronize(); dt = time.perf_counter() - t0
        if r:                                              # round 0 = warm-up
            res[name].append(n_seq / dt)
        print(f'round {r} {name:14s}: {1000 * dt / n_seq:6.2f} ms per frame = {n_seq / dt:5.1f} frames/s, labels per frame {sum(len(o[1]["name"]) for o in out) / n_seq:.1f}', flush=True)
for name, _ in variants:
    print(f'{name:14s}: median {statistics.median(res[name]):5.1f} frames/s (min {min(res[name]):.1f}, max {max(res[name]):.1f})')
