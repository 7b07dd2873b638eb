"""Interleaved A/B of whole-pipeline variants in ONE process (guide rule 24: separate processes / boxes differ by +-5 %).

    python tools/ab_pipeline.py [K=48] [ROUNDS=3] name1:ENV=V,ENV2=V2 name2:ENV=V ...

Every variant is a PseudoLabelPipeline built while its environment settings are in force (the switches this tool is for are read when
the handles are created: VG_GEMM_RI, VILGOD_PATCH_1CH, VG_VIT_RESID16, VG_VIT_CLS_LAST, ... -- per-launch switches such as VG_ATT_TR are
set around the variant's blocks as well; AB_WORKERS=n = frames in flight of the variant, default 6).  The same K distinct 150k-point frames (resident) go through every variant, ROUNDS times,
variants interleaved; prints frames/s per block and the median per variant, plus the per-launch time of the projection GEMMs of one
sequential pass (event pairs, vg_vit_profile)."""
import os
import statistics
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault('GPU_MAX_HW_QUEUES', '16')
import torch  # noqa: E402
from vilgod_amd import synthetic  # noqa: E402
from vilgod_amd.pipeline import PseudoLabelPipeline  # noqa: E402

args = sys.argv[1:]
nums = [a for a in args if a.isdigit()]
K = int(nums[0]) if nums else 48
ROUNDS = int(nums[1]) if len(nums) > 1 else 3
specs = [a for a in args if not a.isdigit()] or ['default:']
variants = []
for sp in specs:
    name, _, envs = sp.partition(':')
    env = dict(e.split('=', 1) for e in envs.split(',') if e)
    variants.append((name, env))


class Env:
    def __init__(self, env):
        self.env, self.old = env, {}

    def __enter__(self):
        for k, v in self.env.items():
            self.old[k] = os.environ.get(k)
            os.environ[k] = v

    def __exit__(self, *a):
        for k, v in self.old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


dev = torch.device('cuda:0')
poses = synthetic.make_poses(K + 8)
pipes = {}
for name, env in variants:
    with Env(env):
        pipes[name] = PseudoLabelPipeline(device=dev, max_points=151_024, clip_model_path='/nonexistent')
frames = [pipes[variants[0][0]].upload(synthetic.make_frame(100 + i, 150_000, n_objects=60)) for i in range(K)]


def block(name, env, n=K):
    p = pipes[name]
    with Env(env):
        p.new_sequence()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        p.process_frames(frames[:n], [poses[1 + i] for i in range(n)], poses[0], n_workers=int(os.environ.get('AB_WORKERS', '6')))
        torch.cuda.synchronize()
        return time.perf_counter() - t0


for name, env in variants:                 # warm-up: allocator, workspaces, a fresh box's slow first second
    for _ in range(2):
        block(name, env, min(K, 24))
res = {name: [] for name, _ in variants}
for r in range(ROUNDS):
    for name, env in variants:
        dt = block(name, env)
        res[name].append(K / dt)
        print(f'round {r} {name:16s}: {1000 * dt / K:7.2f} ms per frame = {K / dt:6.1f} frames/s', flush=True)
for name, env in variants:
    p = pipes[name]
    enc = p.clip.encoder
    with Env(env):
        p.new_sequence()
        enc.profile(True)
        for i in range(4):
            p.process_frame(frames[i], poses[1 + i], poses[0], fnr=i)
        torch.cuda.synchronize()
        n, ms, fl = enc.profile_read(1)
        enc.profile(False)
    print(f'{name:16s}: median {statistics.median(res[name]):6.1f} frames/s (min {min(res[name]):.1f}, max {max(res[name]):.1f}); '
          f'projection GEMMs {n // 4} launches, {ms / 4:.2f} ms per frame, {1000 * ms / max(n, 1):.1f} us per launch, {fl / ms / 1e9:.0f} TF', flush=True)
