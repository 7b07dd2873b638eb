import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vilgod_amd import synthetic
from vilgod_amd.pipeline import PseudoLabelPipeline, default_preprocessor_cfg
n = 100
t0 = time.perf_counter()
frames, poses = synthetic.make_sequence(seed=5, n_frames=n, n_points=150000, n_objects=60)
print('gen', time.perf_counter() - t0)
pipe = PseudoLabelPipeline(default_preprocessor_cfg(), device='cuda:0', vit_dtype='f16', clip_model_path='/nonexistent')
d = [pipe.upload(f) for f in frames]
pipe.process_sequence(d[:4], poses[:4], poses[0], n_workers=3)
torch.cuda.synchronize(); t0 = time.perf_counter()
out = pipe.process_sequence(d, poses, poses[0], n_workers=3)
torch.cuda.synchronize(); dt = time.perf_counter() - t0
print(f'{n} frames N1 mode: {dt:.2f} s -> {n/dt:.1f} fps; peak mem {torch.cuda.max_memory_allocated()/2**30:.2f} GiB; labels/frame {sum(len(o[1]["name"]) for o in out)/n:.1f}; moving/frame {sum(int((~o[0].static).sum()) for o in out)/n:.1f}')
