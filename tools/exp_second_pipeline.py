"""Is a second PseudoLabelPipeline on the same tower slower than the first?  (bench.py's information blocks run on second pipelines.)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vilgod_amd import synthetic
from vilgod_amd.pipeline import PseudoLabelPipeline
K = int(sys.argv[1]) if len(sys.argv) > 1 else 96
dev = torch.device('cuda:0')
pipe = PseudoLabelPipeline(device=dev, max_points=151_024, clip_model_path='/nonexistent')
frames = [pipe.upload(synthetic.make_frame(1 + i, 150_000, n_objects=60)) for i in range(4)]
poses = synthetic.make_poses(K + 40)
order = [(i // 6) % 4 for i in range(24)]
def prep(p):
    p.new_sequence()
    p.process_frames([frames[c] for c in order], [poses[1 + (i % 4)] for i in range(24)], poses[0], n_workers=6)
def block(p):
    p.new_sequence()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    p.process_frames([frames[i % 4] for i in range(K)], [poses[8 + i] for i in range(K)], poses[0], n_workers=6)
    torch.cuda.synchronize()
    return time.perf_counter() - t0
prep(pipe)
print('first pipeline     ', ' '.join(f'{K / block(pipe):.1f}' for _ in range(3)), flush=True)
p2 = PseudoLabelPipeline(device=dev, max_points=151_024, clip_model_path='/nonexistent', clip=pipe.clip)
prep(p2)
print('second, shared clip', ' '.join(f'{K / block(p2):.1f}' for _ in range(3)), flush=True)
print('first again        ', ' '.join(f'{K / block(pipe):.1f}' for _ in range(3)), flush=True)
p3 = PseudoLabelPipeline(device=dev, max_points=151_024, clip_model_path='/nonexistent', clip=pipe.clip, vit_graph=False)
prep(p3)
print('third, no graph    ', ' '.join(f'{K / block(p3):.1f}' for _ in range(3)), flush=True)
print('second again       ', ' '.join(f'{K / block(p2):.1f}' for _ in range(3)), flush=True)
