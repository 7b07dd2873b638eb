"""Why is the host-input run slower than the resident one?  Three variants of the same 96-frame block (development aid):
resident (four device tensors cycled), device copies (a fresh device tensor per frame: allocation, no PCIe), pinned host tensors."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vilgod_amd import synthetic
from vilgod_amd.pipeline import PseudoLabelPipeline
K = int(sys.argv[1]) if len(sys.argv) > 1 else 96
dev = torch.device('cuda:0')
pipe = PseudoLabelPipeline(device=dev, max_points=151_024, clip_model_path='/nonexistent')
frames = [pipe.upload(synthetic.make_frame(1 + i, 150_000, n_objects=60)) for i in range(4)]
host = [f.cpu().pin_memory() for f in frames]
poses = synthetic.make_poses(K + 40)
order = [(i // 6) % 4 for i in range(24)]
pipe.process_frames([frames[c] for c in order], [poses[1 + (i % 4)] for i in range(24)], poses[0], n_workers=6)
def block(kind):
    if kind == 'resident':
        fr = [frames[i % 4] for i in range(K)]
    elif kind == 'device copies':
        fr = None
    else:
        fr = [host[i % 4] for i in range(K)]
    pipe.new_sequence()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    if fr is None:
        fr = [frames[i % 4].clone() for i in range(K)]
    pipe.process_frames(fr, [poses[8 + i] for i in range(K)], poses[0], n_workers=6)
    torch.cuda.synchronize()
    return time.perf_counter() - t0
for rep in range(3):
    for kind in ('resident', 'device copies', 'pinned host'):
        dt = block(kind)
        print(f'rep {rep} {kind:14s}: {1000 * dt / K:.2f} ms per frame = {K / dt:.1f} frames/s', flush=True)
