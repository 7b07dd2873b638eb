"""Stress run of the in-flight frame path (process_frames: worker threads + streams, captured ViT graphs, box helper processes) with a
watchdog: if an iteration stalls, the Python stacks of all threads are dumped (faulthandler) -- development aid for hangs.
    python tools/stress_inflight.py [iterations=30] [frames=40] [vit_graph=1] [box_mode=reference]"""
import faulthandler, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vilgod_amd import synthetic
from vilgod_amd.pipeline import PseudoLabelPipeline
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 30
K = int(sys.argv[2]) if len(sys.argv) > 2 else 40
graph = bool(int(sys.argv[3])) if len(sys.argv) > 3 else True
box_mode = sys.argv[4] if len(sys.argv) > 4 else 'reference'
dev = torch.device('cuda:0')
pipe = PseudoLabelPipeline(device=dev, max_points=151_024, clip_model_path='/nonexistent', vit_graph=graph, box_mode=box_mode)
frames = [pipe.upload(synthetic.make_frame(1 + i, 150_000, n_objects=60)) for i in range(4)]
poses = synthetic.make_poses(K + 8)
faulthandler.enable()
for it in range(iters):
    faulthandler.dump_traceback_later(45, repeat=False, exit=True)          # a healthy iteration takes < 2 s
    pipe.new_sequence()
    t0 = time.perf_counter()
    out = pipe.process_frames([frames[i % 4] for i in range(K)], [poses[i + 1] for i in range(K)], poses[0], n_workers=6)
    torch.cuda.synchronize()
    faulthandler.cancel_dump_traceback_later()
    print(f'iteration {it}: {1000 * (time.perf_counter() - t0) / K:.2f} ms per frame, {sum(len(r[1]["name"]) for r in out)} labels, '
          f'{torch.cuda.memory_allocated() / 2**30:.2f} GB allocated / {torch.cuda.memory_reserved() / 2**30:.2f} GB reserved', flush=True)
print('done')
