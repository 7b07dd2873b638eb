import logging, os, sys, tempfile
sys.path.insert(0, '/root/repo/tools'); sys.path.insert(0, '/root/repo')
import preprocess_data
import numpy as np
frames, points, nseq = int(sys.argv[1]), 150000, 2
from vilgod_amd import entropy as E
orig = E.TwoFrameClusterer.cluster_input
def ci(self, fnr, X_list, ent_list):
    out = orig(self, fnr, X_list, ent_list)
    if out.shape[0] > 140000:
        rng = self.used_frames(fnr, len(X_list))
        print('BIG', fnr, out.shape, [X_list[f].shape for f in rng], [float((ent_list[f] < 0.6).float().mean()) for f in rng], flush=True)
    return out
E.TwoFrameClusterer.cluster_input = ci
with tempfile.TemporaryDirectory() as root:
    try:
        preprocess_data.main(['preprocessor=waymo', f'dataset.DATA_PATH={root}', f'dataset.SYNTHETIC.frames_per_sequence={frames}',
                          f'dataset.SYNTHETIC.points_per_frame={points}', f'dataset.SYNTHETIC.n_sequences={nseq}', f'end_sequence={nseq - 1}',
                          f'device.max_points={points + 1024}', 'paths.clip_model=/nonexistent'] + sys.argv[2:])
    except Exception as e:
        print('ERR', e)
for q in preprocess_data.LAST_RUN['sequences']:
    print(q['name'], q['frames'], q['seconds'])
