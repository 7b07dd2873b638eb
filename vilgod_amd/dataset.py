"""Synthetic dataset adapter with the duck-type the reference pipeline uses (SURVEY §8b "Dataset duck-type"):
`class_names`, `sequence_length`, `get_annos(fnr)`, `get_lidar_points(fnr)`, `sequence_infos[fnr]['pose']`,
`next_sequence()`, `sequence_indices`, `set_split`, `training`, `evaluation(...)`
(src/datasets/waymo_dataset.py:46,88,155,202; used at src/vilgod/zero_shot_detector.py:78-96 and
tools/preprocess_data.py:35-40,73,96,120).  Frames come from vilgod_amd.synthetic (seeded)."""
import numpy as np

from . import synthetic


def _get(cfg, k, d=None):
    return cfg.get(k, d) if hasattr(cfg, 'get') else getattr(cfg, k, d)


class SyntheticDataset:
    def __init__(self, dataset_cfg=None, class_names=('Vehicle', 'Pedestrian', 'Cyclist'), logger=None, training=True,
                 start_sequence=0, end_sequence=2, **kw):
        syn = _get(dataset_cfg, 'SYNTHETIC', {}) if dataset_cfg is not None else {}
        self.class_names = list(class_names)
        self.logger = logger
        self.training = training
        self.split = 'train'
        self.n_sequences = int(_get(syn, 'n_sequences', 4))
        self.frames_per_sequence = int(_get(syn, 'frames_per_sequence', 199))
        self.points_per_frame = int(_get(syn, 'points_per_frame', 150_000))
        self.objects_per_frame = int(_get(syn, 'objects_per_frame', 60))
        self.step = float(_get(syn, 'step', 0.5))
        self.seed = int(_get(syn, 'seed', 0))
        # coherent: one world per sequence (static scene + moving objects + ego motion) -- what the entropy scores and
        # the two-frame clustering need; False: independent frames (`synthetic.make_frame`)
        self.coherent = bool(_get(syn, 'coherent', True))
        self._frames = None
        self.start_sequence, self.end_sequence = int(start_sequence), int(end_sequence)
        self.sequence_name = None
        self.sequence_infos = []
        self.sequence_indices = []
        self._seq_id = -1

    def set_split(self, split):
        self.split = split

    @property
    def sequence_length(self):
        return len(self.sequence_infos)

    def next_sequence(self):
        """Generator of sequence names; loads poses/infos of each sequence (waymo_dataset.py `next_sequence`)."""
        for sid in range(self.start_sequence, min(self.end_sequence + 1, self.n_sequences)):
            self._seq_id = sid
            self.sequence_name = f'synthetic_{self.split}_{sid:04d}'
            poses = synthetic.make_poses(self.frames_per_sequence, step=self.step, seed=self.seed + sid)
            self.sequence_infos = [{'pose': p, 'frame_id': f'{self.sequence_name}_{i:03d}'} for i, p in enumerate(poses)]
            base = sid * self.frames_per_sequence
            self.sequence_indices = list(range(base, base + self.frames_per_sequence))
            self._frames = None
            yield self.sequence_name

    def get_lidar_points(self, fnr):
        """(N,5) float32 [x,y,z,intensity,elongation] in the vehicle frame."""
        if not self.coherent:
            return synthetic.make_frame(self.seed + self._seq_id * 100_003 + fnr, self.points_per_frame,
                                        n_objects=self.objects_per_frame)
        if self._frames is None:
            self._frames, _ = synthetic.make_sequence(self.seed + self._seq_id, self.frames_per_sequence, self.points_per_frame,
                                                      n_objects=self.objects_per_frame, step=self.step)
        return self._frames[fnr]

    def get_annos(self, fnr):
        return {'gt_names': np.array([]), 'moving': np.array([], dtype=bool), 'gt_boxes_lidar': np.zeros((0, 7))}

    def evaluation(self, det_annos, class_names, **kwargs):
        """No ground truth for synthetic frames: report label statistics instead of AP (the TF Waymo metrics of
        src/datasets/waymo_eval.py are out of scope)."""
        names = np.concatenate([d['name'] for d in det_annos]) if det_annos else np.array([])
        out = {'n_frames': len(det_annos), 'n_labels': int(len(names))}
        for c in class_names:
            out[f'n_{c}'] = int((names == c).sum())
        return out
