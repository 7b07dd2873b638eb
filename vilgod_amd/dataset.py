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
        # seed of sequence sid = seed + sid * seed_stride; 0 = every sequence is the SAME world under its own name (bench.py's cli_mode:
        # several sequences of the benchmark workload; generated once)
        self.seed_stride = int(_get(syn, 'seed_stride', 1))
        self._gen_cache = {}
        # coherent: one world per sequence (static scene + moving objects + ego motion) -- what the entropy scores and
        # the two-frame clustering need; False: independent frames (`synthetic.make_frame`)
        self.coherent = bool(_get(syn, 'coherent', True))
        self._frames = None
        self.dataset_cfg = dataset_cfg
        self.point_cloud_range = np.array(_get(dataset_cfg, 'POINT_CLOUD_RANGE', [-75.2, -75.2, -2, 75.2, 75.2, 4]), dtype=np.float32)
        self.infos = {}                      # global frame index -> info (filled as sequences are generated; `evaluation` reads it)
        self.index_mapping = []
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

    @property
    def sequence_names(self):
        """Names `next_sequence` will yield (the real-data adapters' member of the same name, sequence_datasets.py)."""
        return [f'synthetic_{self.split}_{sid:04d}' for sid in range(self.start_sequence, min(self.end_sequence + 1, self.n_sequences))]

    def next_sequence(self):
        """Generator of sequence names; loads poses/infos of each sequence (waymo_dataset.py `next_sequence`)."""
        for sid in range(self.start_sequence, min(self.end_sequence + 1, self.n_sequences)):
            self._seq_id = sid
            self.sequence_name = f'synthetic_{self.split}_{sid:04d}'
            poses = synthetic.make_poses(self.frames_per_sequence, step=self.step, seed=self.seed + sid * self.seed_stride)
            self.sequence_infos = [{'pose': p, 'frame_id': f'{self.sequence_name}_{i:03d}'} for i, p in enumerate(poses)]
            base = sid * self.frames_per_sequence
            self.sequence_indices = list(range(base, base + self.frames_per_sequence))
            self._frames = None
            yield self.sequence_name

    def get_lidar_points(self, fnr):
        """(N,5) float32 [x,y,z,intensity,elongation] in the vehicle frame."""
        if not self.coherent:
            return synthetic.make_frame(self.seed + self._seq_id * self.seed_stride * 100_003 + fnr, self.points_per_frame,
                                        n_objects=self.objects_per_frame)
        self._generate()
        return self._frames[fnr]

    def prefetch_sequence(self):
        """Generate the current sequence now (tools/preprocess_data.py calls it before it starts the sequence clock)."""
        if self.coherent:
            self._generate()

    def _generate(self):
        """The coherent sequence and its ground truth (the generator's boxes in the OpenPCDet `annos` layout), once per sequence."""
        if self._frames is not None:
            return
        from .fixture_data import WAYMO_NAME
        gseed = self.seed + self._seq_id * self.seed_stride
        if gseed not in self._gen_cache:
            if self.seed_stride != 0:
                self._gen_cache.clear()                  # (one sequence's clouds at a time: 0.6 GB per 199 x 150k-point sequence)
            self._gen_cache[gseed] = synthetic.make_sequence(gseed, self.frames_per_sequence, self.points_per_frame,
                                                             n_objects=self.objects_per_frame, step=self.step, return_objects=True)
        self._frames, _, truth = self._gen_cache[gseed]
        for info, idx, t in zip(self.sequence_infos, self.sequence_indices, truth):
            n = len(t['kind'])
            info['annos'] = {'name': np.array([WAYMO_NAME.get(k, 'unknown') for k in t['kind']]), 'gt_boxes_lidar': t['box'].astype(np.float32),
                             'num_points_in_gt': t['n_points'].astype(np.int64), 'obj_ids': np.array([f'obj_{i:04d}' for i in t['id']]),
                             'difficulty': np.zeros(n, np.int32), 'moving': t['moving'].copy()}
            self.infos[idx] = info

    def get_annos(self, fnr, transformation=None, filtered=True):
        """Ground truth of the classes of interest (waymo_dataset.py:88-153 without the real-data filters)."""
        if not self.coherent:
            return {'gt_names': np.array([]), 'moving': np.array([], dtype=bool), 'gt_boxes': np.zeros((0, 7), np.float32)}
        self._generate()
        a = self.sequence_infos[fnr]['annos']
        keep = np.isin(a['name'], self.class_names) if filtered else np.ones(len(a['name']), bool)
        return {'gt_names': a['name'][keep], 'gt_boxes': a['gt_boxes_lidar'][keep], 'num_points_in_gt': a['num_points_in_gt'][keep],
                'obj_ids': a['obj_ids'][keep], 'moving': a['moving'][keep]}

    def evaluation(self, det_annos, class_names, **kwargs):
        """Coherent sequences carry the generator's ground truth: the same AP / APH evaluation as the real-data adapters
        (vilgod_amd/evaluation.py).  Independent frames have none: label statistics only."""
        names = np.concatenate([d['name'] for d in det_annos]) if det_annos else np.array([])
        out = {'n_frames': len(det_annos), 'n_labels': int(len(names))}
        for c in class_names:
            out[f'n_{c}'] = int((names == c).sum())
        indices = list(kwargs.get('indices', []))
        if self.coherent and len(indices) == len(det_annos) and all(i in self.infos for i in indices):
            from . import evaluation as ev
            out.update(ev.evaluate_detections(self, det_annos, class_names, **dict(kwargs, style='waymo')))
        return out
