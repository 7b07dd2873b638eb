"""Real-data adapters -- SURVEY §8f row N3 (loader half): Waymo and Argoverse 2 sequences read from OpenPCDet's preprocessed
files, WITHOUT importing pcdet.

Mirrors (same member names, argument meaning and return layout)
  WaymoDataset    src/datasets/waymo_dataset.py:12-200   (+ `evaluation` :202-329, see vilgod_amd/evaluation.py)
  Argo2Dataset    src/datasets/argo2_dataset.py:10-215   (+ `evaluation` :217-377)
The two upstream classes repeat the same sequence bookkeeping line by line on top of two different pcdet base classes; here it
lives once in `SequenceDataset` and the two adapters only say where the infos and the point files are.

What upstream inherits from OpenPCDet (un-vendored, README.md:60-66) is restated from its published layout -- **parity
unpinned** for that dependency, pinned for the reference's own logic (tests/golden/make_golden.py::make_dataset runs the
reference classes over stand-in base classes that read the same files):
  DatasetTemplate.__init__         root_path / class_names / training / point_cloud_range, mode = 'train' if training else 'test'
  WaymoDataset.include_waymo_data  ImageSets/<split>.txt -> <DATA_PATH>/<TAG>/<seq>/<seq>.pkl (list of frame infos), SAMPLED_INTERVAL
  WaymoDataset.get_lidar           <seq>/%04d.npy rows [x, y, z, intensity, elongation, NLZ_flag]; NLZ rows dropped unless
                                   DISABLE_NLZ_FLAG_ON_POINTS; tanh on the intensity column (POINTS_TANH_DIM overrides)
  Argo2Dataset.include_argo2_data  <DATA_PATH>/<INFO_PATH[mode][i]> (list of frame infos)
  Argo2Dataset.get_lidar           <DATA_PATH>/training|testing/velodyne/<sample_idx>.bin float32 [n, 4]
  common_utils.drop_info_with_name / keep_arrays_by_name, box_utils.boxes_to_corners_3d / boxes3d_kitti_fakelidar_to_lidar
"""
import copy
import os
import pickle
from pathlib import Path

import numpy as np
from scipy.spatial.transform import Rotation


def _get(cfg, k, d=None):
    if cfg is None:
        return d
    return cfg.get(k, d) if hasattr(cfg, 'get') else getattr(cfg, k, d)


# ---- the handful of OpenPCDet helpers the adapters call -----------------------------------------------------------------------
def drop_info_with_name(info, name):
    keep = [i for i, x in enumerate(info['name']) if x != name]
    return {k: v[keep] for k, v in info.items()}


def keep_arrays_by_name(gt_names, used_classes):
    return np.array([i for i, x in enumerate(gt_names) if x in used_classes], dtype=np.int64)


def boxes_to_corners_3d(boxes3d):
    """[n,7] (x,y,z,dx,dy,dz,heading) -> [n,8,3]; corner order of pcdet's template (bottom 0-3, top 4-7)."""
    boxes3d = np.asarray(boxes3d)
    template = np.array([[1, 1, -1], [1, -1, -1], [-1, -1, -1], [-1, 1, -1], [1, 1, 1], [1, -1, 1], [-1, -1, 1], [-1, 1, 1]]) / 2
    corners = boxes3d[:, None, 3:6] * template[None]
    c, s = np.cos(boxes3d[:, 6]), np.sin(boxes3d[:, 6])
    zeros, ones = np.zeros_like(c), np.ones_like(c)
    rot = np.stack([c, s, zeros, -s, c, zeros, zeros, zeros, ones], axis=1).reshape(-1, 3, 3)        # points @ rot
    return np.matmul(corners, rot) + boxes3d[:, None, 0:3]


def boxes3d_kitti_fakelidar_to_lidar(boxes3d_lidar):
    b = np.array(boxes3d_lidar, copy=True)
    w, l, h, r = b[:, 3:4], b[:, 4:5], b[:, 5:6], b[:, 6:7]
    b[:, 2] += h[:, 0] / 2
    return np.concatenate([b[:, 0:3], l, w, h, -(r + np.pi / 2)], axis=-1)


def apply_transform(pts, transformation, box=False):
    """src/utils/pointcloud_utils.py:21-46 (numpy, mode 'left'): rows are transformed in place of a deep copy; boxes also get
    the transform's yaw added to column 6."""
    if len(pts) == 0:
        return pts
    out = copy.deepcopy(pts)
    hom = np.hstack((out[:, :3], np.ones((len(out), 1))))
    out[..., :3] = np.einsum('ij,kj->ki', transformation, hom)[..., :3]
    if box:
        out[..., 6] += Rotation.from_matrix(np.asarray(transformation)[:3, :3]).as_euler('xyz')[-1]
    return out


class _Log:
    def info(self, *a, **k):
        pass


class SequenceDataset:
    """The sequence bookkeeping both upstream adapters share (waymo_dataset.py:13-200 == argo2_dataset.py:11-215)."""

    def __init__(self, dataset_cfg, class_names, training=True, root_path=None, logger=None, start_sequence=None,
                 end_sequence=None):
        self.dataset_cfg = dataset_cfg
        self.class_names = list(class_names)
        self.training = training
        self.logger = logger if logger is not None else _Log()
        self.root_path = Path(root_path) if root_path is not None else Path(_get(dataset_cfg, 'DATA_PATH'))
        self.point_cloud_range = np.array(_get(dataset_cfg, 'POINT_CLOUD_RANGE'), dtype=np.float32)
        self.split = _get(dataset_cfg, 'DATA_SPLIT')[self.mode]
        self.infos = []
        self._load_infos()
        self.start_sequence = None
        self.end_sequence = None
        self.index_mapping = []
        self._sequence_mapping = self.create_sequence_mapping(start_sequence, end_sequence)
        self._sequence_indices = None
        self._moving_track_ids = None

    @property
    def mode(self):
        return 'train' if self.training else 'test'

    # -- what the two adapters define ---------------------------------------------------------------------------------------
    def _load_infos(self):
        raise NotImplementedError

    def _sequence_name_of(self, info):
        raise NotImplementedError

    def _prepare_sequence(self):
        """hook between 'indices set' and 'moving tracks extracted' (Argoverse sorts by time stamp and adapts the annos)."""

    def get_lidar_points(self, index, transformation=None):
        raise NotImplementedError

    # -- shared ---------------------------------------------------------------------------------------------------------------
    @property
    def sequence_mapping(self):
        return self._sequence_mapping.copy()

    @property
    def sequence_names(self):
        return self.filter_sequence_names(list(self._sequence_mapping.keys()), self.start_sequence, self.end_sequence)

    @property
    def sequence_length(self):
        return len(self._sequence_indices) if self._sequence_indices is not None else 0

    @property
    def sequence_indices(self):
        return self._sequence_indices.copy()

    @property
    def sequence_infos(self):
        return [self.infos[idx] for idx in self._sequence_indices]

    def next_sequence(self):
        for name in self.sequence_names:
            m = self._sequence_mapping[name]
            self._sequence_indices = list(range(m['start'], m['start'] + m['length']))
            self._prepare_sequence()
            tracks, _ = self.extract_moving_tracks()
            self._moving_track_ids = [k for k, v in tracks.items() if v['moving']]
            self._after_tracks()
            yield name

    def _after_tracks(self):
        pass

    def create_sequence_mapping(self, start=0, end=999):
        mapping = {}
        for i, info in enumerate(self.infos):
            name = self._sequence_name_of(info)
            if name not in mapping:
                mapping[name] = self._new_mapping_entry(i, info)
            else:
                self._extend_mapping_entry(mapping[name], info)
        n = len(mapping)
        self.start_sequence = start if (start is not None) and (start < n) else 0
        self.end_sequence = end if (end is not None) and (end <= n) else n
        self.logger.info(f'Using [{self.end_sequence - self.start_sequence}/{n}] sequences from {self.start_sequence} to {self.end_sequence}.')
        return mapping

    def _new_mapping_entry(self, i, info):
        return {'start': i, 'length': 1}

    def _extend_mapping_entry(self, entry, info):
        entry['length'] += 1

    def set_split(self, split):
        self.split = split
        self.infos = []
        self._load_infos()
        self._sequence_mapping = self.create_sequence_mapping(self.start_sequence, self.end_sequence)

    @staticmethod
    def filter_sequence_names(sequence_names, sequence_start_idx=0, sequence_end_idx=0):
        if sequence_end_idx - sequence_start_idx > 0:
            return sequence_names[sequence_start_idx:sequence_end_idx]
        if 0 < sequence_start_idx < len(sequence_names):
            return sequence_names[sequence_start_idx:]
        return sequence_names

    def get_annos(self, index, transformation=None, filtered=True):
        """waymo_dataset.py:88-153.  NOTE (upstream behaviour, kept): the filtered call REPLACES the frame's stored annos by the
        copy without 'unknown' objects and without empty boxes, and stores the `moving` flags in them."""
        info = self.infos[self._sequence_indices[index]]
        out = {}
        if 'annos' not in info:
            return out
        annos = info['annos']
        if not filtered:
            return {'gt_names': annos['name'], 'gt_boxes': annos['gt_boxes_lidar'],
                    'num_points_in_gt': annos.get('num_points_in_gt', None), 'obj_ids': annos['obj_ids']}
        if self._moving_track_ids is not None:
            info['annos']['moving'] = np.array([oid in self._moving_track_ids for oid in annos['obj_ids']])
        annos = drop_info_with_name(annos, name='unknown')
        keep = annos['num_points_in_gt'] >= 1
        for k, v in annos.items():
            annos[k] = v[keep]
        info['annos'] = annos
        boxes = boxes3d_kitti_fakelidar_to_lidar(annos['gt_boxes_lidar']) if _get(self.dataset_cfg, 'INFO_WITH_FAKELIDAR', False) \
            else annos['gt_boxes_lidar']
        if self.training and _get(self.dataset_cfg, 'FILTER_EMPTY_BOXES_FOR_TRAIN', False) and len(annos['name']) > 0:
            mask = annos['num_points_in_gt'] > 0
            mask &= np.array([n in self.class_names for n in annos['name']], dtype=bool)
            annos['name'] = annos['name'][mask]
            boxes = boxes[mask]
            annos['num_points_in_gt'] = annos['num_points_in_gt'][mask]
            annos['obj_ids'] = annos['obj_ids'][mask]
        if len(boxes) > 0 and transformation is not None:
            boxes = apply_transform(boxes, transformation, box=True)
        out.update({'gt_names': annos['name'], 'gt_boxes': boxes, 'num_points_in_gt': annos.get('num_points_in_gt', None),
                    'obj_ids': annos['obj_ids']})
        if out.get('gt_boxes', None) is not None:
            sel = keep_arrays_by_name(out['gt_names'], self.class_names)
            for k, v in out.items():
                if isinstance(v, np.ndarray):
                    out[k] = v[sel]
        if self._moving_track_ids is not None:
            out['moving'] = np.array([oid in self._moving_track_ids for oid in out['obj_ids']])
        return out

    def extract_moving_tracks(self, threshold=1.0):
        """waymo_dataset.py:166-200: an object moves when, expressed in the pose of its first frame, its box centre gets
        farther than `threshold` from where it started."""
        tracks = {}
        for f in range(self.sequence_length):
            a = self.get_annos(f, transformation=None, filtered=False)
            for t, tid in enumerate(a['obj_ids']):
                tr = tracks.setdefault(tid, {'indices': [], 'gt_boxes': [], 'gt_boxes_ref': [], 'gt_names': [], 'num_points_in_gt': []})
                tr['indices'].append(f)
                tr['gt_boxes'].append(a['gt_boxes'][t].copy())
                tr['gt_names'].append(a['gt_names'][t])
                tr['num_points_in_gt'].append(a['num_points_in_gt'][t])
        infos = self.sequence_infos
        n_moving = 0
        for tr in tracks.values():
            tr['moving'] = False
            if len(tr['indices']) > 1:
                ref_pose = infos[tr['indices'][0]]['pose']
                ref_box = tr['gt_boxes'][0].copy()
                tr['gt_boxes_ref'].append(ref_box)
                for i in range(len(tr['indices']) - 1):
                    pose = infos[tr['indices'][i + 1]]['pose']
                    box = tr['gt_boxes'][i + 1].copy()
                    box[:7] = apply_transform(np.array([box[:7]]), np.linalg.inv(ref_pose) @ pose, box=True)
                    tr['gt_boxes_ref'].append(box)
                    if np.linalg.norm(ref_box[:3] - box[:3]) > threshold:
                        tr['moving'] = True
                        tr['gt_boxes_ref'] = np.array(tr['gt_boxes_ref'])
                        n_moving += len(tr['gt_boxes'])
                        break
        return tracks, n_moving

    def evaluation(self, det_annos, class_names, **kwargs):
        from . import evaluation as ev
        if 'annos' not in self.infos[0].keys():
            return 'No ground-truth boxes for evaluation', {}
        return ev.evaluate_detections(self, det_annos, class_names, **kwargs)


class WaymoDataset(SequenceDataset):
    """src/datasets/waymo_dataset.py over OpenPCDet's `waymo_processed_data_v0_5_0` layout."""

    def _load_infos(self):
        self.data_path = self.root_path / _get(self.dataset_cfg, 'PROCESSED_DATA_TAG')
        split_file = self.root_path / 'ImageSets' / (self.split + '.txt')
        self.sample_sequence_list = [x.strip() for x in open(split_file).readlines()]
        infos, skipped = [], 0
        self.seq_name_to_infos = {}
        for entry in self.sample_sequence_list:
            name = os.path.splitext(entry)[0]
            path = self._with_all_versions(self.data_path / name / f'{name}.pkl')
            if not path.exists():
                skipped += 1
                continue
            with open(path, 'rb') as f:
                seq = pickle.load(f)
            infos.extend(seq)
            self.seq_name_to_infos[seq[0]['point_cloud']['lidar_sequence']] = seq
        self.infos.extend(infos)
        self.logger.info(f'Total skipped info {skipped}')
        self.logger.info(f'Total samples for Waymo dataset: {len(infos)}')
        interval = _get(self.dataset_cfg, 'SAMPLED_INTERVAL', {'train': 1, 'test': 1})[self.mode]
        if interval > 1:
            self.infos = self.infos[::interval]
            self.logger.info(f'Total sampled samples for Waymo dataset: {len(self.infos)}')

    @staticmethod
    def _with_all_versions(path):
        """pcdet `check_sequence_name_with_all_version`: segments exist with and without the `_with_camera_labels` suffix."""
        if path.exists():
            return path
        stem, tag = path.stem, '_with_camera_labels'
        other = stem[:-len(tag)] if stem.endswith(tag) else stem + tag
        alt = path.parent.parent / other / f'{other}.pkl'
        return alt if alt.exists() else path

    def _sequence_name_of(self, info):
        return '_'.join(info['frame_id'].split('_')[:-1])

    def _after_tracks(self):
        for f in range(self.sequence_length):            # waymo_dataset.py:53-54: filters every frame's annos once
            self.get_annos(f, transformation=None, filtered=True)

    def get_lidar(self, sequence_name, sample_idx):
        feats = np.load(self.data_path / sequence_name / ('%04d.npy' % sample_idx))
        points, nlz = feats[:, 0:5], feats[:, 5]
        if not _get(self.dataset_cfg, 'DISABLE_NLZ_FLAG_ON_POINTS', False):
            points = points[nlz == -1]
        dims = _get(self.dataset_cfg, 'POINTS_TANH_DIM', None)
        for d in ([3] if dims is None else dims):
            points[:, d] = np.tanh(points[:, d])
        return points

    def get_lidar_points(self, index, transformation=None):
        pc = self.infos[self._sequence_indices[index]]['point_cloud']
        pts = self.get_lidar(pc['lidar_sequence'], pc['sample_idx'])
        return apply_transform(pts, transformation) if transformation is not None else pts


class Argo2Dataset(SequenceDataset):
    """src/datasets/argo2_dataset.py over OpenPCDet's `argo2_infos_<split>.pkl` + KITTI-style velodyne files."""

    def _load_infos(self):
        mode = [k for k, v in dict(_get(self.dataset_cfg, 'DATA_SPLIT')).items() if v == self.split][0]
        self.root_split_path = self.root_path / ('training' if self.split != 'test' else 'testing')
        self.argo2_infos = []
        for rel in _get(self.dataset_cfg, 'INFO_PATH')[mode]:
            path = self.root_path / rel
            if not path.exists():
                continue
            with open(path, 'rb') as f:
                self.argo2_infos.extend(pickle.load(f))
        self.infos = self.argo2_infos
        self.logger.info(f'Total samples for Argo2 dataset: {len(self.infos)}')

    def _sequence_name_of(self, info):
        return info['uuid'].split('/')[0]

    def _new_mapping_entry(self, i, info):
        return {'start': i, 'length': 1, 'indices': [int(info['sample_idx'][4:7])]}

    def _extend_mapping_entry(self, entry, info):
        entry['length'] += 1
        entry['indices'].append(int(info['sample_idx'][4:7]))

    def _prepare_sequence(self):
        order = np.argsort([int(self.infos[i]['uuid'].split('/')[1]) for i in self._sequence_indices])     # time stamps
        self._sequence_indices = [self._sequence_indices[i] for i in order]
        self.adapt_annos()

    def adapt_annos(self):
        """argo2_dataset.py:95-107: KITTI-style location/dimensions/rotation_y -> gt_boxes_lidar; Argoverse categories -> the
        three evaluation classes through CLASS_MAPPING, everything else 'unknown'."""
        mapping = dict(_get(self.dataset_cfg, 'CLASS_MAPPING', {}))
        for idx in self._sequence_indices:
            annos = self.infos[idx]['annos']
            annos['gt_boxes_lidar'] = np.concatenate([annos['location'], annos['dimensions'], annos['rotation_y'][..., np.newaxis]],
                                                     axis=1).astype(np.float32)
            for i, name in enumerate(annos['name']):
                if name in mapping:
                    annos['name'][i] = mapping[name]
                elif name not in self.class_names:
                    annos['name'][i] = 'unknown'

    def get_lidar(self, idx):
        return np.fromfile(str(self.root_split_path / 'velodyne' / ('%s.bin' % idx)), dtype=np.float32).reshape(-1, 4)

    def get_lidar_points(self, index, transformation=None):
        pts = self.get_lidar(self.infos[self._sequence_indices[index]]['sample_idx'])
        return apply_transform(pts, transformation) if transformation is not None else pts
