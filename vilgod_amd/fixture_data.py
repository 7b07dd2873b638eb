"""Writes small seeded data sets in the ON-DISK LAYOUT OpenPCDet's preprocessing produces for Waymo and Argoverse 2, from the
synthetic sequence generator (no dataset can be downloaded here).  Used by the tests and tools/make_fixture_dataset.py to run the
real-data adapters (vilgod_amd/sequence_datasets.py) and the evaluation end to end.

Waymo     <root>/ImageSets/<split>.txt
          <root>/<TAG>/<seq>/<seq>.pkl          list of frame infos: point_cloud{lidar_sequence, sample_idx, num_features},
                                                frame_id '<seq>_%03d', pose [4,4], annos{name, difficulty, dimensions, location,
                                                heading_angles, obj_ids, tracking_difficulty, num_points_in_gt, gt_boxes_lidar}
          <root>/<TAG>/<seq>/%04d.npy           float32 [n,6] x, y, z, intensity, elongation, NLZ_flag
Argoverse <root>/argo2_infos_<split>.pkl        list of frame infos: uuid '<log>/<time stamp>', sample_idx '<s><qqq><fff>', pose,
                                                annos{name, location, dimensions, rotation_y, obj_ids, num_points_in_gt, difficulty}
          <root>/training/velodyne/<sample_idx>.bin   float32 [n,4]
"""
import os
import pickle

import numpy as np

from . import synthetic

WAYMO_NAME = {'car': 'Vehicle', 'truck': 'Vehicle', 'pedestrian': 'Pedestrian', 'cyclist': 'Cyclist', 'pole': 'Sign'}
ARGO_NAME = {'car': 'Regular_vehicle', 'truck': 'Truck', 'pedestrian': 'Pedestrian', 'cyclist': 'Bicyclist', 'pole': 'Bollard',
             'wall': 'Construction_barrel'}


def _annos(truth, naming, rng, empty_every=7):
    n = len(truth['kind'])
    name = np.array([naming.get(k, 'unknown') for k in truth['kind']])
    npts = truth['n_points'].astype(np.int64).copy()
    npts[(truth['id'] % empty_every) == empty_every - 1] = 0            # a few boxes without points, as in real annotations
    return name, npts, np.array([f'obj_{i:04d}' for i in truth['id']]), truth['box'].astype(np.float32), n


def write_waymo(root, n_sequences=2, n_frames=6, n_points=4000, n_objects=10, seed=0, split='train', tag='waymo_processed_data_v0_5_0',
                nlz_fraction=0.05):
    os.makedirs(os.path.join(root, 'ImageSets'), exist_ok=True)
    names = []
    for s in range(n_sequences):
        seq = f'segment-{1000 + seed + s}_{17 * (s + 1)}_000_{37 * (s + 1)}_000_with_camera_labels'
        names.append(seq)
        d = os.path.join(root, tag, seq)
        os.makedirs(d, exist_ok=True)
        frames, poses, truth = synthetic.make_sequence(seed + s, n_frames, n_points, n_objects=n_objects, return_objects=True)
        rng = np.random.default_rng(seed + 31 * s)
        infos = []
        for f in range(n_frames):
            name, npts, ids, box, n = _annos(truth[f], WAYMO_NAME, rng)
            pts = frames[f]
            nlz = np.where(rng.uniform(size=len(pts)) < nlz_fraction, 1.0, -1.0).astype(np.float32)
            raw = np.c_[pts[:, :3], np.arctanh(np.clip(pts[:, 3:4], 0, 0.999)), pts[:, 4:5], nlz].astype(np.float32)
            np.save(os.path.join(d, '%04d.npy' % f), raw)
            infos.append({'point_cloud': {'num_features': 5, 'lidar_sequence': seq, 'sample_idx': f}, 'frame_id': f'{seq}_{f:03d}',
                          'pose': np.asarray(poses[f], np.float64),
                          'annos': {'name': name, 'difficulty': np.zeros(n, np.int32), 'dimensions': box[:, 3:6], 'location': box[:, :3],
                                    'heading_angles': box[:, 6], 'obj_ids': ids, 'tracking_difficulty': np.zeros(n, np.int32),
                                    'num_points_in_gt': npts, 'gt_boxes_lidar': box}})
        with open(os.path.join(d, f'{seq}.pkl'), 'wb') as fp:
            pickle.dump(infos, fp)
    with open(os.path.join(root, 'ImageSets', f'{split}.txt'), 'w') as fp:
        fp.write('\n'.join(n + '.tfrecord' for n in names) + '\n')
    return names


def write_argo2(root, n_sequences=2, n_frames=6, n_points=4000, n_objects=10, seed=0, split='train'):
    os.makedirs(os.path.join(root, 'training', 'velodyne'), exist_ok=True)
    infos, logs = [], []
    for s in range(n_sequences):
        log = f'log{seed + s:05d}-synthetic'
        logs.append(log)
        frames, poses, truth = synthetic.make_sequence(seed + s, n_frames, n_points, n_objects=n_objects, return_objects=True)
        rng = np.random.default_rng(seed + 31 * s)
        order = rng.permutation(n_frames)                              # infos are not stored in time order (the adapter sorts)
        for f in order:
            name, npts, ids, box, n = _annos(truth[f], ARGO_NAME, rng)
            sample_idx = f'0{s:03d}{f:03d}'
            frames[f][:, :4].astype(np.float32).tofile(os.path.join(root, 'training', 'velodyne', f'{sample_idx}.bin'))
            infos.append({'uuid': f'{log}/{315969904359876000 + 100000000 * int(f)}', 'sample_idx': sample_idx,
                          'pose': np.asarray(poses[f], np.float64),
                          'annos': {'name': name.astype('<U24'), 'location': box[:, :3], 'dimensions': box[:, 3:6], 'rotation_y': box[:, 6],
                                    'obj_ids': ids, 'num_points_in_gt': npts, 'difficulty': np.zeros(n, np.int32)}})
    with open(os.path.join(root, f'argo2_infos_{split}.pkl'), 'wb') as fp:
        pickle.dump(infos, fp)
    return logs


WAYMO_CFG = {'DATASET': 'WaymoDataset', 'PROCESSED_DATA_TAG': 'waymo_processed_data_v0_5_0',
             'POINT_CLOUD_RANGE': [-75.2, -75.2, -2, 75.2, 75.2, 4], 'DATA_SPLIT': {'train': 'train', 'test': 'val'},
             'SAMPLED_INTERVAL': {'train': 1, 'test': 1}, 'FILTER_EMPTY_BOXES_FOR_TRAIN': True, 'DISABLE_NLZ_FLAG_ON_POINTS': True}
ARGO2_CFG = {'DATASET': 'Argo2Dataset', 'POINT_CLOUD_RANGE': [-100, -100, -20, 100, 100, 20],
             'DATA_SPLIT': {'train': 'train', 'test': 'val'}, 'INFO_PATH': {'train': ['argo2_infos_train.pkl'], 'test': ['argo2_infos_val.pkl']},
             'CLASS_MAPPING': {'Regular_vehicle': 'Vehicle', 'Pedestrian': 'Pedestrian', 'Bicyclist': 'Cyclist', 'Motorcyclist': 'Cyclist',
                               'Wheeled_rider': 'Cyclist', 'Large_vehicle': 'Vehicle', 'Bus': 'Vehicle', 'Box_truck': 'Vehicle',
                               'Truck': 'Vehicle', 'Vehicular_trailer': 'Vehicle', 'Truck_cab': 'Vehicle', 'School_bus': 'Vehicle',
                               'Articulated_bus': 'Vehicle', 'Message_board_trailer': 'Vehicle'}}
