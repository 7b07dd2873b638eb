"""Pseudo-label exporter (exporter half): turns the per-frame result dicts the pipeline returns
(`{'boxes_lidar' (n,7) ego frame, 'name', 'score', 'moving'}`, zero_shot_detector.py:852-857) into what an OpenPCDet-style
training step reads as ground truth.

The reference declares `paths.pseudo_label` (tools/configs/preprocessing.yaml:13) but never writes to it; its result pickles
are consumed by a modified dataset class upstream.  This writer is therefore an ADDITION, not parity:

  <paths.pseudo_label>/<sequence>.pkl   list of OpenPCDet `infos`-style dicts, one per frame:
        {'frame_id': str, 'sample_idx': int, 'annos': {'name': ndarray[str], 'gt_boxes_lidar': ndarray (n,7) float32
         [x,y,z,dx,dy,dz,heading] in the vehicle frame, 'score': ndarray float32, 'moving': ndarray bool,
         'num_points_in_gt': ndarray int32 (-1: unknown)}}
  <paths.pseudo_label>/<sequence>.npz   the same, flat: `frame_ptr` int64 [F+1] offsets into `gt_boxes_lidar` [N,7] float32,
        `name` [N] str, `score` [N] float32, `moving` [N] bool, `frame_id` [F] str, `sample_idx` [F] int64
"""
import pickle
from pathlib import Path

import numpy as np


def frame_info(result, frame_id, sample_idx, score_thresh=0.0, class_names=None):
    keep = np.asarray(result['score'], dtype=np.float32) >= score_thresh if len(result['score']) else np.zeros(0, bool)
    if class_names is not None and len(keep):
        keep &= np.isin(np.asarray(result['name']), list(class_names))
    boxes = np.asarray(result['boxes_lidar'], dtype=np.float32).reshape(-1, 7)[keep]
    return {'frame_id': str(frame_id), 'sample_idx': int(sample_idx),
            'annos': {'name': np.asarray(result['name'])[keep].astype(str),
                      'gt_boxes_lidar': boxes,
                      'score': np.asarray(result['score'], dtype=np.float32)[keep],
                      'moving': np.asarray(result['moving'], dtype=bool)[keep] if len(result.get('moving', [])) else np.zeros(int(keep.sum()), bool),
                      'num_points_in_gt': np.full(int(keep.sum()), -1, np.int32)}}


def write_sequence(out_dir, sequence_name, results, frame_ids, sample_indices, score_thresh=0.0, class_names=None, npz=True):
    """results: list of per-frame result dicts; frame_ids / sample_indices: per frame.  -> (pkl path, npz path or None)."""
    out_dir = Path(out_dir)
    out_dir.mkdir(parents=True, exist_ok=True)
    infos = [frame_info(r, f, i, score_thresh, class_names) for r, f, i in zip(results, frame_ids, sample_indices)]
    pkl = out_dir / f'{sequence_name}.pkl'
    with open(pkl, 'wb') as fp:
        pickle.dump(infos, fp)
    npz_path = None
    if npz:
        counts = [len(i['annos']['name']) for i in infos]
        npz_path = out_dir / f'{sequence_name}.npz'
        cat = lambda k, dt, shape: (np.concatenate([i['annos'][k] for i in infos]) if sum(counts) else np.zeros(shape, dt))
        np.savez_compressed(npz_path, frame_ptr=np.r_[0, np.cumsum(counts)].astype(np.int64),
                            gt_boxes_lidar=cat('gt_boxes_lidar', np.float32, (0, 7)).astype(np.float32),
                            name=cat('name', '<U1', (0,)).astype(str), score=cat('score', np.float32, (0,)),
                            moving=cat('moving', bool, (0,)),
                            frame_id=np.array([i['frame_id'] for i in infos]), sample_idx=np.array([i['sample_idx'] for i in infos], np.int64))
    return pkl, npz_path


def read_npz(path):
    """-> list of per-frame annos dicts (inverse of the NPZ layout)."""
    z = np.load(path, allow_pickle=False)
    out = []
    for f in range(len(z['frame_id'])):
        a, b = int(z['frame_ptr'][f]), int(z['frame_ptr'][f + 1])
        out.append({'frame_id': str(z['frame_id'][f]), 'sample_idx': int(z['sample_idx'][f]),
                    'annos': {'name': z['name'][a:b], 'gt_boxes_lidar': z['gt_boxes_lidar'][a:b], 'score': z['score'][a:b],
                              'moving': z['moving'][a:b]}})
    return out
