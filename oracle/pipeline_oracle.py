"""CPU oracle of the WHOLE per-frame hot path (SURVEY §3.2 [A]-[F]) chained from the per-stage oracles.

TEST INFRASTRUCTURE ONLY: end-to-end parity tests, smoke(), and bench.py's cpu_baseline leg (kind "port":
the reference itself cannot travel to the GPU box).  Single frame, fresh Patchwork++ state unless a
`ground` object is passed in.
"""
import time

import numpy as np
import torch

from . import hdbscan_oracle as ho
from . import neighbors_oracle as no
from . import patchworkpp as opw
from . import render_oracle as ro
from . import segment_oracle as so
from . import vit_oracle as vo


class OraclePipeline:
    def __init__(self, weights, text_features, class_list, class_mapping, class_names=('Vehicle', 'Pedestrian', 'Cyclist'),
                 heads=12, view_angles=ro.VIEW_ANGLES, min_range=1.5, z_offset=1.723, plane_seed=666,
                 clusterer='oracle', box_all_edges=True):
        self.wd, self.text, self.heads = weights, text_features, heads
        self.class_list = list(class_list)
        self.mapping = dict(class_mapping)
        self.class_names = list(class_names)
        self.rot = ro.view_matrices(view_angles)
        self.z_offset, self.plane_seed = z_offset, plane_seed
        p = opw.Parameters()
        p.min_range = min_range
        self._params = p
        self.ground = opw.patchworkpp(p)
        self.clusterer = clusterer
        self.box_all_edges = box_all_edges
        self.timings = {}

    def new_sequence(self):
        self.ground = opw.patchworkpp(self._params)

    def process_frame(self, points, pose, ref_pose):
        self._t, self._t0 = {}, time.perf_counter()
        prep = self._prepare(points, pose, ref_pose)
        X = prep['X']
        labels, probs = (ho.fit if self.clusterer == 'oracle' else ho.sklearn_fit)(X[:, :3])
        self._tick('cluster')
        return self._label(prep, labels, probs)

    def process_sequence(self, frames, poses, ref_pose, n_neighbouring_frames=15, skip_frames=1, n_frames=2, seed=0):
        """SURVEY 8f N1 (the reference's default stage order, preprocessing.yaml:50-68): ground -> entropy scores over
        the sliding window -> two-frame clustering + nearest-label transfer -> the per-frame rest."""
        self.new_sequence()
        self._t, self._t0 = {}, time.perf_counter()
        preps = [self._prepare(f, p, ref_pose) for f, p in zip(frames, poses)]
        X_list = [p['X'] for p in preps]
        kept = no.entropy_scores_sequence(X_list, n_neighbouring_frames, skip_frames)
        ent = [no.full_scores(len(X), s, i) for X, (s, i) in zip(X_list, kept)]
        out = []
        for f, prep in enumerate(preps):
            if n_frames > 1 and len(frames) >= n_frames:
                seq, _, _ = no.two_frame_input(X_list, ent, f, n_frames, seed)
                lab_seq, prob_seq = ho.fit(seq) if len(seq) >= 2 else (np.zeros(len(seq), np.int64) - 1, np.zeros(len(seq)))
                labels, probs = no.knn_labels(X_list[f], seq, lab_seq, prob_seq)
            else:
                labels, probs = ho.fit(X_list[f][:, :3])
            r = self._label(prep, labels, probs, entropy=ent[f])
            r['entropy_scores'], r['entropy_indices'] = kept[f]
            out.append(r)
        return out

    def _tick(self, k):
        self._t[k] = self._t.get(k, 0.0) + time.perf_counter() - self._t0
        self._t0 = time.perf_counter()

    def _prepare(self, points, pose, ref_pose):
        T_ref = np.linalg.inv(ref_pose) @ pose
        T_ego = np.linalg.inv(pose) @ ref_pose
        gidx = opw.mask_ground_points(points, self.ground, self.z_offset)
        self._tick('ground')
        pref = so.apply_transform(points, T_ref)
        gm = np.zeros(len(points), bool)
        gm[gidx] = True
        X = pref[~gm]
        self._tick('to_ref')
        return dict(T_ego=T_ego, gidx=gidx, pref=pref, gm=gm, X=X)

    def _label(self, prep, labels, probs, entropy=None):
        T_ego, gidx, pref, gm, X = prep['T_ego'], prep['gidx'], prep['pref'], prep['gm'], prep['X']
        t = self._t
        tick = self._tick
        dets = so.generate_detections(labels, probs)
        static = np.array([not no.filter_by_ephemeral_score(entropy[idx]) for _, idx in dets], bool) if entropy is not None \
            else np.ones(len(dets), bool)
        plane = so.fit_plane(pref[gm], self.plane_seed) if gm.sum() >= 3 else np.array([0., 0., 1., 0.])
        valid = [so.filter_cluster(X[idx], plane)[0] for _, idx in dets]
        tick('filter')
        vdets = [d for d, v in zip(dets, valid) if v]
        clusters_ego = [so.apply_transform(X[idx][:, :3], T_ego) for _, idx in vdets]
        u8, norm = ro.render_clusters(clusters_ego, self.rot)
        tick('render')
        with torch.no_grad():
            feats = vo.encode_in_chunks(self.wd, norm, self.heads, 50) if len(norm) else torch.zeros(0, self.text.shape[1])
            pr = vo.clip_probabilities(feats, self.text) if len(norm) else torch.zeros(0, len(self.class_list))
        top, score = vo.top1(pr)
        tick('vit')
        V = self.rot.shape[0]
        names, scores, boxes = [], [], []
        for k, (_, idx) in enumerate(vdets):
            fine = [self.class_list[i] for i in top[k * V:(k + 1) * V]]
            mapped = np.array([self.mapping[f] for f in fine])
            n, s = vo.vote(mapped, np.asarray(score[k * V:(k + 1) * V], dtype=np.float32))
            names.append(n)
            scores.append(s)
            boxes.append(so.fit_box(X[idx], self.box_all_edges))
        keep = [n in self.class_names for n in names]
        bx = np.array([b for b, k in zip(boxes, keep) if k]).reshape(-1, 7)
        result = {'boxes_lidar': so.apply_transform(bx, T_ego, box=True) if len(bx) else np.zeros((0, 7)),
                  'name': np.array([n for n, k in zip(names, keep) if k]),
                  'score': np.array([s for s, k in zip(scores, keep) if k]),
                  'moving': np.zeros(int(sum(keep)), bool)}
        tick('vote+boxes')
        self.timings = t
        return dict(ground_idx=np.sort(gidx), labels=labels, probs=probs, dets=dets, static=static, plane=plane, valid=np.array(valid),
                    u8=u8, probs_clip=pr.numpy() if len(norm) else np.zeros((0, len(self.class_list))), top1=top,
                    names=names, scores=scores, boxes_ref=np.array(boxes).reshape(-1, 7), result=result)
