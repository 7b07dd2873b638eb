"""Line-by-line restatement of the motion-vector and motion-aligned-box steps of the reference's track branch of
`fit_bounding_boxes_simple` (src/vilgod/zero_shot_detector.py:491-659) -- TEST INFRASTRUCTURE ONLY.

These were the product's implementation until round 3 (pinned against the reference's own run by tests/golden/track_golden.pkl);
vilgod_amd/tracking.py now evaluates the same arithmetic with array operations per track, and tests/test_tracking.py requires it
to reproduce these functions bit for bit on seeded tracks.
"""
import numpy as np


def _angle_deg(v1, v2):
    """common_utils.py:73-76 (a zero vector gives nan, which fails every comparison below, as upstream)."""
    with np.errstate(invalid='ignore', divide='ignore'):
        cos = v1 @ v2 / (np.linalg.norm(v1) * np.linalg.norm(v2))
    return np.rad2deg(np.arccos(np.clip(cos, -0.9999, 0.9999)))


def motion_vectors(centers_xy, look_ahead=10, min_far=0.5, min_step=0.3, max_angle=60):
    """Direction of travel per track entry from the cluster medians (zero_shot_detector.py:491-566).
    For entry c: `far` = vector to the entry look_ahead-1 steps on (searched further while shorter than 0.5 m and none is known
    yet; otherwise the last good one is kept); the steps to the entries in between that point within 60 deg of `far` and are
    longer than 0.3 m are averaged with weights 0.95^(i+1) / sum 0.9^(i+1) (i = ABSOLUTE entry index) and blended 50:50 with
    the previous direction; no such step -> previous direction, else `far`.  Any entry without a `far` vector voids the whole
    track (-> []), which sends it down the static path.  float32 throughout, like the medians."""
    n = len(centers_xy)
    out = []
    far = None
    for c in range(n):
        here = centers_xy[c]
        c_far = min(c + look_ahead - 1, n - 1)
        cand = np.array([centers_xy[c_far, 0] - here[0], centers_xy[c_far, 1] - here[1]])
        if np.linalg.norm(cand) < min_far and far is None:
            k = 1
            while np.linalg.norm(cand) < min_far and (c_far + k) < n:
                cand = np.array([centers_xy[c_far + k, 0] - here[0], centers_xy[c_far + k, 1] - here[1]])
                k += 1
            if np.linalg.norm(cand) >= min_far:
                far = cand
        elif np.linalg.norm(cand) < min_far:
            pass                                         # slow stretch in the middle / at the end: keep the last direction
        else:
            far = cand
        if far is None:
            return []
        steps, wsum = [], 0
        for i in range(c + 1, c_far):
            step = np.array([centers_xy[i, 0] - here[0], centers_xy[i, 1] - here[1]])
            if _angle_deg(far, step) < max_angle and np.linalg.norm(step) > min_step:
                steps.append(step * (0.95 ** (i + 1)))
                wsum += (0.9 ** (i + 1))
        if steps:
            v = np.mean(steps, axis=0) / wsum
            if out:
                v = v * 0.5 + out[-1] * 0.5
            out.append(v)
        elif out:
            out.append(out[-1])
        else:
            out.append(far)
    return out


def moving_boxes(points_list, directions, to_ego_list, top_k=3, centers3=None):
    """zero_shot_detector.py:572-659: a box per entry aligned with its direction of travel, all resized to the median size of the
    top_k entries with the most points and shifted so that the corner closest to the ego vehicle stays where it was."""
    from scipy.spatial.transform import Rotation as R
    boxes, corner_list = [], []
    for j, (pts_all, d) in enumerate(zip(points_list, directions)):
        angle = np.arctan2(d[1], d[0])
        rot = R.from_euler('z', angle, degrees=False).as_matrix()
        center = np.median(pts_all[..., :3], axis=0) if centers3 is None else np.asarray(centers3[j], dtype=pts_all.dtype)
        proj = np.dot(pts_all[..., :3] - center, rot)
        min_x, max_x = proj[:, 0].min(), proj[:, 0].max()
        min_y, max_y = proj[:, 1].min(), proj[:, 1].max()
        rect = np.array([[max_x, min_y], [min_x, min_y], [min_x, max_y], [max_x, max_y]], dtype=np.float32)
        corners = np.dot(rect, rot[:2, :2].T)
        corners += center[:2]
        w = np.linalg.norm(corners[0] - corners[1])
        l = np.linalg.norm(corners[0] - corners[-1])
        c = (corners[0] + corners[2]) / 2
        corner_list.append(corners)
        height = pts_all[:, 2].max() - pts_all[:, 2].min()
        boxes.append(np.array([c[0], c[1], pts_all[:, 2].min() + height / 2, w, l, height, angle]))
    boxes = np.array(boxes)
    top = np.argsort([len(p) for p in points_list])[-top_k:]
    ref = np.median(boxes[top], axis=0)
    tops = np.array([np.max(p[..., 2]) for p in points_list])
    for i, (corners, T) in enumerate(zip(corner_list, to_ego_list)):
        h = np.hstack((np.concatenate([corners, np.zeros((4, 1))], axis=1), np.ones((4, 1))))
        ego = np.einsum('ij,kj->ki', T, h)[:, :2]                      # apply_transform (pointcloud_utils.py:21-46)
        cc = int(np.linalg.norm(ego, axis=1).argmin())
        dw, dl = ref[3] - boxes[i, 3], ref[4] - boxes[i, 4]
        ang = np.arctan2(directions[i][1], directions[i][0])
        sx = -1.0 if cc in (0, 3) else 1.0                             # corners 0,3 hold max x: grow towards -x
        sy = 1.0 if cc in (0, 1) else -1.0                             # corners 0,1 hold min y: grow towards +y
        boxes[i, 0] += sx * (dw / 2) * np.cos(ang)
        boxes[i, 1] += sx * (dw / 2) * np.sin(ang)
        boxes[i, 0] += sy * (dl / 2) * np.sin(-ang)
        boxes[i, 1] += sy * (dl / 2) * np.cos(-ang)
    boxes[..., 3:6] = ref[3:6]
    boxes[..., 2] = tops - (ref[5] / 2)
    return boxes


