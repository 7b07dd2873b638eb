"""Seeded synthetic LiDAR frames and sequences (SURVEY.md §8d "Synthetic inputs").

No dataset can be downloaded in the build/bench environment, so every BASELINE config runs on
frames from this generator: a noisy, gently sloped ground plane sampled on 64 beam rings, box-shaped
objects (car / pedestrian / cyclist / pole / wall / ...) with points on their sensor-facing faces at
a density falling with range, and uniform clutter.  Output layout is the Waymo one the reference
consumes (`get_lidar_points` -> (N,5) float32 [x, y, z, intensity, elongation], vehicle frame with the
ground near z = 0 and the sensor 1.723 m above it; zero_shot_detector.py:87, preprocessing.yaml:56).
"""
import numpy as np

OBJECT_TYPES = [
    # name, (l, w, h), relative frequency
    ('car', (4.5, 1.9, 1.6), 0.40),
    ('pedestrian', (0.6, 0.6, 1.7), 0.20),
    ('cyclist', (1.8, 0.6, 1.7), 0.10),
    ('pole', (0.3, 0.3, 4.0), 0.10),
    ('wall', (8.0, 0.3, 2.5), 0.08),
    ('truck', (8.0, 2.5, 3.2), 0.06),
    ('bush', (1.5, 1.5, 1.2), 0.06),
]
SENSOR_HEIGHT = 1.723


def _box_surface_points(rng, n, center, dims, yaw):
    """n points on the faces of an oriented box that face the sensor at the origin (height 1.723)."""
    l, w, h = dims
    c, s = np.cos(yaw), np.sin(yaw)
    Rm = np.array([[c, -s], [s, c]])
    # faces: +x, -x, +y, -y, top  (normal, area)
    normals = np.array([[1, 0, 0], [-1, 0, 0], [0, 1, 0], [0, -1, 0], [0, 0, 1]], dtype=np.float64)
    areas = np.array([w * h, w * h, l * h, l * h, l * w])
    sensor = np.array([0.0, 0.0, SENSOR_HEIGHT])
    nw = normals.copy()
    nw[:, :2] = normals[:, :2] @ Rm.T
    to_sensor = sensor - np.asarray(center)
    vis = (nw @ to_sensor) > 0
    if not vis.any():
        vis[:] = True
    p = areas * vis
    p = p / p.sum()
    face = rng.choice(5, size=n, p=p)
    u = rng.uniform(-0.5, 0.5, size=n)
    v = rng.uniform(-0.5, 0.5, size=n)
    loc = np.zeros((n, 3))
    fx = face < 2
    loc[fx, 0] = np.where(face[fx] == 0, l / 2, -l / 2)
    loc[fx, 1] = u[fx] * w
    loc[fx, 2] = v[fx] * h
    fy = (face >= 2) & (face < 4)
    loc[fy, 1] = np.where(face[fy] == 2, w / 2, -w / 2)
    loc[fy, 0] = u[fy] * l
    loc[fy, 2] = v[fy] * h
    ft = face == 4
    loc[ft, 0] = u[ft] * l
    loc[ft, 1] = v[ft] * w
    loc[ft, 2] = h / 2
    out = loc.copy()
    out[:, :2] = loc[:, :2] @ Rm.T
    out += np.asarray(center)
    out += rng.normal(0, 0.01, size=out.shape)
    return out


def make_frame(seed=0, n_points=150_000, n_objects=60, ground_frac=0.45, clutter_frac=0.05,
               return_meta=False):
    """One frame: (n_points,5) float32.  Deterministic in (seed, arguments)."""
    rng = np.random.default_rng(seed)
    n_ground = int(n_points * ground_frac)
    n_clutter = int(n_points * clutter_frac)
    n_obj_pts = n_points - n_ground - n_clutter

    # ground: 64 rings from 2.5 m to 75 m, azimuth uniform, gentle slope (<= 2 deg), z noise 3 cm
    rings = np.geomspace(2.5, 75.0, 64)
    ring_of = rng.integers(0, 64, size=n_ground)
    r = rings[ring_of] * (1 + rng.normal(0, 0.002, size=n_ground))
    th = rng.uniform(0, 2 * np.pi, size=n_ground)
    slope = np.deg2rad(rng.uniform(-2, 2, size=2))
    gx, gy = r * np.cos(th), r * np.sin(th)
    gz = np.tan(slope[0]) * gx * 0.2 + np.tan(slope[1]) * gy * 0.2 + rng.normal(0, 0.03, size=n_ground)
    ground = np.stack([gx, gy, gz], 1)

    # objects
    names = [t[0] for t in OBJECT_TYPES]
    freq = np.array([t[2] for t in OBJECT_TYPES])
    kinds = rng.choice(len(OBJECT_TYPES), size=n_objects, p=freq / freq.sum())
    rr = rng.uniform(5, 60, size=n_objects)
    tt = rng.uniform(0, 2 * np.pi, size=n_objects)
    yaw = rng.uniform(0, 2 * np.pi, size=n_objects)
    weights = np.zeros(n_objects)
    for i, k in enumerate(kinds):
        l, w, h = OBJECT_TYPES[k][1]
        weights[i] = (l * h + w * h) / (rr[i] ** 2)
    cnt = np.maximum(30, np.floor(weights / weights.sum() * n_obj_pts)).astype(int)
    # fix the total
    while cnt.sum() > n_obj_pts:
        cnt[np.argmax(cnt)] -= min(cnt.sum() - n_obj_pts, cnt.max() - 30)
    cnt[np.argmax(cnt)] += n_obj_pts - cnt.sum()
    objs, meta = [], []
    for i, k in enumerate(kinds):
        dims = OBJECT_TYPES[k][1]
        cx, cy = rr[i] * np.cos(tt[i]), rr[i] * np.sin(tt[i])
        gz0 = np.tan(slope[0]) * cx * 0.2 + np.tan(slope[1]) * cy * 0.2
        center = (cx, cy, gz0 + dims[2] / 2 + 0.02)
        objs.append(_box_surface_points(rng, int(cnt[i]), center, dims, yaw[i]))
        meta.append(dict(name=names[k], center=center, dims=dims, yaw=float(yaw[i]), n=int(cnt[i])))
    objs = np.concatenate(objs) if objs else np.zeros((0, 3))

    clutter = np.stack([rng.uniform(-75, 75, n_clutter), rng.uniform(-75, 75, n_clutter),
                        rng.uniform(0, 4, n_clutter)], 1)
    xyz = np.concatenate([ground, objs, clutter])
    perm = rng.permutation(len(xyz))
    xyz = xyz[perm]
    intensity = rng.uniform(0, 1, size=len(xyz))
    pts = np.zeros((len(xyz), 5), dtype=np.float32)
    pts[:, :3] = xyz
    pts[:, 3] = intensity
    if return_meta:
        kind = np.concatenate([np.zeros(n_ground, np.int8), np.ones(len(objs), np.int8),
                               np.full(n_clutter, 2, np.int8)])[perm]
        return pts, dict(objects=meta, point_kind=kind)
    return pts


def make_poses(n_frames, step=0.5, seed=0):
    """Smooth SE(2) trajectory, `step` metres per frame, as 4x4 float64 vehicle->world poses."""
    rng = np.random.default_rng(seed + 104729)
    yaw_rate = rng.normal(0, 0.004)
    poses = []
    x = y = yaw = 0.0
    for _ in range(n_frames):
        c, s = np.cos(yaw), np.sin(yaw)
        T = np.eye(4)
        T[:2, :2] = [[c, -s], [s, c]]
        T[0, 3], T[1, 3] = x, y
        poses.append(T)
        x += step * c
        y += step * s
        yaw += yaw_rate
    return poses


def make_sequence(seed=0, n_frames=6, n_points=20_000, n_objects=12, moving_frac=0.35, ground_frac=0.45,
                  clutter_frac=0.03, step=0.5, return_objects=False):
    """A sequence over ONE world: static objects stay put, a fraction moves along its heading (0.3-1.2 m per frame),
    the ego vehicle follows `make_poses`.  Every frame re-samples the surfaces (a LiDAR never hits the same spot
    twice), so static structure keeps a similar neighbour count from frame to frame (entropy score near 1) and
    moving objects do not (low score) -- the signal `calculate_entropy_scores` measures.
    -> (list of (n_points,5) float32 frames in the VEHICLE frame, list of 4x4 poses); with `return_objects` also, per frame,
    the objects' ground truth in the vehicle frame: dict(kind [n] str, box [n,7] = x,y,z,dx,dy,dz,heading, n_points [n], id [n], moving [n])."""
    rng = np.random.default_rng(seed + 7919)
    poses = make_poses(n_frames, step=step, seed=seed)
    freq = np.array([t[2] for t in OBJECT_TYPES])
    kinds = rng.choice(len(OBJECT_TYPES), size=n_objects, p=freq / freq.sum())
    rr = rng.uniform(5, 45, size=n_objects)
    tt = rng.uniform(0, 2 * np.pi, size=n_objects)
    yaw = rng.uniform(0, 2 * np.pi, size=n_objects)
    centers = np.stack([rr * np.cos(tt), rr * np.sin(tt)], 1)              # world
    movable = np.array([OBJECT_TYPES[k][0] in ('car', 'pedestrian', 'cyclist', 'truck') for k in kinds])
    moving = movable & (rng.uniform(size=n_objects) < moving_frac / max(movable.mean(), 1e-9))
    speed = np.where(moving, rng.uniform(0.3, 1.2, size=n_objects), 0.0)
    n_ground = int(n_points * ground_frac)
    n_clutter = int(n_points * clutter_frac)
    n_obj_pts = n_points - n_ground - n_clutter
    clutter_w = np.stack([rng.uniform(-60, 60, n_clutter), rng.uniform(-60, 60, n_clutter), rng.uniform(0.3, 4, n_clutter)], 1)
    rings = np.geomspace(2.5, 75.0, 64)
    frames, truth = [], []
    for f in range(n_frames):
        T = poses[f]
        Ti = np.linalg.inv(T)
        ego_yaw = np.arctan2(T[1, 0], T[0, 0])
        ring_of = rng.integers(0, 64, size=n_ground)
        r = rings[ring_of] * (1 + rng.normal(0, 0.002, size=n_ground))
        th = rng.uniform(0, 2 * np.pi, size=n_ground)
        ground = np.stack([r * np.cos(th), r * np.sin(th), rng.normal(0, 0.03, size=n_ground)], 1)
        cw = centers + (speed * f)[:, None] * np.stack([np.cos(yaw), np.sin(yaw)], 1)
        ce = (np.c_[cw, np.zeros(n_objects), np.ones(n_objects)] @ Ti.T)[:, :2]
        dist2 = np.maximum((ce ** 2).sum(1), 9.0)
        weights = np.array([(OBJECT_TYPES[k][1][0] + OBJECT_TYPES[k][1][1]) * OBJECT_TYPES[k][1][2] for k in kinds]) / dist2
        cnt = np.maximum(40, np.floor(weights / weights.sum() * n_obj_pts)).astype(int)
        while cnt.sum() > n_obj_pts:
            cnt[np.argmax(cnt)] -= min(cnt.sum() - n_obj_pts, cnt.max() - 40)
        cnt[np.argmax(cnt)] += n_obj_pts - cnt.sum()
        objs = [_box_surface_points(rng, int(cnt[i]), (ce[i, 0], ce[i, 1], OBJECT_TYPES[k][1][2] / 2 + 0.02),
                                    OBJECT_TYPES[k][1], yaw[i] - ego_yaw) for i, k in enumerate(kinds)]
        cl = (np.c_[clutter_w[:, :2], np.zeros(n_clutter), np.ones(n_clutter)] @ Ti.T)[:, :2]
        clutter = np.c_[cl + rng.normal(0, 0.01, size=cl.shape), clutter_w[:, 2]]
        xyz = np.concatenate([ground] + objs + [clutter])
        xyz = xyz[rng.permutation(len(xyz))]
        pts = np.zeros((len(xyz), 5), dtype=np.float32)
        pts[:, :3] = xyz
        pts[:, 3] = rng.uniform(0, 1, size=len(xyz))
        frames.append(pts)
        if return_objects:
            dims = np.array([OBJECT_TYPES[k][1] for k in kinds], dtype=np.float64)
            truth.append(dict(kind=np.array([OBJECT_TYPES[k][0] for k in kinds]), n_points=cnt.copy(), id=np.arange(n_objects), moving=moving.copy(),
                              box=np.c_[ce, dims[:, 2] / 2 + 0.02, dims, yaw - ego_yaw]))
    if return_objects:
        return frames, poses, truth
    return frames, poses
