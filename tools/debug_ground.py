import sys, numpy as np, torch
sys.path.insert(0, '.')
from oracle import patchworkpp as opw
from vilgod_amd import patchworkpp as gpw
G='tests/golden'
po, pg = opw.Parameters(), gpw.Parameters()
po.min_range = pg.min_range = 1.5
o = opw.patchworkpp(po); g = gpw.patchworkpp(pg, max_points=130000, device='cuda:0')
for k in range(6):
    pts = np.fromfile(f'{G}/kitti_00000{k}.bin', dtype=np.float32).reshape(-1, 4)
    so_before, sg_before = o.state(), g.state()
    want = np.sort(opw.mask_ground_points(pts, o, 0.0)); got = np.sort(gpw.mask_ground_points_patchwork_pp(pts, g, 0.0))
    io, ig = o.patch_info(), g.patch_info()
    io, ig = io[:, :11], ig[:, :11]
    d = np.flatnonzero(~np.all((io == ig) | (np.isnan(io) & np.isnan(ig)), axis=1))
    print('frame', k, 'want', len(want), 'got', len(got), 'xor', len(np.setxor1d(want, got)), 'patches differ', d.tolist())
    for p in d[:6]:
        print('  patch', p, '\n   oracle', io[p], '\n   gpu   ', ig[p])
    so, sg = o.state(), g.state()
    for key in so:
        if not np.array_equal(np.asarray(so[key]), np.asarray(sg[key])):
            print('  state differs', key, so[key], sg[key])
    if len(np.setxor1d(want, got)):
        x = np.setxor1d(want, got)
        print('  differing points', x[:10], pts[x[:10]])
        print('  state before frame: oracle', so_before, '\n  gpu', sg_before)
        break
