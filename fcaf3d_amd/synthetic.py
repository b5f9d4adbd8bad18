"""Synthetic indoor scenes shaped like the reference's datasets.

The reference's dataloader hands the detector, per scene, a float32 tensor of
exactly 100 000 rows ``[x, y, z, r, g, b]`` (IndoorPointSample,
mmdet3d/datasets/pipelines/transforms_3d.py:856-863) plus GT boxes as a
``DepthInstance3DBoxes`` and int64 labels.  Real ScanNet / SUN RGB-D / S3DIS are
not available, so the benchmark and the parity tests use this generator.

Pure numpy, deterministic in ``seed``.
"""
import numpy as np


def _rect(origin, u, v):
    return dict(o=np.asarray(origin, np.float64), u=np.asarray(u, np.float64),
                v=np.asarray(v, np.float64))


def _sample_rects(rects, n, rng, noise):
    areas = np.array([np.linalg.norm(np.cross(r['u'], r['v'])) for r in rects])
    which = rng.choice(len(rects), size=n, p=areas / areas.sum())
    a = rng.random(n)
    b = rng.random(n)
    o = np.stack([r['o'] for r in rects])[which]
    u = np.stack([r['u'] for r in rects])[which]
    v = np.stack([r['v'] for r in rects])[which]
    pts = o + a[:, None] * u + b[:, None] * v
    pts += rng.normal(0.0, noise, size=pts.shape)
    return pts


def _box_faces(c, d, yaw=0.0):
    """5 visible faces (top + 4 sides) of a cuboid standing on the floor."""
    cx, cy, cz = c
    w, l, h = d
    cs, sn = np.cos(yaw), np.sin(yaw)
    ex = np.array([cs, sn, 0.0]) * w
    ey = np.array([-sn, cs, 0.0]) * l
    ez = np.array([0.0, 0.0, h])
    base = np.array([cx, cy, cz - h / 2]) - ex / 2 - ey / 2
    return [
        _rect(base + ez, ex, ey),          # top
        _rect(base, ex, ez),               # side y-
        _rect(base + ey, ex, ez),          # side y+
        _rect(base, ey, ez),               # side x-
        _rect(base + ex, ey, ez),          # side x+
    ]


def make_scene(seed, n_points=100000, n_boxes=15, n_classes=18,
               room=(6.0, 5.0, 2.7), rotated=False, single_view=False,
               rgb_unit=False, noise=0.005):
    """One scene.

    Returns ``points (n_points, 6) float32``, ``gt_boxes (m, 7) float32`` with
    gravity centre ``(cx, cy, cz, w, l, h, yaw)`` and ``labels (m,) int64``.
    """
    rng = np.random.default_rng(seed)
    X, Y, Z = room
    rects = [
        _rect((0, 0, 0), (X, 0, 0), (0, Y, 0)),      # floor
        _rect((0, 0, 0), (X, 0, 0), (0, 0, Z)),      # wall y=0
        _rect((0, Y, 0), (X, 0, 0), (0, 0, Z)),      # wall y=Y
        _rect((0, 0, 0), (0, Y, 0), (0, 0, Z)),      # wall x=0
        _rect((X, 0, 0), (0, Y, 0), (0, 0, Z)),      # wall x=X
    ]
    boxes = []
    for _ in range(n_boxes):
        d = rng.uniform([0.4, 0.4, 0.4], [1.8, 1.2, 1.5])
        yaw = rng.uniform(-np.pi, np.pi) if rotated else 0.0
        r = 0.5 * np.hypot(d[0], d[1]) if rotated else 0.0
        lo = np.array([max(d[0] / 2, r), max(d[1] / 2, r)])
        hi = np.array([X, Y]) - lo
        cxy = rng.uniform(lo, np.maximum(hi, lo + 1e-3))
        c = np.array([cxy[0], cxy[1], d[2] / 2])
        boxes.append(np.concatenate([c, d, [yaw]]))
        rects.extend(_box_faces(c, d, yaw))
    if single_view:
        # keep only surfaces whose outward side faces a camera in a room corner
        cam = np.array([0.2, 0.2, 1.5])
        kept = []
        for r in rects:
            centre = r['o'] + 0.5 * (r['u'] + r['v'])
            dist = np.linalg.norm(centre - cam)
            if 0.5 <= dist <= 6.0:
                kept.append(r)
        rects = kept[: max(3, (len(kept) * 2) // 3)]
    xyz = _sample_rects(rects, n_points, rng, noise)
    if rgb_unit:
        rgb = rng.random((n_points, 3))
    else:
        rgb = rng.integers(0, 256, size=(n_points, 3)).astype(np.float64)
    points = np.concatenate([xyz, rgb], axis=1).astype(np.float32)
    gt = np.stack(boxes).astype(np.float32) if boxes else np.zeros((0, 7), np.float32)
    labels = rng.integers(0, n_classes, size=n_boxes).astype(np.int64)
    return points, gt, labels


def make_batch(seeds, **kw):
    pts, gts, lbs = [], [], []
    for s in seeds:
        p, g, l = make_scene(s, **kw)
        pts.append(p)
        gts.append(g)
        lbs.append(l)
    return pts, gts, lbs


WORKLOADS = {
    # name -> (make_scene kwargs, model overrides)
    'plumbing-20k': dict(scene=dict(n_points=20000), n_classes=18),
    'scannet-100k': dict(scene=dict(n_points=100000), n_classes=18),
    'sunrgbd-100k': dict(scene=dict(n_points=100000, n_boxes=6, n_classes=10,
                                    rotated=True, single_view=True,
                                    rgb_unit=True), n_classes=10),
    's3dis-500k': dict(scene=dict(n_points=500000, room=(12.0, 10.0, 2.7),
                                  n_boxes=30, n_classes=5), n_classes=5),
}
