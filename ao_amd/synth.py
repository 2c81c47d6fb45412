"""Synthetic S3DIS-shaped inputs.

There is no dataset in the build or bench environment, so the workload is a
procedural "room": an axis-aligned box with furniture boxes inside, sampled on
its surfaces, then pushed through a restatement of the reference's own train
pipeline pieces that fix the point count and density
(configs/s3dis/semseg-pt-v2m2-0-base.py:88-96): GridSample(0.04) -> SphereCrop(point_max)
-> CenterShift.  numpy only; deterministic per seed.
"""
import numpy as np

ROOMS = ((6.0, 5.0, 3.0), (10.0, 8.0, 3.0), (14.0, 10.0, 3.2))


def _box_faces(lo, hi, skip_bottom):
    lo, hi = np.asarray(lo, np.float64), np.asarray(hi, np.float64)
    faces = []
    for ax in range(3):
        for side in (0, 1):
            if skip_bottom and ax == 2 and side == 0:
                continue
            faces.append((ax, lo[ax] if side == 0 else hi[ax], lo, hi))
    return faces


def _sample_faces(faces, density, rng):
    out = []
    for ax, val, lo, hi in faces:
        u, v = [a for a in range(3) if a != ax]
        area = (hi[u] - lo[u]) * (hi[v] - lo[v])
        cnt = int(area * density)
        p = np.empty((cnt, 3))
        p[:, ax] = val
        p[:, u] = rng.uniform(lo[u], hi[u], cnt)
        p[:, v] = rng.uniform(lo[v], hi[v], cnt)
        out.append(p)
    return np.concatenate(out, 0)


def voxel_dedupe(coord, size):
    """One point per occupied voxel (first in input order), as GridSample(train) keeps one per voxel."""
    key = np.floor(coord / size).astype(np.int64)
    key -= key.min(0)
    dims = key.max(0) + 1
    lin = (key[:, 0] * dims[1] + key[:, 1]) * dims[2] + key[:, 2]
    _, first = np.unique(lin, return_index=True)
    return coord[np.sort(first)]


def room_scene(seed=0, room=0, point_max=80000, density=4000.0, voxel=0.04):
    """(coord fp32 (N,3)) for one cropped scene, N <= point_max."""
    rng = np.random.default_rng(seed)
    L, W, H = ROOMS[room % len(ROOMS)]
    faces = _box_faces((0, 0, 0), (L, W, H), skip_bottom=False)
    for _ in range(int(L * W / 3)):
        sz = rng.uniform((0.4, 0.4, 0.4), (1.8, 1.2, 1.6))
        org = rng.uniform((0, 0, 0), (L - sz[0], W - sz[1], 0.0))
        faces += _box_faces(org, org + sz, skip_bottom=True)
    pts = _sample_faces(faces, density, rng)
    pts = voxel_dedupe(pts, voxel)
    if pts.shape[0] > point_max:  # SphereCrop: keep the point_max nearest to a random centre
        centre = pts[rng.integers(pts.shape[0])]
        d2 = ((pts - centre) ** 2).sum(1)
        keep = np.sort(np.argpartition(d2, point_max)[:point_max])
        pts = pts[keep]
    pts = pts - np.concatenate([pts[:, :2].mean(0), pts[:, 2:].min(0)])  # CenterShift(apply_z=True)
    return np.ascontiguousarray(pts, dtype=np.float32)


def room_cloud(n, seed=0):
    """Small dense patch of a room with exactly n points (unit tests)."""
    pts = room_scene(seed=seed, room=0, point_max=n, density=1500.0)
    assert pts.shape[0] == n, (pts.shape, n)
    return pts


def random_cloud(n, seed=0, scale=2.0):
    rng = np.random.default_rng(seed)
    return np.ascontiguousarray(rng.uniform(-scale, scale, (n, 3)), dtype=np.float32)


def lattice_cloud(nx, ny, nz, step):
    """Regular lattice: every query has many exactly tied neighbours."""
    g = np.stack(np.meshgrid(np.arange(nx), np.arange(ny), np.arange(nz), indexing="ij"), -1).reshape(-1, 3)
    return np.ascontiguousarray(g * step, dtype=np.float32)


def scene_batch(seeds, point_max=80000, in_channels=6, num_classes=13, ignore_frac=0.1, room=None):
    """Collated batch in the reference's offset format (pointcept/datasets/utils.py:14-40):
    dict(coord (N,3) f32, feat (N,in_channels) f32, segment (N,) i64, offset (B,) i32 cumulative)."""
    coords, feats, labels, counts = [], [], [], []
    for i, seed in enumerate(seeds):
        c = room_scene(seed=seed, room=seed if room is None else room, point_max=point_max)
        rng = np.random.default_rng(10_000 + seed)
        extra = rng.uniform(-1, 1, (c.shape[0], in_channels - 3)).astype(np.float32)
        lab = rng.integers(0, num_classes, c.shape[0]).astype(np.int64)
        lab[rng.uniform(size=c.shape[0]) < ignore_frac] = -1
        coords.append(c)
        feats.append(np.concatenate([c, extra], 1))
        labels.append(lab)
        counts.append(c.shape[0])
    return dict(
        coord=np.concatenate(coords, 0),
        feat=np.ascontiguousarray(np.concatenate(feats, 0), dtype=np.float32),
        segment=np.concatenate(labels, 0),
        offset=np.cumsum(counts).astype(np.int32),
    )
