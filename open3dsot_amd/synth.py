"""Seeded synthetic KITTI-Car-like template/search pairs with the tracker's input contract.

No dataset exists in the sandbox, so tests and bench.py use this generator (SURVEY.md
section 8d).  Shapes and keys follow datasets/sampler.py::siamese_processing (:67-78):
  template_points (B,M,3)  search_points (B,N,3)  points2cc_dist_t (B,M,9)
  points2cc_dist_s (B,N,9) seg_label (B,N) in {0,1} box_label (B,4) bbox_size (B,3)
Geometry: a car-sized box (w,l,h) ~ (1.6,3.9,1.5)*U(0.9,1.1) (wlh as in the nuScenes Box
the reference uses: w along y, l along x); the template is a LiDAR-like sampling of its
surface in the box frame; the search area is the same surface displaced by a Kalman-like
offset (U(-1,1)^2 m, +-5 deg), a ground plane and uniform clutter inside the
search_bb_scale=1.25 / search_bb_offset=2 window (cfgs/BAT_Car.yaml:5-8).  Point counts are
log-uniform and resampled to M / N exactly like datasets/points_utils.py::regularize_pc
(:24-40): without replacement when enough points, WITH replacement otherwise (duplicates),
all-zero cloud when <= 2 points.  BoxCloud = distances to centre + 8 corners (:127-143).
`sample i` depends only on (seed0 + i): any batch slice is reproducible on any rank.
"""
import numpy as np


def _box_points(wlh):
    w, l, h = wlh
    x = l / 2 * np.array([1, 1, 1, 1, -1, -1, -1, -1.0])
    y = w / 2 * np.array([1, -1, -1, 1, 1, -1, -1, 1.0])
    z = h / 2 * np.array([1, 1, -1, -1, 1, 1, -1, -1.0])
    return np.stack([x, y, z], 1)  # (8,3) corners, box frame


def _rotz(a):
    c, s = np.cos(a), np.sin(a)
    return np.array([[c, -s, 0], [s, c, 0], [0, 0, 1.0]])


def _surface(rng, wlh, n):
    """n points on the visible faces of the box (two sides + top), with 2 cm noise."""
    w, l, h = wlh
    face = rng.integers(0, 3, n)
    u, v = rng.uniform(-0.5, 0.5, n), rng.uniform(-0.5, 0.5, n)
    p = np.empty((n, 3))
    p[:, 0] = np.where(face == 0, l / 2, u * l)
    p[:, 1] = np.where(face == 1, w / 2, np.where(face == 0, u * w, v * w))
    p[:, 2] = np.where(face == 2, h / 2, v * h)
    return p + rng.normal(0, 0.02, (n, 3))


def regularize(rng, pts, size):
    n = pts.shape[0]
    if n > 2:
        sel = rng.choice(n, size=size, replace=size > n) if n != size else np.arange(n)
        return pts[sel].astype(np.float32), sel
    return np.zeros((size, 3), np.float32), None


def boxcloud(pts, center, rot, wlh):
    ref = np.concatenate([center[None, :], _box_points(wlh) @ rot.T + center], 0)  # (9,3)
    return np.linalg.norm(pts[:, None, :].astype(np.float64) - ref[None], axis=2).astype(np.float32)


def make_pair(index, template_size=512, search_size=1024, seed0=1234):
    rng = np.random.default_rng(seed0 + int(index))
    wlh = np.array([1.6, 3.9, 1.5]) * rng.uniform(0.9, 1.1, 3)
    empty_t = rng.uniform() < 0.02
    empty_s = rng.uniform() < 0.02
    # template, box frame (model_bb_scale 1.25: the crop keeps everything generated here)
    n_t = int(np.exp(rng.uniform(np.log(30), np.log(2000))))
    t_raw = _surface(rng, wlh, 0 if empty_t else n_t)
    t_pts, _ = regularize(rng, t_raw, template_size)
    t_bc = boxcloud(t_pts, np.zeros(3), np.eye(3), wlh)
    # search area, frame of the perturbed box
    off = np.array([rng.uniform(-1, 1), rng.uniform(-1, 1), np.deg2rad(rng.uniform(-5, 5))])
    rot = _rotz(-off[2])
    center = rot @ np.array([-off[0], -off[1], 0.0])
    n_s = int(np.exp(rng.uniform(np.log(50), np.log(5000))))
    n_obj = max(3, int(n_s * rng.uniform(0.1, 0.5)))
    n_gnd = int((n_s - n_obj) * 0.6)
    n_clu = max(0, n_s - n_obj - n_gnd)
    ext = wlh[[1, 0, 2]] * 1.25 + 4.0  # window extent along x (l), y (w), z (h)
    obj = _surface(rng, wlh, n_obj) @ rot.T + center
    gnd = np.stack([rng.uniform(-ext[0] / 2, ext[0] / 2, n_gnd), rng.uniform(-ext[1] / 2, ext[1] / 2, n_gnd),
                    -wlh[2] / 2 + rng.normal(0, 0.03, n_gnd)], 1)
    clu = rng.uniform(-ext / 2, ext / 2, (n_clu, 3))
    s_raw = np.concatenate([obj, gnd, clu], 0)
    lab_raw = np.concatenate([np.ones(n_obj), np.zeros(n_gnd + n_clu)])
    if empty_s:
        s_raw, lab_raw = s_raw[:0], lab_raw[:0]
    s_pts, sel = regularize(rng, s_raw, search_size)
    seg = lab_raw[sel].astype(np.float32) if sel is not None else np.zeros(search_size, np.float32)
    s_bc = boxcloud(s_pts, center, rot, wlh)
    return {
        "template_points": t_pts, "search_points": s_pts,
        "points2cc_dist_t": t_bc, "points2cc_dist_s": s_bc,
        "seg_label": seg,
        "box_label": np.array([center[0], center[1], center[2], -off[2]], np.float32),
        "bbox_size": wlh.astype(np.float32),
    }


def make_batch(first_index, batch_size, template_size=512, search_size=1024, seed0=1234):
    """Stack samples first_index .. first_index+batch_size-1 into a dict of numpy arrays."""
    items = [make_pair(first_index + i, template_size, search_size, seed0) for i in range(batch_size)]
    return {k: np.stack([it[k] for it in items], 0) for k in items[0]}


def make_dense_batch(first_index, batch_size, template_size=512, search_size=1024, seed0=1234):
    """Worst case of the data-dependent work: the same pairs with both clouds replaced by distinct points inside a
    17 cm cube (diagonal 0.294 m < the smallest ball radius 0.3 m, away from the origin so the FPS near-origin rule
    of Appendix A.1 skips nothing): every ball of every set-abstraction level is full of DISTINCT neighbours, so the
    distinct-neighbour layout of csrc/compact.hip compacts nothing (live fraction 1.0).  Labels / BoxClouds are
    recomputed for the new points; used by `bench.py --dense` only."""
    out = make_batch(first_index, batch_size, template_size, search_size, seed0)
    for i in range(batch_size):
        rng = np.random.default_rng(seed0 + 7919 * (int(first_index) + i) + 1)
        wlh = out["bbox_size"][i].astype(np.float64)
        t = (np.array([0.5, 0.3, 0.2]) + rng.uniform(-0.085, 0.085, (template_size, 3))).astype(np.float32)
        c = out["box_label"][i]
        rot, center = _rotz(c[3]), c[:3].astype(np.float64)
        sp = (center + np.array([0.5, 0.3, 0.2]) + rng.uniform(-0.085, 0.085, (search_size, 3))).astype(np.float32)
        out["template_points"][i], out["search_points"][i] = t, sp
        out["points2cc_dist_t"][i] = boxcloud(t, np.zeros(3), np.eye(3), wlh)
        out["points2cc_dist_s"][i] = boxcloud(sp, center, rot, wlh)
        out["seg_label"][i] = 1.0
    return out


def make_motion_batch(first_index, batch_size, point_sample_size=1024, seed0=4321):
    """Synthetic M2-Track input (datasets/sampler.py::motion_processing :143-179): two consecutive
    frames of one target, each resampled to `point_sample_size` points, stacked as
    points (B,2N,5) = [xyz, time stamp (0 | 0.1), prior-box mask (previous frame only)], candidate_bc
    (B,2N,9), seg_label (B,2N) int64, box_label / box_label_prev / motion_label (B,4),
    motion_state_label (B,) int64, prev_bc / this_bc (B,N,9)."""
    out = []
    N = point_sample_size
    for i in range(batch_size):
        rng = np.random.default_rng(seed0 + int(first_index) + i)
        wlh = np.array([1.6, 3.9, 1.5]) * rng.uniform(0.9, 1.1, 3)
        moving = rng.uniform() < 0.6
        motion = np.array([rng.uniform(0.2, 1.0) if moving else rng.uniform(0, 0.1), rng.uniform(-0.1, 0.1), 0.0,
                           np.deg2rad(rng.uniform(-5, 5))])
        prev_off = np.array([rng.uniform(-0.3, 0.3), rng.uniform(-0.3, 0.3), 0.0, np.deg2rad(rng.uniform(-3, 3))])
        frames, labels, bcs = [], [], []
        for t, pose in enumerate((prev_off, prev_off + motion)):
            rot, ctr = _rotz(pose[3]), pose[:3]
            n_obj, n_bg = int(rng.integers(20, 400)), int(rng.integers(100, 1500))
            ext = wlh[[1, 0, 2]] * 1.25 + 4.0
            pts = np.concatenate([_surface(rng, wlh, n_obj) @ rot.T + ctr, rng.uniform(-ext / 2, ext / 2, (n_bg, 3))], 0)
            lab = np.concatenate([np.ones(n_obj), np.zeros(n_bg)])
            p, sel = regularize(rng, pts, N)
            frames.append(p)
            labels.append(lab[sel])
            bcs.append(boxcloud(p, ctr, rot, wlh))
        stamp = np.concatenate([np.zeros(N), np.full(N, 0.1)])[:, None]
        prior = np.concatenate([labels[0] * 0.8 + 0.1, np.full(N, 0.15)])[:, None]
        out.append({
            "points": np.concatenate([np.concatenate(frames, 0), stamp, prior], 1).astype(np.float32),
            "candidate_bc": np.concatenate([bcs[0], np.zeros((N, 9))], 0).astype(np.float32),
            "seg_label": np.concatenate(labels).astype(np.int64),
            "box_label": (prev_off + motion).astype(np.float32), "box_label_prev": prev_off.astype(np.float32),
            "motion_label": motion.astype(np.float32), "motion_state_label": np.int64(moving),
            "prev_bc": bcs[0].astype(np.float32), "this_bc": bcs[1].astype(np.float32)})
    return {k: np.stack([o[k] for o in out], 0) for k in out[0]}


def to_torch(batch, device=None):
    import torch
    return {k: torch.from_numpy(v).to(device) if device is not None else torch.from_numpy(v)
            for k, v in batch.items()}
