"""Deterministic synthetic multi-traversal LiDAR scenes.

No dataset ships with the reference (its ``valid_idx_info.pkl`` blobs are
missing), so tests and ``bench.py`` run on seeded synthetic scenes shaped like
the data the reference processes: a static world (ground plane, walls, parked
boxes) observed from an ego vehicle driving along +x, re-observed by T
historical traversals of F frames each (2 m apart, lateral jitter), plus mobile
boxes that exist only in the live scan.  Points are produced in each frame's
KITTI-velodyne coordinates; poses are produced as ``oxts`` / ``l2e`` so that the
whole CLI path (``get_relative_pose`` included) can be exercised.

This is data generation, not part of the hot path.
"""
from __future__ import annotations

import os
import pickle
from dataclasses import dataclass, field
from typing import Dict, List, Sequence, Tuple

import numpy as np
from scipy.spatial.transform import Rotation as R

SENSOR_H = 1.7          # lidar height above the ground plane [m]
ROAD_HALF_WIDTH = 12.0  # walls at y_w = +-12 m


def rot_z(angle: float) -> np.ndarray:
    t = np.eye(4)
    c, s = np.cos(angle), np.sin(angle)
    t[:2, :2] = [[c, -s], [s, c]]
    return t


def kitti2nu(nusc: bool) -> np.ndarray:
    """4x4 rotation about z by pi (Lyft) or pi/2 (nuScenes)."""
    return rot_z(np.pi / 2 if nusc else np.pi)


@dataclass
class World:
    """Static surfaces, fixed for a scene seed."""
    boxes: np.ndarray          # (B,6) cx,cy,cz,l,w,h  axis aligned, world frame
    poles: np.ndarray          # (P,2) x,y
    seed: int = 0


def make_world(seed: int, length: float = 400.0) -> World:
    """Static boxes and poles along x in [-100, length]; their number grows with the length (160 boxes and 120
    poles per 500 m, the density every fixture was generated with at the default length)."""
    rng = np.random.default_rng(10_000 + seed)
    nb = int(round(160 * (length + 100.0) / 500.0))
    side = rng.choice([-1.0, 1.0], nb)
    cx = rng.uniform(-100.0, length, nb)
    cy = side * rng.uniform(5.5, 10.5, nb)
    l = rng.uniform(3.5, 9.0, nb)
    w = rng.uniform(1.6, 3.0, nb)
    h = rng.uniform(1.3, 3.5, nb)
    boxes = np.stack([cx, cy, h / 2, l, w, h], axis=1)
    npole = int(round(120 * (length + 100.0) / 500.0))
    poles = np.stack([rng.uniform(-100.0, length, npole), rng.choice([-1.0, 1.0], npole) * rng.uniform(4.0, 11.5, npole)], 1)
    return World(boxes=boxes, poles=poles, seed=seed)


def _sample_box_surface(rng, box, m):
    """m points on the 4 vertical faces + top of an axis-aligned box."""
    cx, cy, cz, l, w, h = box
    face = rng.integers(0, 5, m)
    u = rng.uniform(-0.5, 0.5, m)
    v = rng.uniform(-0.5, 0.5, m)
    x = np.where(face == 0, -0.5, np.where(face == 1, 0.5, u)) * l + cx
    y = np.where(face == 2, -0.5, np.where(face == 3, 0.5, np.where(face < 2, v, u * 0 + v))) * w + cy
    z = np.where(face == 4, 0.5, rng.uniform(-0.5, 0.5, m)) * h + cz
    return np.stack([x, y, z], 1)


def _pose_matrix(x, y, yaw, roll=0.0, pitch=0.0):
    t = np.eye(4)
    t[:3, :3] = R.from_euler("xyz", [roll, pitch, yaw]).as_matrix()
    t[:3, 3] = [x, y, 0.0]
    return t


def default_l2e() -> np.ndarray:
    t = np.eye(4)
    t[:3, 3] = [1.2, 0.0, SENSOR_H]
    return t


def sample_frame(world: World, seed: int, n: int, ego_pose: np.ndarray, l2e: np.ndarray,
                 nusc: bool = False, mobiles: np.ndarray | None = None,
                 max_range: float = 80.0, point_order: str = "shuffled") -> np.ndarray:
    """One LiDAR frame, (n,4) float32 in KITTI-velodyne coordinates
    (x forward, y left, z up; intensity in column 3)."""
    rng = np.random.default_rng(seed)
    sensor2world = ego_pose @ l2e @ kitti2nu(nusc)
    world2sensor = np.linalg.inv(sensor2world)
    origin = sensor2world[:3, 3]

    n_mob = 0
    if mobiles is not None and len(mobiles):
        n_mob = int(0.06 * n)
    n_wall = int(0.14 * n)
    n_box = int(0.10 * n)
    n_pole = int(0.02 * n)
    n_ground = n - n_mob - n_wall - n_box - n_pole

    parts = []
    # ground: lidar-like polar sampling (log-uniform range, uniform azimuth)
    rr = np.exp(rng.uniform(np.log(2.0), np.log(max_range), n_ground))
    az = rng.uniform(-np.pi, np.pi, n_ground)
    gx = origin[0] + rr * np.cos(az)
    gy = origin[1] + rr * np.sin(az)
    gz = 0.02 * rng.standard_normal(n_ground) + 0.004 * (gx - origin[0]) * 0  # flat world
    parts.append(np.stack([gx, gy, gz], 1))
    # walls (vertical sheets along the road)
    wx = origin[0] + rng.uniform(-max_range, max_range, n_wall)
    wy = rng.choice([-ROAD_HALF_WIDTH, ROAD_HALF_WIDTH], n_wall) + 0.02 * rng.standard_normal(n_wall)
    wz = rng.uniform(0.0, 4.0, n_wall)
    parts.append(np.stack([wx, wy, wz], 1))
    # static boxes within range, sampled proportionally
    d = np.abs(world.boxes[:, 0] - origin[0])
    near = np.where(d < max_range * 0.9)[0]
    if len(near) == 0:
        near = np.arange(len(world.boxes))
    which = rng.choice(near, n_box)
    pts = np.empty((n_box, 3))
    for b in np.unique(which):
        sel = which == b
        pts[sel] = _sample_box_surface(rng, world.boxes[b], int(sel.sum()))
    pts += 0.01 * rng.standard_normal(pts.shape)
    parts.append(pts)
    # poles
    dpole = np.abs(world.poles[:, 0] - origin[0])
    nearp = np.where(dpole < max_range * 0.9)[0]
    if len(nearp) == 0:
        nearp = np.arange(len(world.poles))
    wp = rng.choice(nearp, n_pole)
    ang = rng.uniform(0, 2 * np.pi, n_pole)
    parts.append(np.stack([world.poles[wp, 0] + 0.1 * np.cos(ang), world.poles[wp, 1] + 0.1 * np.sin(ang),
                           rng.uniform(0, 5.0, n_pole)], 1))
    # mobile objects (live scan only)
    if n_mob:
        dm = np.linalg.norm(mobiles[:, :2] - origin[None, :2], axis=1)
        wgt = 1.0 / np.maximum(dm, 3.0) ** 1.5
        wm = rng.choice(len(mobiles), n_mob, p=wgt / wgt.sum())
        mp = np.empty((n_mob, 3))
        for b in np.unique(wm):
            sel = wm == b
            mp[sel] = _sample_box_surface(rng, mobiles[b], int(sel.sum()))
        mp += 0.01 * rng.standard_normal(mp.shape)
        parts.append(mp)
    pw = np.concatenate(parts, 0)
    ps = pw @ world2sensor[:3, :3].T + world2sensor[:3, 3]
    out = np.empty((n, 4), dtype=np.float32)
    out[:, :3] = ps.astype(np.float32)
    out[:, 3] = rng.uniform(0, 1, n).astype(np.float32)
    perm = rng.permutation(n)   # drawn in both modes: the point SET does not depend on the order
    if point_order == "azimuth":   # spinning-LiDAR firing order: azimuth-major, ranges interleaved
        return out[np.argsort(np.arctan2(out[:, 1], out[:, 0]), kind="stable")]
    return out[perm]


def make_mobiles(seed: int, ego_x: float, count: int = 16) -> np.ndarray:
    rng = np.random.default_rng(20_000 + seed)
    ncar = count * 2 // 3
    nped = count - ncar
    rows = []
    for _ in range(ncar):
        rows.append([ego_x + rng.uniform(-45, 60), rng.uniform(-4.0, 4.0), 0.75, 4.0, 1.8, 1.5])
    for _ in range(nped):
        rows.append([ego_x + rng.uniform(-25, 35), rng.uniform(-4.8, 4.8), 0.85, 0.6, 0.6, 1.7])
    return np.asarray(rows)


@dataclass
class ScanInputs:
    """Array-level inputs of one PP-score unit, already in the common frame."""
    live_raw: np.ndarray              # (N,4) f32 KITTI-velodyne frame of the live scan
    live_xyz: np.ndarray              # (N,3) f32 transformed into the first-history-frame system
    hist: List[np.ndarray]            # T arrays (M_t,3) f32 transformed
    frames: List[List[Tuple[np.ndarray, np.ndarray, np.ndarray]]] = field(default_factory=list)  # raw frame, 4x4 f32 relative pose, 4x4 f64 raw->world
    live_rel: np.ndarray = None       # (4,4) f32 relative pose of the live scan
    live_W: np.ndarray = None         # (4,4) f64 live raw frame -> world
    world_from_ref: np.ndarray = None  # (4,4) f64 common (first history frame) system -> world
    first_pose: np.ndarray = None      # (4,4) f64 ego pose of the first history frame (fixed_ego of get_relative_pose)
    l2e: np.ndarray = None             # (4,4) f64 lidar -> ego (fixed_l2e and every frame's query_l2e)
    K: np.ndarray = None               # (4,4) KITTI2NU


def _transform_f32(pts_xyz: np.ndarray, T: np.ndarray) -> np.ndarray:
    hom = np.hstack((pts_xyz.astype(np.float32), np.ones((pts_xyz.shape[0], 1), dtype=np.float32)))
    return np.dot(hom, np.transpose(T.astype(np.float32))).reshape(-1, 4)[:, 0:3]


def relative_pose(fixed_l2e, fixed_ego, query_l2e, query_ego, K) -> np.ndarray:
    return np.linalg.solve(K, np.linalg.solve(fixed_l2e, np.linalg.solve(fixed_ego, query_ego @ query_l2e @ K))
                           ).astype(np.float32)


def make_scan(scan_id: int, n_live: int = 30_000, n_trav: int = 10, n_frames: int = 36,
              n_per_frame: int | None = None, nusc: bool = False, frame_gap: float = 2.0,
              keep_frames: bool = False, world_seed: int = 0, point_order: str = "shuffled") -> ScanInputs:
    """Seeds: live 1000+scan_id, history 2000+scan_id*1000+t*100+f."""
    n_per_frame = n_live if n_per_frame is None else n_per_frame
    world = make_world(world_seed)
    ego_x = 5.0 * scan_id % 200.0
    K = kitti2nu(nusc)
    l2e = default_l2e()
    live_pose = _pose_matrix(ego_x, 0.0, 0.01)
    mobiles = make_mobiles(scan_id, ego_x)
    live_raw = sample_frame(world, 1000 + scan_id, n_live, live_pose, l2e, nusc, mobiles, point_order=point_order)
    first_pose = None
    hist, frames = [], []
    rngp = np.random.default_rng(30_000 + scan_id)
    for t in range(n_trav):
        lat = rngp.uniform(-1.5, 1.5)
        yaw = rngp.uniform(-0.02, 0.02)
        q, fr = [], []
        for f in range(n_frames):
            pose = _pose_matrix(ego_x + frame_gap * f + rngp.uniform(-0.5, 0.5), lat, yaw)
            if first_pose is None:
                first_pose = pose
            raw = sample_frame(world, 2000 + scan_id * 1000 + t * 100 + f, n_per_frame, pose, l2e, nusc,
                               point_order=point_order)
            rel = relative_pose(l2e, first_pose, l2e, pose, K)
            xyz = raw[:, :3]
            if nusc:
                m = (xyz[:, 0] < 1.75) & (xyz[:, 0] >= -1.15) & (xyz[:, 1] < 0.65) & (xyz[:, 1] >= -0.65)
                xyz = xyz[~m]
            q.append(_transform_f32(xyz, rel))
            if keep_frames:
                fr.append((raw, rel, pose @ l2e @ K))
        hist.append(np.concatenate(q).astype(np.float32))
        frames.append(fr)
    rel_live = relative_pose(l2e, first_pose, l2e, live_pose, K)
    live_xyz = _transform_f32(live_raw[:, :3], rel_live)
    return ScanInputs(live_raw=live_raw, live_xyz=np.ascontiguousarray(live_xyz), hist=hist, frames=frames,
                      live_rel=rel_live, live_W=live_pose @ l2e @ K, world_from_ref=first_pose @ l2e @ K,
                      first_pose=first_pose, l2e=l2e, K=K)


@dataclass
class ShardScan:
    """One live scan of a synthetic shard: which frames of the shard's traversals are its history."""
    index: int
    live_raw: np.ndarray             # (N,4) f32
    live_W: np.ndarray               # (4,4) f64 raw -> world
    live_rel: np.ndarray             # (4,4) f32
    hist: List[Tuple[int, int]]      # (traversal, frame number in the traversal's track), reference order
    rels: np.ndarray                 # (T*F,4,4) f32 relative poses, same order
    world_from_ref: np.ndarray       # (4,4) f64 common frame -> world
    first_pose: np.ndarray = None
    l2e: np.ndarray = None
    K: np.ndarray = None
    travs: List[int] = None          # traversal index of every history entry IN THIS SCAN's list (None: the track number) -- a
    #                                  traversal is accepted per scan (split_traintest.py:17,79), so a track may be traversal 3 of one
    #                                  scan, traversal 2 of the next and absent from a third

    def trav_list(self) -> List[int]:
        return [t for t, _ in self.hist] if self.travs is None else list(self.travs)

    @property
    def n_trav(self) -> int:
        return max(self.trav_list()) + 1 if self.hist else 0


def presence_ramp(n_scans: int, n_trav: int, t_min: int = 2, seed: int = 0) -> List[Tuple[int, int]]:
    """Per track the scans [i0, i1) it is a traversal of: tracks 0 and 1 always (the reference needs two, split_traintest.py:111),
    the others enter and leave at seeded scan numbers -- T moves between t_min and n_trav along the shard."""
    rng = np.random.default_rng([63_000 + seed, n_scans, n_trav])
    out = [(0, n_scans)] * min(t_min, n_trav)
    for t in range(t_min, n_trav):
        a, b = sorted(int(v) for v in rng.integers(1, max(n_scans, 2), 2))
        kind = t % 3
        if kind == 0:
            out.append((0, b))                      # leaves
        elif kind == 1:
            out.append((a, n_scans))                # enters
        else:
            out.append((a, max(b, a + 1)))          # passes by
    return out


@dataclass
class Shard:
    """S consecutive live scans of one sequence and the T history traversals they look at: scan i uses frames
    i .. i+F-1 of every traversal's track, so consecutive scans share F-1 of their F frames per traversal (the
    structure data_preprocessing/lyft/split_traintest.py:64,97 produces)."""
    scans: List[ShardScan]
    tracks: List[List[Tuple[np.ndarray, np.ndarray]]]   # [t][j] = (raw (n,4) f32, W (4,4) f64)
    nusc: bool = False

    def stacked(self, i: int) -> Tuple[np.ndarray, List[np.ndarray]]:
        """(live points, per-traversal history) of scan i in its common frame -- what the reference stacks
        (pre_compute_pp_score.py:132-150), for the oracle"""
        sc = self.scans[i]
        T = sc.n_trav
        parts = [[] for _ in range(T)]
        for (t, j), tv, rel in zip(sc.hist, sc.trav_list(), sc.rels):
            xyz = self.tracks[t][j][0][:, :3]
            if self.nusc:
                m = (xyz[:, 0] < 1.75) & (xyz[:, 0] >= -1.15) & (xyz[:, 1] < 0.65) & (xyz[:, 1] >= -0.65)
                xyz = xyz[~m]
            parts[tv].append(_transform_f32(xyz, rel))
        return (np.ascontiguousarray(_transform_f32(sc.live_raw[:, :3], sc.live_rel)),
                [np.concatenate(p).astype(np.float32) for p in parts])


def make_shard(n_scans: int, n_live: int = 30_000, n_trav: int = 10, n_frames: int = 36, n_per_frame: int | None = None,
               nusc: bool = False, frame_gap: float = 2.0, seed: int = 0, x0: float = 0.0, world_seed: int = 0,
               point_order: str = "shuffled", presence=None) -> Shard:
    """Seeds: live 7000 + 1000*seed + i, history 9000 + 100000*seed + 1000*t + j.
    presence: per track the scans [i0, i1) that have it as a traversal (presence_ramp); None: every scan has every track."""
    n_per_frame = n_live if n_per_frame is None else n_per_frame
    L = n_scans + n_frames - 1
    world = make_world(world_seed, length=max(400.0, x0 + frame_gap * L + 150.0))
    K, l2e = kitti2nu(nusc), default_l2e()
    tracks, poses = [], []
    for t in range(n_trav):
        rngp = np.random.default_rng([60_000 + seed, t])   # per traversal: a longer shard extends the tracks of a shorter one
        lat, yaw = rngp.uniform(-1.5, 1.5), rngp.uniform(-0.02, 0.02)
        tr, ps = [], []
        for j in range(L):
            pose = _pose_matrix(x0 + frame_gap * j + rngp.uniform(-0.5, 0.5), lat, yaw)
            raw = sample_frame(world, 9000 + 100_000 * seed + 1000 * t + j, n_per_frame, pose, l2e, nusc, point_order=point_order)
            tr.append((raw, pose @ l2e @ K))
            ps.append(pose)
        tracks.append(tr)
        poses.append(ps)
    scans = []
    for i in range(n_scans):
        ex = x0 + frame_gap * i
        live_pose = _pose_matrix(ex, 0.0, 0.01)
        live_raw = sample_frame(world, 7000 + 1000 * seed + i, n_live, live_pose, l2e, nusc, make_mobiles(1000 * seed + i, ex),
                                point_order=point_order)
        mine = [t for t in range(n_trav) if presence is None or presence[t][0] <= i < presence[t][1]]
        assert len(mine) >= 2, "a scan needs two traversals (pre_compute_pp_score.py:125-126)"
        first_pose = poses[mine[0]][i]
        hist, rels, travs = [], [], []
        for k, t in enumerate(mine):
            for j in range(i, i + n_frames):
                hist.append((t, j))
                travs.append(k)
                rels.append(relative_pose(l2e, first_pose, l2e, poses[t][j], K))
        scans.append(ShardScan(index=i, live_raw=live_raw, live_W=live_pose @ l2e @ K,
                               live_rel=relative_pose(l2e, first_pose, l2e, live_pose, K), hist=hist, rels=np.stack(rels),
                               world_from_ref=first_pose @ l2e @ K, first_pose=first_pose, l2e=l2e, K=K,
                               travs=None if presence is None else travs))
    return Shard(scans=scans, tracks=tracks, nusc=nusc)


DIS_CHOICE_LYFT = np.arange(2, 71, 2)            # data_preprocessing/lyft/split_traintest.py:64
DIS_CHOICE_NUSC = np.linspace(0, 30, 16)[1:]     # data_preprocessing/nuscenes/split_traintest.py:74


def match_history(origin_pose: np.ndarray, track_poses: Sequence[np.ndarray], dis_choice=DIS_CHOICE_LYFT,
                  max_allow_dist: float = 3.0):
    """The history frames one traversal contributes to one live scan, by the reference's rule
    (data_preprocessing/lyft/split_traintest.py:79-101 with only_forward, nuscenes/split_traintest.py:90-113): the closest
    pose of the track (refused beyond max_allow_dist), then for every distance threshold the FIRST frame behind it, in the
    origin's driving direction, that lies further than the threshold from the origin.  Thresholds that select the same
    frame repeat it in the list -- the reference stacks it as often.  None: the traversal does not qualify."""
    loc = np.array([p[:2, 3] for p in track_poses])
    distance = np.linalg.norm(loc - origin_pose[:2, 3], axis=1)
    k0 = int(np.argmin(distance))
    if distance[k0] > max_allow_dist:
        return None
    forward = origin_pose[0, :3] @ track_poses[k0][0, :3] > 0
    indices = [k0]
    for dis in dis_choice:
        temp = np.where(distance > dis)[0]
        side = temp[temp > k0] if forward else temp[temp < k0]
        if len(side) == 0:
            return None
        indices.append(int(side.min() if forward else side.max()))
    return indices


def make_shard_matched(n_scans: int, n_live: int = 30_000, n_trav: int = 10, n_per_frame: int | None = None, nusc: bool = False,
                       live_speed: float = 8.0, hist_speeds=(3.0, 15.0), hz: float = 5.0, seed: int = 0, x0: float = 0.0,
                       world_seed: int = 0, point_order: str = "shuffled", opposite: int = 0, presence=None) -> Shard:
    """A shard whose history windows are chosen as the reference chooses them (match_history): the live vehicle drives at
    `live_speed` m/s, traversal t at a speed drawn from `hist_speeds` (a (lo, hi) range, or one value per traversal), all
    sampled at `hz`; the last `opposite` traversals drive the other way.  Fast traversals (> 2 m per frame: > 10 m/s at 5 Hz)
    repeat frames inside a window, slow ones advance by less than a frame per live scan -- how many frames consecutive
    scans share follows from the speeds instead of being 35 of 36.  Only the frames some scan names are sampled."""
    n_per_frame = n_live if n_per_frame is None else n_per_frame
    dis = DIS_CHOICE_NUSC if nusc else DIS_CHOICE_LYFT
    reach = float(dis[-1]) + 6.0
    dt = 1.0 / hz
    span = live_speed * dt * n_scans
    world = make_world(world_seed, length=max(400.0, x0 + span + reach + 150.0))
    K, l2e = kitti2nu(nusc), default_l2e()
    rngs = np.random.default_rng([61_000 + seed, n_trav])
    speeds = (np.asarray(hist_speeds, dtype=np.float64) if len(hist_speeds) == n_trav and n_trav != 2
              else rngs.uniform(hist_speeds[0], hist_speeds[1], n_trav))
    live_poses = [_pose_matrix(x0 + live_speed * dt * i, 0.0, 0.01) for i in range(n_scans)]
    poses = []
    for t in range(n_trav):
        rngp = np.random.default_rng([62_000 + seed, t])
        lat, yaw = rngp.uniform(-1.5, 1.5), rngp.uniform(-0.02, 0.02)
        step = speeds[t] * dt
        back = t >= n_trav - opposite
        lo, hi = x0 - 5.0, x0 + span + reach + 2 * step   # (either way the window lies AHEAD of the live vehicle)
        xs = np.arange(lo, hi, step)
        xs = xs + rngp.uniform(-0.1, 0.1, len(xs)) * step
        if back:
            xs = xs[::-1]
        poses.append([_pose_matrix(x, lat, yaw + (np.pi if back else 0.0)) for x in xs])
    # (presence: per track the scans [i0, i1) that accept it as a traversal -- presence_ramp; the reference's 3 m test per scan,
    # split_traintest.py:17,79, makes traversals enter and leave along a sequence)
    has = [[presence is None or presence[t][0] <= i < presence[t][1] for t in range(n_trav)] for i in range(n_scans)]
    assert all(sum(h) >= 2 for h in has), "a scan needs two traversals (pre_compute_pp_score.py:125-126)"
    picks = [[match_history(lp, poses[t], dis) if has[i][t] else [] for t in range(n_trav)] for i, lp in enumerate(live_poses)]
    assert all(ix is not None for row in picks for ix in row), "a traversal does not cover the shard"
    used = [sorted({j for row in picks for j in row[t]}) for t in range(n_trav)]
    renum = [{j: k for k, j in enumerate(u)} for u in used]
    tracks = [[(sample_frame(world, 9000 + 100_000 * seed + 1000 * t + j, n_per_frame, poses[t][j], l2e, nusc, point_order=point_order),
                poses[t][j] @ l2e @ K) for j in used[t]] for t in range(n_trav)]
    scans = []
    for i, lp in enumerate(live_poses):
        live_raw = sample_frame(world, 7000 + 1000 * seed + i, n_live, lp, l2e, nusc, make_mobiles(1000 * seed + i, lp[0, 3]),
                                point_order=point_order)
        mine = [t for t in range(n_trav) if has[i][t]]
        first_pose = poses[mine[0]][picks[i][mine[0]][0]]
        hist = [(t, renum[t][j]) for t in mine for j in picks[i][t]]
        travs = [k for k, t in enumerate(mine) for _ in picks[i][t]]
        rels = np.stack([relative_pose(l2e, first_pose, l2e, poses[t][j], K) for t in mine for j in picks[i][t]])
        scans.append(ShardScan(index=i, live_raw=live_raw, live_W=lp @ l2e @ K, live_rel=relative_pose(l2e, first_pose, l2e, lp, K),
                               hist=hist, rels=rels, world_from_ref=first_pose @ l2e @ K, first_pose=first_pose, l2e=l2e, K=K,
                               travs=None if presence is None else travs))
    return Shard(scans=scans, tracks=tracks, nusc=nusc)


def sharing_stats(sh: Shard, block: int = 16) -> dict:
    """how much consecutive scans of a shard share: members per scan, distinct frames per scan, repeats, union of a block"""
    mem = np.array([len(sc.hist) for sc in sh.scans])
    dist = np.array([len(set(sc.hist)) for sc in sh.scans])
    unions = [len(set(h for sc in sh.scans[b:b + block] for h in sc.hist)) for b in range(0, len(sh.scans), block)]
    shared = [len(set(a.hist) & set(b.hist)) / max(len(set(b.hist)), 1) for a, b in zip(sh.scans[:-1], sh.scans[1:])]
    return dict(members_per_scan=float(mem.mean()), distinct_per_scan=float(dist.mean()), repeats_per_scan=float((mem - dist).mean()),
                union_per_block=float(np.mean(unions)), union_over_members=float(np.mean(unions) / mem.mean()),
                shared_with_previous=float(np.mean(shared)) if shared else 1.0)


CALIB_TXT = (
    "P0: 8.8e+02 0 6.12e+02 0 0 8.8e+02 5.12e+02 0 0 0 1 0\n"
    "P1: 8.8e+02 0 6.12e+02 0 0 8.8e+02 5.12e+02 0 0 0 1 0\n"
    "P2: 8.8e+02 0 6.12e+02 0 0 8.8e+02 5.12e+02 0 0 0 1 0\n"
    "P3: 8.8e+02 0 6.12e+02 0 0 8.8e+02 5.12e+02 0 0 0 1 0\n"
    "R0_rect: 1 0 0 0 1 0 0 0 1\n"
    "Tr_velo_to_cam: 0 -1 0 0 0 0 -1 -0.3 1 0 0 -0.5\n"
    "Tr_imu_to_velo: 1 0 0 0 0 1 0 0 0 0 1 0\n"
)


def write_kitti_tree(root: str, meta: str, n_seq: int = 3, n_frames: int = 12, n_pts: int = 8000,
                     nusc: bool = False, world_seed: int = 0, origins: Tuple[int, ...] = (2,),
                     hist_frames: int = 8, max_range: float = 60.0, presence=None) -> Dict[str, str]:
    """Tiny KITTI-format tree + MODEST meta data (track list, valid idx info,
    idx list) for CLI tests: sequence 0 holds the live scans (with mobile
    objects), sequences 1.. are the historical traversals.
    File formats: data_preprocessing/lyft/lyft2kitti.py:258-272,365-393.
    presence: per history sequence (index s - 1) the range [k0, k1) of POSITIONS in `origins` whose scans have it as a traversal
    (presence_ramp; the reference accepts a traversal per scan, split_traintest.py:17,79): T then changes along the idx list."""
    train = os.path.join(root, "training")
    for d in ("velodyne", "oxts", "l2e", "calib"):
        os.makedirs(os.path.join(train, d), exist_ok=True)
    os.makedirs(meta, exist_ok=True)
    # the world reaches past the end of the track (frames are 2 m apart): a scan beyond it would see no static
    # boxes, and sample_frame would then put its box points on boxes hundreds of metres away -- one clamped border
    # cell of the live grid with a million history points (found as a 20x slower pp3_join from scan ~210 on)
    world = make_world(world_seed, length=max(400.0, 2.0 * n_frames + 150.0))
    l2e = default_l2e()
    track, idx = [], 0
    rng = np.random.default_rng(40_000 + world_seed)
    for s in range(n_seq):
        seq = []
        lat = 0.0 if s == 0 else rng.uniform(-1.0, 1.0)
        for f in range(n_frames):
            ex = 2.0 * f + 0.3 * s
            yaw = 0.01 * s
            pose = _pose_matrix(ex, lat, yaw)
            mob = make_mobiles(s * 100 + f, ex, 8) if (s == 0 and f in origins) else None
            raw = sample_frame(world, 50_000 + 1000 * s + f, n_pts, pose, l2e, nusc, mob, max_range=max_range)
            raw.tofile(os.path.join(train, "velodyne", f"{idx:06d}.bin"))
            with open(os.path.join(train, "oxts", f"{idx:06d}.txt"), "w") as fh:
                fh.write(f"{ex!r} {lat!r} 0.0 0.0 0.0 {yaw!r}")
            np.save(os.path.join(train, "l2e", f"{idx:06d}.npy"), l2e)
            with open(os.path.join(train, "calib", f"{idx:06d}.txt"), "w") as fh:
                fh.write(CALIB_TXT)
            seq.append(idx)
            idx += 1
        track.append(seq)
    valid = {}
    for k, o in enumerate(origins):
        hist = [(s, list(range(o, min(o + hist_frames, n_frames)))) for s in range(1, n_seq)
                if presence is None or presence[s - 1][0] <= k < presence[s - 1][1]]
        assert len(hist) >= 2, "a scan needs two traversals (pre_compute_pp_score.py:125-126)"
        valid[track[0][o]] = (0, o, hist)
    paths = {
        "track_path": os.path.join(meta, "track_list.pkl"),
        "idx_info": os.path.join(meta, "valid_idx_info.pkl"),
        "idx_list": os.path.join(meta, "train_idx.txt"),
    }
    with open(paths["track_path"], "wb") as fh:
        pickle.dump(track, fh)
    with open(paths["idx_info"], "wb") as fh:
        pickle.dump(valid, fh)
    with open(paths["idx_list"], "w") as fh:
        fh.write("\n".join(f"{track[0][o]:06d}" for o in origins))
    return paths
