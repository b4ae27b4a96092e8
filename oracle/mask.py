"""ORACLE — test infrastructure only, never imported by the product path.

CPU restatement of MODEST's mask/cluster stage: ``generate_cluster_mask/
generate_mask.py`` with ``utils/pointcloud_utils.py`` and
``utils/clustering_utils.py`` of the reference checkout (default config
branches: radius_mutual_knn / l1 / DBSCAN / closeness_to_edge).  Only
``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg
may import this module.

Third-party arithmetic the reference delegates to (not under /root/reference,
unpinned by it; this image: scikit-learn 1.7.2, numpy 2.2.6):
``sklearn.linear_model.RANSACRegressor``, ``sklearn.neighbors.kneighbors_graph``
/ ``radius_neighbors_graph``, ``sklearn.cluster.DBSCAN``.  The same calls are
made here.  Pinned by tests/golden/mask_stage.npz and e2e_tree.npz, generated
by importing the reference (tools/make_goldens.py).
"""
from __future__ import annotations

import types

import numpy as np
import scipy.sparse
import sklearn.neighbors as neighbors
from sklearn import cluster
from sklearn.linear_model import RANSACRegressor


# ----------------------------------------------------------------------------- plane
def distance_to_plane(ptc, plane, directional=False):
    """utils/pointcloud_utils.py:76-81."""
    d = ptc @ plane[:3] + plane[3]
    if not directional:
        d = np.abs(d)
    d /= np.sqrt((plane[:3] ** 2).sum())
    return d


def above_plane(ptc, plane, offset=0.05, only_range=((-30, 30), (-30, 30))):
    """utils/pointcloud_utils.py:68-74."""
    mask = distance_to_plane(ptc, plane, directional=True) < offset
    if only_range is not None:
        range_mask = (ptc[:, 0] < only_range[0][1]) * (ptc[:, 0] > only_range[0][0]) * \
            (ptc[:, 1] < only_range[1][1]) * (ptc[:, 1] > only_range[1][0])
        mask *= range_mask
    return np.logical_not(mask)


def plane_candidate_mask(origin_ptc, max_hs, ptc_range):
    """utils/pointcloud_utils.py:45-49."""
    return (origin_ptc[:, 2] < max_hs) & \
        (origin_ptc[:, 0] > ptc_range[0][0]) & (origin_ptc[:, 0] < ptc_range[0][1]) & \
        (origin_ptc[:, 1] > ptc_range[1][0]) & (origin_ptc[:, 1] < ptc_range[1][1])


def plane_from_linear_model(coef, intercept):
    """utils/pointcloud_utils.py:53-62 (w = [c0, c1, -1]/|w|, h/|w|, negated)."""
    w = np.zeros(3)
    w[0] = coef[0]
    w[1] = coef[1]
    w[2] = -1.0
    h = intercept
    norm = np.linalg.norm(w)
    w /= norm
    h = h / norm
    result = np.array((w[0], w[1], w[2], h))
    result *= -1
    return result


def estimate_plane(origin_ptc, max_hs=-1.5, ptc_range=((-20, 70), (-20, 20)), random_state=None,
                   return_reg=False):
    """utils/pointcloud_utils.py:44-65 with it=1 (the trailing above_plane of the
    loop body is dead work and is skipped).  ``random_state`` replaces the
    reference's use of numpy's global RNG (RANSACRegressor() with
    random_state=None draws from np.random.mtrand._rand)."""
    mask = plane_candidate_mask(origin_ptc, max_hs, ptc_range)
    ptc = origin_ptc[mask]
    reg = RANSACRegressor(random_state=random_state).fit(ptc[:, [0, 1]], ptc[:, 2])
    result = plane_from_linear_model(reg.estimator_.coef_, reg.estimator_.intercept_)
    return (result, reg, ptc) if return_reg else result


# ----------------------------------------------------------------------------- graph + DBSCAN
def precompute_affinity_matrix(ptc, pp_score, n_neighbors=70, radius=2.0, n_jobs=-1,
                               neighbor_type="radius_mutual_knn", affinity_type="l1"):
    """utils/clustering_utils.py:7-60; defaults = configs/generate_mask.yaml:20-25
    (neighbor_type='radius_mutual_knn', affinity_type='l1').  All branches of the reference are
    restated (the product builds radius_mutual_knn and radius with the three affinities)."""
    assert ptc.shape[0] == pp_score.shape[0]
    if neighbor_type == "knn":                                                      # :16-18
        graph = neighbors.kneighbors_graph(ptc[:, :3], n_neighbors=n_neighbors, n_jobs=n_jobs)
    elif neighbor_type == "sym_knn":                                                # :19-23
        graph = neighbors.kneighbors_graph(ptc[:, :3], n_neighbors=n_neighbors, n_jobs=n_jobs)
        graph = graph + graph.T
        graph.eliminate_zeros()
    elif neighbor_type == "mutual_knn":                                             # :24-28
        graph = neighbors.kneighbors_graph(ptc[:, :3], n_neighbors=n_neighbors, n_jobs=n_jobs)
        graph = graph.multiply(graph.T)
        graph.eliminate_zeros()
    elif neighbor_type == "radius":                                                 # :29-31
        graph = neighbors.radius_neighbors_graph(ptc[:, :3], radius=radius, n_jobs=n_jobs)
    elif neighbor_type == "radius_mutual_knn":                                      # :32-38
        graph = neighbors.kneighbors_graph(ptc[:, :3], n_neighbors=n_neighbors, n_jobs=n_jobs)
        graph = graph.multiply(graph.T)
        graph = graph.multiply(neighbors.radius_neighbors_graph(ptc[:, :3], radius=radius, n_jobs=n_jobs))
        graph.eliminate_zeros()
    else:
        raise NotImplementedError(neighbor_type)
    graph = scipy.sparse.csr_matrix(graph)
    dist_data = graph.data.copy()
    # the reference fills row by row (:43-56); the vectorised forms below perform the same
    # float32 operations per stored entry and the same cast to float64
    rows = np.repeat(np.arange(graph.shape[0]), np.diff(graph.indptr))
    if affinity_type == "l1":
        dist_data[:] = np.abs(pp_score[rows] - pp_score[graph.indices])
    elif affinity_type == "exp":
        dist_data[:] = np.exp((pp_score[rows] - pp_score[graph.indices]) ** 2)
    elif affinity_type == "3d_l2_distance":
        # np.linalg.norm(ptc[_r].reshape(1, -1) - ptc[idx], axis=1): ALL columns of the rows passed in
        dist_data[:] = np.linalg.norm(ptc[rows] - ptc[graph.indices], axis=1)
    else:
        raise NotImplementedError(affinity_type)
    return scipy.sparse.csr_matrix((dist_data, graph.indices, graph.indptr), shape=graph.shape)


def dbscan_labels(graph, eps=0.1, min_samples=10, n_jobs=-1):
    """generate_mask.py:77-81."""
    return cluster.DBSCAN(metric="precomputed", eps=eps, min_samples=min_samples, n_jobs=n_jobs).fit(graph).labels_


def dbscan_closed_form(xyz, pp, n_neighbors=70, radius=2.0, eps=0.1, min_samples=10):
    """O(n^2) numpy statement of the implicit-graph definition the HIP kernel
    implements (see modest_amd/csrc/cluster.hip); checked against the two
    sklearn calls above in tests/test_oracle_mask.py.  Small n only."""
    p = xyz[:, :3].astype(np.float64)
    n = len(p)
    d = p[:, None, :] - p[None, :, :]
    d2 = d[..., 0] * d[..., 0] + d[..., 1] * d[..., 1]
    d2 = d2 + d[..., 2] * d[..., 2]
    np.fill_diagonal(d2, np.inf)
    r2 = radius * radius
    inr = np.where(d2 <= r2, d2, np.inf)
    srt = np.sort(inr, axis=1)
    kth = srt[:, n_neighbors - 1] if n > n_neighbors else np.full(n, np.inf)
    lim = np.minimum(np.minimum(kth[:, None], kth[None, :]), r2)
    w = np.abs(pp[:, None] - pp[None, :]).astype(np.float64)      # float32 |.|, then widened
    edge = (d2 <= lim) & (w <= eps)
    core = edge.sum(1) + 1 >= min_samples
    parent = np.arange(n)

    def find(x):
        while parent[x] != x:
            parent[x] = parent[parent[x]]
            x = parent[x]
        return x

    ii, jj = np.nonzero(edge & core[:, None] & core[None, :])
    for a, b in zip(ii, jj):
        ra, rb = find(a), find(b)
        if ra != rb:
            parent[max(ra, rb)] = min(ra, rb)
    roots = np.array([find(i) for i in range(n)])
    ids = {r: k for k, r in enumerate(sorted(set(roots[core])))}
    labels = np.full(n, -1, dtype=np.int64)
    for i in range(n):
        if core[i]:
            labels[i] = ids[roots[i]]
        else:
            adj = np.nonzero(edge[i] & core)[0]
            if len(adj):
                labels[i] = min(ids[roots[j]] for j in adj)
    return labels, kth


# ----------------------------------------------------------------------------- filtering
def is_valid_cluster(ptc, pp_score, plane, min_points=10, max_volume=40, min_volume=0.5, max_min_height=4,
                     min_max_height=0, percentile=10, min_percentile_pp_score=0.7):
    """utils/clustering_utils.py:94-117."""
    if ptc.shape[0] < min_points:
        return False
    distance_to_ground = distance_to_plane(ptc, plane, directional=True)
    if distance_to_ground.min() > max_min_height:
        return False
    if distance_to_ground.max() < min_max_height:
        return False
    if np.percentile(pp_score, percentile) > min_percentile_pp_score:
        return False
    return True


def filter_labels(ptc, pp_score, labels, random_state=None, plane=None, **kwargs):
    """utils/clustering_utils.py:119-135.  ``plane`` may be injected (stage-wise
    parity tests); otherwise the second, hard-coded estimate_plane call runs."""
    labels = labels.copy()
    if plane is None:
        plane = estimate_plane(ptc, max_hs=-1.5, ptc_range=((-70, 70), (-50, 50)), random_state=random_state)
    for i in range(labels.max() + 1):
        if not is_valid_cluster(ptc[labels == i, :3], pp_score[labels == i], plane, **kwargs):
            labels[labels == i] = -1
    mapping = {x: i for i, x in enumerate(sorted(set(labels.tolist())))}
    return np.array([mapping[v] for v in labels.tolist()], dtype=labels.dtype)


# ----------------------------------------------------------------------------- box fit
def closeness_rectangle(cluster_ptc, delta=0.1, d0=1e-2, return_index=False):
    """utils/pointcloud_utils.py:167-216."""
    best_beta, best_angle, best_idx = -float("inf"), None, -1
    for k, deg in enumerate(np.arange(0, 90 + delta, delta)):
        angle = deg / 180. * np.pi
        comp = np.array([[np.cos(angle), np.sin(angle)], [-np.sin(angle), np.cos(angle)]])
        proj = cluster_ptc @ comp.T
        min_x, max_x = proj[:, 0].min(), proj[:, 0].max()
        min_y, max_y = proj[:, 1].min(), proj[:, 1].max()
        Dx = np.vstack((proj[:, 0] - min_x, max_x - proj[:, 0])).min(axis=0)
        Dy = np.vstack((proj[:, 1] - min_y, max_y - proj[:, 1])).min(axis=0)
        beta = np.vstack((Dx, Dy)).min(axis=0)
        beta = np.maximum(beta, d0)
        beta = (1 / beta).sum()
        if beta > best_beta:
            best_beta, best_angle, best_idx = beta, angle, k
    out = rectangle_at_angle(cluster_ptc, best_angle)
    return out + (best_idx,) if return_index else out


def rectangle_at_angle(cluster_ptc, choose_angle):
    """utils/pointcloud_utils.py:188-216: tight rectangle at the chosen angle,
    rotated by 90 degrees if needed so that the first side is the long one."""
    angle = choose_angle
    comp = np.array([[np.cos(angle), np.sin(angle)], [-np.sin(angle), np.cos(angle)]])
    proj = cluster_ptc @ comp.T
    min_x, max_x = proj[:, 0].min(), proj[:, 0].max()
    min_y, max_y = proj[:, 1].min(), proj[:, 1].max()
    if (max_x - min_x) < (max_y - min_y):
        angle = choose_angle + np.pi / 2
        comp = np.array([[np.cos(angle), np.sin(angle)], [-np.sin(angle), np.cos(angle)]])
        proj = cluster_ptc @ comp.T
        min_x, max_x = proj[:, 0].min(), proj[:, 0].max()
        min_y, max_y = proj[:, 1].min(), proj[:, 1].max()
    area = (max_x - min_x) * (max_y - min_y)
    rval = np.array([[max_x, min_y], [min_x, min_y], [min_x, max_y], [max_x, max_y]])
    rval = rval @ comp
    return rval, angle, area


def variance_rectangle(cluster_ptc, delta=0.1, return_index=False):
    """utils/pointcloud_utils.py:218-275 (fit_method='variance_to_edge')."""
    max_var, choose_angle, best_idx = -float("inf"), None, -1
    for k, deg in enumerate(np.arange(0, 90 + delta, delta)):
        angle = deg / 180. * np.pi
        comp = np.array([[np.cos(angle), np.sin(angle)], [-np.sin(angle), np.cos(angle)]])
        proj = cluster_ptc @ comp.T
        min_x, max_x = proj[:, 0].min(), proj[:, 0].max()
        min_y, max_y = proj[:, 1].min(), proj[:, 1].max()
        Dx = np.vstack((proj[:, 0] - min_x, max_x - proj[:, 0])).min(axis=0)
        Dy = np.vstack((proj[:, 1] - min_y, max_y - proj[:, 1])).min(axis=0)
        Ex, Ey = Dx[Dx < Dy], Dy[Dy < Dx]
        var = 0
        if (Dx < Dy).sum() > 0:
            var += -np.var(Ex)
        if (Dy < Dx).sum() > 0:
            var += -np.var(Ey)
        if var > max_var:
            max_var, choose_angle, best_idx = var, angle, k
    out = rectangle_at_angle(cluster_ptc, choose_angle)   # :253-275 is the same tail as :188-216
    return out + (best_idx,) if return_index else out


def PCA_rectangle(cluster_ptc):
    """utils/pointcloud_utils.py:189-206 (fit_method='PCA')."""
    import sklearn.decomposition
    components = sklearn.decomposition.PCA(n_components=2).fit(cluster_ptc).components_
    on_component_ptc = cluster_ptc @ components.T
    min_x, max_x = on_component_ptc[:, 0].min(), on_component_ptc[:, 0].max()
    min_y, max_y = on_component_ptc[:, 1].min(), on_component_ptc[:, 1].max()
    area = (max_x - min_x) * (max_y - min_y)
    rval = np.array([[max_x, min_y], [min_x, min_y], [min_x, max_y], [max_x, max_y]])
    rval = rval @ components
    angle = np.arctan2(components[0, 1], components[0, 0])
    return rval, angle, area


def minimum_bounding_rectangle(points):
    """utils/pointcloud_utils.py:88-147 (fit_method='min_zx_area_fit'): rotating calipers over the
    edges hull[1:] - hull[:-1] of scipy's (Qhull's) vertex order -- the closing edge is not tried."""
    from scipy.spatial import ConvexHull
    pi2 = np.pi / 2.
    hull_points = points[ConvexHull(points).vertices]
    edges = hull_points[1:] - hull_points[:-1]
    angles = np.arctan2(edges[:, 1], edges[:, 0])
    angles = np.abs(np.mod(angles, pi2))
    angles = np.unique(angles)
    rotations = np.vstack([np.cos(angles), np.cos(angles - pi2), np.cos(angles + pi2), np.cos(angles)]).T
    rotations = rotations.reshape((-1, 2, 2))
    rot_points = np.dot(rotations, hull_points.T)
    min_x, max_x = np.nanmin(rot_points[:, 0], axis=1), np.nanmax(rot_points[:, 0], axis=1)
    min_y, max_y = np.nanmin(rot_points[:, 1], axis=1), np.nanmax(rot_points[:, 1], axis=1)
    areas = (max_x - min_x) * (max_y - min_y)
    best_idx = np.argmin(areas)
    x1, x2, y1, y2 = max_x[best_idx], min_x[best_idx], max_y[best_idx], min_y[best_idx]
    r = rotations[best_idx]
    rval = np.zeros((4, 2))
    rval[0] = np.dot([x1, y2], r)
    rval[1] = np.dot([x2, y2], r)
    rval[2] = np.dot([x2, y1], r)
    rval[3] = np.dot([x1, y1], r)
    return rval, angles[best_idx], areas[best_idx]


FIT_METHODS = {"closeness_to_edge": None, "variance_to_edge": None, "PCA": None, "min_zx_area_fit": None}


def get_lowest_point_rect(ptc, xz_center, l, w, ry):
    """utils/pointcloud_utils.py:278-290."""
    ptc_xz = ptc[:, [0, 2]] - xz_center
    rot = np.array([[np.cos(ry), -np.sin(ry)], [np.sin(ry), np.cos(ry)]])
    ptc_xz = ptc_xz @ rot.T
    mask = (ptc_xz[:, 0] > -l / 2) & (ptc_xz[:, 0] < l / 2) & (ptc_xz[:, 1] > -w / 2) & (ptc_xz[:, 1] < w / 2)
    return ptc[mask, 1].max()


def get_obj(ptc, full_ptc, fit_method="closeness_to_edge"):
    """utils/pointcloud_utils.py:292-317 (configs/generate_mask.yaml uses 'closeness_to_edge')."""
    fit = {"closeness_to_edge": closeness_rectangle, "variance_to_edge": variance_rectangle,
           "PCA": PCA_rectangle, "min_zx_area_fit": minimum_bounding_rectangle}[fit_method]
    corners, ry, area = fit(ptc[:, [0, 2]])
    ry *= -1
    l = np.linalg.norm(corners[0] - corners[1])
    w = np.linalg.norm(corners[0] - corners[-1])
    c = (corners[0] + corners[2]) / 2
    bottom = get_lowest_point_rect(full_ptc, c, l, w, ry)
    h = bottom - ptc[:, 1].min()
    obj = types.SimpleNamespace()
    obj.t = np.array([c[0], bottom, c[1]])
    obj.l, obj.w, obj.h, obj.ry = l, w, h, ry
    obj.volume = area * h
    return obj


# ----------------------------------------------------------------------------- whole stage
DEFAULT_CFG = dict(
    plane_estimate=dict(range=[[-70, 70], [-20, 20]], max_hs=-1.5, offset=0.05),
    limit_range=[[-70, 70], [-40, 40]],
    graph=dict(n_neighbors=70, radius=2.0),
    DBSCAN=dict(eps=0.1, min_samples=10),
    filtering=dict(min_points=10, max_volume=120, min_volume=0.5, min_max_height=0.5, max_min_height=1.0,
                   percentile=20, min_percentile_pp_score=0.7),
)


def generate_mask_scan(ptc, pp_score, calib, cfg=None, random_state=None, planes=None, n_jobs=-1):
    """generate_mask.py:52-103 for one scan.  ``random_state``: one RandomState
    consumed by both RANSAC calls in order (the reference consumes the global
    stream in that order).  ``planes``=(plane1, plane2) injects both planes."""
    cfg = cfg or DEFAULT_CFG
    pe = cfg["plane_estimate"]
    plane = planes[0] if planes is not None else estimate_plane(
        ptc[:, :3], max_hs=pe["max_hs"], ptc_range=pe["range"], random_state=random_state)
    plane_mask = above_plane(ptc[:, :3], plane, offset=pe["offset"], only_range=pe["range"])
    lr = cfg["limit_range"]
    range_mask = (ptc[:, 0] <= lr[0][1]) * (ptc[:, 0] > lr[0][0]) * (ptc[:, 1] <= lr[1][1]) * (ptc[:, 1] > lr[1][0])
    final_mask = plane_mask * range_mask
    graph = precompute_affinity_matrix(ptc[final_mask], pp_score[final_mask], n_neighbors=cfg["graph"]["n_neighbors"],
                                       radius=cfg["graph"]["radius"], n_jobs=n_jobs)
    labels = np.zeros(ptc.shape[0], dtype=int) - 1
    labels[final_mask] = dbscan_labels(graph, cfg["DBSCAN"]["eps"], cfg["DBSCAN"]["min_samples"], n_jobs=n_jobs)
    labels_filtered = filter_labels(ptc, pp_score, labels, random_state=random_state,
                                    plane=None if planes is None else planes[1], **cfg["filtering"])
    ptc_in_rect = calib.project_velo_to_rect(ptc[:, :3])
    objs = []
    for i in range(1, labels_filtered.max() + 1):
        obj = get_obj(ptc_in_rect[labels_filtered == i], ptc_in_rect)
        if obj.volume > cfg["filtering"]["min_volume"] and obj.volume < cfg["filtering"]["max_volume"]:
            objs.append(obj)
        else:
            labels_filtered[labels_filtered == i] = 0
    mapping = {x: i for i, x in enumerate(sorted(set(labels_filtered.tolist())))}
    labels_filtered = np.array([mapping[v] for v in labels_filtered.tolist()], dtype=labels_filtered.dtype)
    return dict(plane=plane, plane_mask=plane_mask, final_mask=final_mask, dbscan=labels[final_mask],
                labels=labels_filtered, objs=objs)
