"""CPU: the PP-score oracle against the fixtures generated from the reference."""
import numpy as np

from oracle import pp_score as opp


def _load(golden_dir, name):
    g = np.load(f"{golden_dir}/{name}.npz")
    off = g["offsets"]
    hist = [g["hist"][off[i]:off[i + 1]] for i in range(len(off) - 1)]
    return g, hist


def test_counts_match_reference(golden_dir):
    for name in ("pp_lyft", "pp_nusc"):
        g, hist = _load(golden_dir, name)
        c = opp.count_neighbors(g["live"], hist, 0.3)
        assert c.dtype == np.int64 and np.array_equal(c, g["count"])
        H = opp.compute_ephe_score(c)
        assert np.array_equal(H.astype(np.float32), g["H32"])
        assert np.max(np.abs(H - g["H"])) <= 1e-12


def test_bruteforce_definition_equals_kdtree(golden_dir):
    g, hist = _load(golden_dir, "pp_lyft")
    live = g["live"][:600]
    assert np.array_equal(opp.count_neighbors_bruteforce(live, hist, 0.3), g["count"][:600])


def test_entropy_edge_cases():
    c = np.array([[0, 0, 0], [5, 5, 5], [9, 0, 0], [1, 2, 3]], dtype=np.int64)
    H = opp.compute_ephe_score(c)
    assert H[0] == 0.0
    assert abs(H[1] - 1.0) < 1e-6
    assert abs(H[2]) < 1e-6


def test_pose_and_transform(golden_dir):
    g = np.load(f"{golden_dir}/pose.npz")
    assert np.array_equal(opp.kitti2nu(False), g["K_lyft"])
    assert np.array_equal(opp.kitti2nu(True), g["K_nusc"])
    for i in range(len(g["fixed_ego"])):
        a = opp.get_relative_pose(g["fixed_l2e"][i], g["fixed_ego"][i], g["query_l2e"][i], g["query_ego"][i],
                                  opp.kitti2nu(False))
        b = opp.get_relative_pose(g["fixed_l2e"][i], g["fixed_ego"][i], g["query_l2e"][i], g["query_ego"][i],
                                  opp.kitti2nu(True))
        assert a.dtype == np.float32
        np.testing.assert_allclose(a, g["rel_lyft"][i], rtol=0, atol=1e-6)
        np.testing.assert_allclose(b, g["rel_nusc"][i], rtol=0, atol=1e-6)
    t = np.load(f"{golden_dir}/transform.npz")
    out = opp.transform_points(t["pts"], t["T"])
    np.testing.assert_allclose(out, t["out"], rtol=0, atol=2e-5)
    # the explicit FMA-chain statement is what the HIP kernel implements; it is
    # bit-identical to the BLAS product the reference ran in the build container
    assert np.array_equal(opp.transform_points_fma(t["pts"], t["T"]), t["out"])
    assert np.array_equal(opp.remove_center(t["pts"]), t["kept"])


def test_kitti2nu_against_independent_rotation():
    """The K matrices of tests/golden/pose.npz were written by the oracle's own restatement of pyquaternion
    (absent from the image): checked here against two independent statements of the same rotations --
    scipy's Rotation and the closed form of a rotation about z -- and against what pyquaternion's
    algorithm guarantees (orthonormal, det +1, homogeneous row)."""
    from scipy.spatial.transform import Rotation
    for nusc, angle in ((False, np.pi), (True, np.pi / 2)):
        K = opp.kitti2nu(nusc)
        ref = np.eye(4)
        ref[:3, :3] = Rotation.from_euler("z", angle).as_matrix()
        assert np.max(np.abs(K - ref)) <= 2.5e-16          # both are within an ulp of cos/sin of the angle
        c, s = np.cos(angle), np.sin(angle)
        closed = np.array([[c, -s, 0, 0], [s, c, 0, 0], [0, 0, 1, 0], [0, 0, 0, 1]])
        assert np.max(np.abs(K - closed)) <= 2.5e-16
        assert np.max(np.abs(K[:3, :3] @ K[:3, :3].T - np.eye(3))) <= 4e-16 and abs(np.linalg.det(K[:3, :3]) - 1) <= 4e-16
        assert np.array_equal(K[3], [0, 0, 0, 1]) and np.array_equal(K[:3, 3], [0, 0, 0])
        # what the rotation is FOR (pre_compute_pp_score.py:22-24): KITTI velodyne axes -> nuScenes lidar axes
        fwd = K[:3, :3] @ np.array([1.0, 0.0, 0.0])
        assert np.allclose(fwd, [-1, 0, 0] if not nusc else [0, 1, 0], atol=1e-15)
        # the float32 relative pose (the only thing that leaves get_relative_pose) does not see the last ulp
        rng = np.random.default_rng(5)
        E = np.eye(4)
        E[:3, :3] = Rotation.from_euler("xyz", rng.uniform(-0.1, 0.1, 3)).as_matrix()
        E[:3, 3] = rng.uniform(-50, 50, 3)
        assert np.array_equal(opp.get_relative_pose(np.eye(4), np.eye(4), np.eye(4), E, K),
                              opp.get_relative_pose(np.eye(4), np.eye(4), np.eye(4), E, ref))


def _repeat_tree(golden_dir):
    import pickle
    g = np.load(f"{golden_dir}/pp_repeats.npz")
    off = g["bin_offsets"]
    bins = [g["bins"][off[i]:off[i + 1]] for i in range(len(off) - 1)]
    track, valid = pickle.loads(g["track"].tobytes()), pickle.loads(g["valid"].tobytes())
    poses = [[opp.load_pose(str(g["oxts"][i])) for i in seq] for seq in track]
    l2es = [[g["l2e"][i] for i in seq] for seq in track]
    return g, bins, track, valid, poses, l2es


def test_repeated_history_frames_are_stacked_as_often_as_listed(golden_dir):
    """tests/golden/pp_repeats.npz (tools/make_golden_pp_repeats.py: the reference's own main + count_neighbors on a tree whose
    index lists repeat a frame once, three times, and at the first and last position): the oracle's stacking follows the
    list, and dropping the repeats changes the counts."""
    g, bins, track, valid, poses, l2es = _repeat_tree(golden_dir)
    changed = 0
    for o in g["origins"]:
        seq0, fr0, trav = valid[int(o)]
        assert any(len(set(ix)) < len(ix) for _, ix in trav)
        stacks, fp, fl = opp.stack_history(trav, track, lambda i: bins[i], poses, l2es)
        assert [len(s) for s in stacks] == list(g[f"stack_sizes_{o}"])
        rel = opp.get_relative_pose(fl, fp, l2es[seq0][fr0], poses[seq0][fr0], opp.kitti2nu(False))
        live = opp.transform_points(bins[track[seq0][fr0]][:, :3], rel)
        H, c = opp.pp_score(live, stacks, 0.3)
        assert np.array_equal(c, g[f"counts_{o}"]) and np.array_equal(H, g[f"pp_{o}"])
        dedup = [(s, list(dict.fromkeys(ix))) for s, ix in trav]
        stacks1, _, _ = opp.stack_history(dedup, track, lambda i: bins[i], poses, l2es)
        changed += int(not np.array_equal(opp.count_neighbors(live, stacks1, 0.3), c))
    assert changed == len(g["origins"])   # a set-semantics implementation fails this fixture on every scan
