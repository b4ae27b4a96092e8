"""GPU: the three CLIs end to end on the reference-generated KITTI tree; every
output file is compared with what the reference's own mains wrote."""
import os
import pickle

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_cli_end_to_end(gpu, golden_dir, tmp_path):
    from modest_amd import config, gen_label_files, generate_mask, pre_compute_pp_score
    from tests.golden_tree import unpack_tree
    g, train, paths = unpack_tree(golden_dir, str(tmp_path))
    out = str(tmp_path / "out")
    origin = int(g["origin"])
    ov = [f"data_root={train}"] + [f"data_paths.{k}={v}" for k, v in paths.items()] + [
        f"data_paths.pp_score_path={out}/pp", f"data_paths.seg_save_dst={out}/seg",
        f"data_paths.bbox_info_save_dst={out}/bbox", f"data_paths.label_file_save_dst={out}/labels"]
    tot = pre_compute_pp_score.main(config.compose("pp_score", ov))
    assert tot["scans"] == 1
    pp = np.load(f"{out}/pp/{origin:06d}.npy")
    assert pp.dtype == np.float32 and pp.shape == g["pp"].shape
    assert np.max(np.abs(pp.astype(np.float64) - g["pp"].astype(np.float64))) <= 1e-6
    # rerun skips (the reference's skip test is broken, ours is not)
    assert pre_compute_pp_score.main(config.compose("pp_score", ov))["scans"] == 0
    # stage 2 consumes the REFERENCE's pp file so that stage parity is isolated
    np.save(f"{out}/pp/{origin:06d}.npy", g["pp"])
    generate_mask.main(config.compose("generate_mask", ov + [f"ransac_seed={int(g['seed']) - origin}"]))
    seg = np.load(f"{out}/seg/{origin:06d}.npy")
    assert seg.dtype == np.int64
    assert os.path.exists(f"{out}/seg/configs.yaml") and os.path.exists(f"{out}/bbox/configs.yaml")
    objs = pickle.load(open(f"{out}/bbox/{origin:06d}.pkl", "rb"))
    assert np.array_equal(seg, g["seg"])
    got = np.array([[*o.t, o.l, o.w, o.h, o.ry, o.volume] for o in objs]).reshape(-1, 8)
    np.testing.assert_allclose(got, g["objs"], rtol=1e-9, atol=1e-12)
    gen_label_files.main(config.compose("generate_label_files", ov))
    txt = open(f"{out}/labels/{origin:06d}.txt").read()
    assert txt == str(g["label_txt"])


def test_cli_end_to_end_nuscenes_config(gpu, golden_dir, tmp_path):
    """BASELINE config 5 in small: nusc=True (remove_center on history frames, rot-z pi/2),
    plane_estimate.max_hs=-1.3, image_shape=[900,1600]; outputs vs the reference's own mains."""
    from modest_amd import config, gen_label_files, generate_mask, pre_compute_pp_score
    from tests.golden_tree import unpack_tree
    g, train, paths = unpack_tree(golden_dir, str(tmp_path), "e2e_tree_nusc.npz")
    out = str(tmp_path / "out")
    origin = int(g["origin"])
    ov = [f"data_root={train}"] + [f"data_paths.{k}={v}" for k, v in paths.items()] + [
        f"data_paths.pp_score_path={out}/pp", f"data_paths.seg_save_dst={out}/seg",
        f"data_paths.bbox_info_save_dst={out}/bbox", f"data_paths.label_file_save_dst={out}/labels"]
    assert pre_compute_pp_score.main(config.compose("pp_score", ov + ["nusc=True"]))["scans"] == 1
    pp = np.load(f"{out}/pp/{origin:06d}.npy")
    assert pp.dtype == np.float32 and pp.shape == g["pp"].shape
    assert np.max(np.abs(pp.astype(np.float64) - g["pp"].astype(np.float64))) <= 1e-6
    np.save(f"{out}/pp/{origin:06d}.npy", g["pp"])
    generate_mask.main(config.compose("generate_mask", ov + ["plane_estimate.max_hs=-1.3",
                                                             f"ransac_seed={int(g['seed']) - origin}"]))
    assert np.array_equal(np.load(f"{out}/seg/{origin:06d}.npy"), g["seg"])
    objs = pickle.load(open(f"{out}/bbox/{origin:06d}.pkl", "rb"))
    got = np.array([[*o.t, o.l, o.w, o.h, o.ry, o.volume] for o in objs]).reshape(-1, 8)
    np.testing.assert_allclose(got, g["objs"], rtol=1e-9, atol=1e-12)
    gen_label_files.main(config.compose("generate_label_files", ov + ["image_shape=[900,1600]"]))
    assert open(f"{out}/labels/{origin:06d}.txt").read() == str(g["label_txt"])
