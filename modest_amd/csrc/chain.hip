// Stages 2 + 3 of a CHAIN of scans behind ONE entry point (round 6; VERDICT r5 item 4, first half).
//
// Reference steps: generate_mask.py:52-103 (both ground fits, masks, graph + DBSCAN, filter_labels, rect projection, get_obj per
// cluster, volume gate, relabelling) and the device part of gen_label_files.py:44-45 (the K x K BEV IoU of objs_nms,
// pointcloud_utils.py:320-327) for every scan of the chain.  modest_mask_stage_batch, modest_scan_boxes_batch and
// modest_objs_iou_batch already did each step for a chain; the interpreter between them (allocating per-scan arrays, re-packing
// their results, three ctypes calls) cost a chain of 16 scans 1.5 of its 5.3 ms (profiles/r06_chain_profile_before.txt).  Here the
// three calls are one: final labels, the KEPT box rows and their IoU matrix leave the library once per scan.  What stays with the
// caller is what must be numpy's: np.diag(iou).argsort() (its order of equal keys is numpy's own) and the cos / sin of
// every heading for the label text -- then modest_label_lines.
#include "common.h"
#include <vector>

extern "C" int modest_seed_chain(const modest_seed_scan *scans, int n_scans, const modest_mask_params *mp, const modest_boxes_params *bp,
                                 int max_boxes, int nms_enable, void *stream) {
    MODEST_REQUIRE(scans != nullptr && mp != nullptr && bp != nullptr && n_scans >= 1 && n_scans <= 64 && max_boxes >= 1, "bad chain");
    std::vector<modest_mask_stage_scan> ms((size_t)n_scans);
    std::vector<int32_t> n_mem((size_t)n_scans, -1);
    for (int s = 0; s < n_scans; ++s) {
        const modest_seed_scan &q = scans[s];
        MODEST_REQUIRE(q.ctx && q.pts_dev && q.pts_host && q.pp_dev && q.mt_key624 && q.mt_pos && q.plane1_out && q.plane2_out && q.labels_out
                           && q.members_scratch && q.objs_out && q.info_out && (q.iou_out || !nms_enable),
                       "NULL argument");
        for (int k = 0; k < 12; ++k) q.info_out[k] = 0;
        modest_mask_stage_scan &m = ms[(size_t)s];
        m.ctx = q.ctx, m.pts_dev = q.pts_dev, m.n = q.n, m.stride = q.stride, m.pp_dev = q.pp_dev;
        m.mt_key624 = q.mt_key624, m.mt_pos = q.mt_pos, m.plane1_out = q.plane1_out, m.plane2_out = q.plane2_out;
        m.labels_out = q.labels_out, m.info_out = q.info_out;   // (info_out[0..7]: the mask stage's)
        m.members_out = q.members_scratch, m.n_members_out = &n_mem[(size_t)s];
    }
    int rc = modest_mask_stage_batch(ms.data(), n_scans, mp, stream);
    if (rc) return rc;
    // the box tail of the scans the stage finished (status 0) whose boxes fit the caller's buffers
    std::vector<modest_boxes_scan> bs;
    std::vector<int> who;
    std::vector<std::vector<double>> rows((size_t)n_scans);
    std::vector<std::vector<int32_t>> keep((size_t)n_scans);
    std::vector<int32_t> binfo((size_t)2 * n_scans, 0);
    for (int s = 0; s < n_scans; ++s) {
        const modest_seed_scan &q = scans[s];
        if (q.info_out[3] != 0) {   // handed back by the mask stage (generator untouched): the caller's host statement takes the scan
            q.info_out[10] = 1;
            continue;
        }
        const int n_lab = q.info_out[2];
        q.info_out[8] = n_lab;
        if (n_lab > max_boxes) {   // (the stage's labels are in labels_out; the tail is the caller's: modest_scan_boxes)
            q.info_out[10] = 3;
            continue;
        }
        rows[(size_t)s].assign((size_t)std::max(n_lab, 1) * 8, 0.0);
        keep[(size_t)s].assign((size_t)std::max(n_lab, 1), 0);
        modest_boxes_scan b;
        b.ctx = q.ctx, b.pts_dev = q.pts_dev, b.pts_host = q.pts_host, b.n = q.n, b.stride = q.stride;
        b.labels_inout = q.labels_out, b.n_lab = n_lab;
        b.objs_out = rows[(size_t)s].data(), b.keep_out = keep[(size_t)s].data(), b.info_out = &binfo[(size_t)2 * s];
        b.members = n_mem[(size_t)s] >= 0 ? q.members_scratch : nullptr, b.n_members = n_mem[(size_t)s] >= 0 ? n_mem[(size_t)s] : 0;
        bs.push_back(b);
        who.push_back(s);
    }
    if (bs.empty()) return MODEST_OK;
    // (labels_inout is overwritten only by a scan whose tail succeeds; a tail that hands back leaves labels_filtered in place)
    rc = modest_scan_boxes_batch(bs.data(), (int)bs.size(), bp, stream);
    if (rc) return rc;
    std::vector<const double *> ip;
    std::vector<float *> op;
    std::vector<int32_t> kk;
    modest_ctx *c0 = nullptr;
    for (int s : who) {
        const modest_seed_scan &q = scans[s];
        if (binfo[(size_t)2 * s + 1] != 0) {   // a cluster too large for the extents kernel / an empty footprint: host statement
            q.info_out[10] = 2;
            continue;
        }
        const int n_lab = q.info_out[8];
        int k = 0;
        for (int c = 0; c < n_lab; ++c)
            if (keep[(size_t)s][(size_t)c]) {
                for (int j = 0; j < 8; ++j) q.objs_out[(size_t)k * 8 + j] = rows[(size_t)s][(size_t)c * 8 + j];
                ++k;
            }
        q.info_out[9] = k;
        if (nms_enable && k > 0) {
            ip.push_back(q.objs_out);
            op.push_back(q.iou_out);
            kk.push_back(k);
            if (!c0) c0 = q.ctx;
        }
    }
    if (!ip.empty()) {
        rc = modest_objs_iou_batch(c0, ip.data(), kk.data(), (int)ip.size(), op.data(), stream);
        if (rc) return rc;
    }
    return MODEST_OK;
}
