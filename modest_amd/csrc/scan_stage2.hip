// Second half of a scan behind the C ABI (round 3): the box tail of generate_mask.py:88-103 and the label
// stage of gen_label_files.py:44-52, which round 2 still strung together in Python (1.2 ms of interpreter
// per scan against 0.3 ms of device work):
//
//   modest_scan_boxes   members of every cluster, their rect-frame points (Calibration.project_velo_to_rect,
//                       utils/kitti_util.py:327-329), the 901-angle closeness fit (device), rectangle_at_angle's
//                       scalar tail (pointcloud_utils.py:188-216), get_obj (:292-317: l, w, centre, lowest point
//                       (device), h, volume), the volume gate and the relabelling (generate_mask.py:91-103).
//   modest_objs_iou     the float32 boxes of objs_nms (pointcloud_utils.py:322-324) and their BEV IoU matrix (device).
//   modest_label_lines  objs_nms' greedy walk (:329-343) in a caller-supplied order, is_within_fov (:373-379),
//                       objs2label (:347-370; compute_box_3d / project_to_image, kitti_util.py:430-488).
//
// Host code only: every device step is one of the library's own entry points.  What is NOT here, on purpose:
// the ORDER of the NMS walk.  The reference ranks boxes by `np.diag(iou).argsort()[::-1]` -- float32 self-IoUs
// that differ by rounding noise or not at all -- and numpy's argsort of ties is its SIMD sort's
// business; the caller makes that one numpy call between modest_objs_iou and modest_label_lines.
//
// Rounding.  The per-cluster numpy arithmetic is restated operation by operation: 2-D products as the
// fma chains OpenBLAS' dgemm / ddot run (first product rounded, then one fma per further term), cos / sin
// only ever taken from the caller's tables (numpy evaluates the 901 headings, heading + pi/2 too; ry = -heading
// uses cos(-a) = cos a, sin(-a) = -sin a), np.linalg.norm as sqrt(ddot).  tests/test_gpu_e2e.py compares every
// output with the Python statement (exact on the test scans; the contract of the existing tests is 1e-9 on
// box values, identical decisions, byte-identical label text).
#include "common.h"
#include "mask_chain.h"
#include <algorithm>
#include <cmath>
#include <ctime>
#include <cstdio>
#include <cstdlib>
#include <cstdio>
#include <cstring>
#include <vector>

namespace {

inline double dot2(double a0, double b0, double a1, double b1) { return std::fma(a1, b1, a0 * b0); }
inline double dot3(double a0, double b0, double a1, double b1, double a2, double b2) {
    return std::fma(a2, b2, std::fma(a1, b1, a0 * b0));
}
inline double dot4(const double *a, const double *b) {
    return std::fma(a[3], b[3], std::fma(a[2], b[2], std::fma(a[1], b[1], a[0] * b[0])));
}

// Calibration.project_velo_to_rect of one point: the device kernel's chain (transform.hip), i.e. numpy's two dgemm calls
inline void velo_to_rect(const float *p, const double *V2C, const double *R0, double *out) {
    const double x = p[0], y = p[1], z = p[2];
    double ref[3];
    for (int j = 0; j < 3; ++j) {
        double acc = x * V2C[4 * j];
        acc = std::fma(y, V2C[4 * j + 1], acc);
        acc = std::fma(z, V2C[4 * j + 2], acc);
        ref[j] = std::fma(1.0, V2C[4 * j + 3], acc);
    }
    for (int j = 0; j < 3; ++j) out[j] = std::fma(R0[3 * j + 2], ref[2], std::fma(R0[3 * j + 1], ref[1], R0[3 * j] * ref[0]));
}

}  // namespace

namespace {
// modest_scan_boxes in three host phases around the two device steps, so that a chain of scans can share the device
// steps (one closeness launch over all clusters of the chain, one lowest-point launch over all boxes)
struct BoxRun {
    // inputs
    modest_ctx *ctx = nullptr;
    const float *pts_dev = nullptr, *pts_host = nullptr;
    int n = 0, stride = 0, n_lab = 0;
    int64_t *labels = nullptr;
    const int32_t *members = nullptr;   // indices of the points with a label > 0 (NULL: look at all n)
    int n_members = 0;
    double *objs_out = nullptr;
    int32_t *keep_out = nullptr, *info_out = nullptr;
    // state
    bool any_zero = false, active = false;
    int m = 0;
    std::vector<int32_t> off, coff;
    std::vector<double> xz, miny, boxes6, area, cx, cz;
    double *rect_dev = nullptr;
};

// members_sorted (clustering_utils) + the members' rect-frame points; enqueues the scan's rect projection
int boxes_begin(BoxRun &r, const modest_boxes_params *P, void *stream_, bool defer_rect = false) {
    const int n = r.n, n_lab = r.n_lab, stride = r.stride;
    r.info_out[0] = r.info_out[1] = 0;
    r.off.assign((size_t)n_lab + 2, 0);
    const bool listed = r.members != nullptr;
    const int walk = listed ? r.n_members : n;   // (with a member list the passes touch the members only)
    if (listed) {
        MODEST_REQUIRE(r.n_members >= 0 && r.n_members <= n, "bad member list");
        r.any_zero = r.n_members < n;
    }
    for (int k = 0; k < walk; ++k) {
        const int i = listed ? r.members[k] : k;
        MODEST_REQUIRE(i >= 0 && i < n, "member index out of range");
        const int64_t l = r.labels[i];
        MODEST_REQUIRE(l >= 0 && l <= n_lab && !(listed && l == 0), "label out of range");
        if (l > 0) ++r.off[(size_t)l + 1];
        else r.any_zero = true;
    }
    for (int c = 1; c <= n_lab + 1; ++c) r.off[(size_t)c] += r.off[(size_t)c - 1];   // off[c] = first member of label c
    r.m = r.off[(size_t)n_lab + 1];
    if (n_lab == 0) {   // compact_labels of an all-background scan
        for (int i = 0; i < n; ++i) r.labels[i] = 0;
        return MODEST_OK;
    }
    for (int c = 1; c <= n_lab; ++c) MODEST_REQUIRE(r.off[(size_t)c + 1] > r.off[(size_t)c], "a label without members");
    r.xz.resize((size_t)r.m * 2);
    r.miny.assign((size_t)n_lab, INFINITY);
    {
        std::vector<int32_t> cur(r.off.begin(), r.off.end());
        for (int k = 0; k < walk; ++k) {
            const int i = listed ? r.members[k] : k;
            const int64_t l = r.labels[i];
            if (l <= 0) continue;
            double q[3];
            velo_to_rect(r.pts_host + (size_t)i * stride, P->V2C, P->R0, q);
            const int pos = cur[(size_t)l]++;
            r.xz[2 * (size_t)pos] = q[0];
            r.xz[2 * (size_t)pos + 1] = q[2];
            r.miny[(size_t)l - 1] = std::min(r.miny[(size_t)l - 1], q[1]);   // ptc[:, 1].min() of get_obj
        }
    }
    // the whole scan in the rect frame on the device (the lowest-point search reads it)
    int rc = modest_ctx_reserve_hold(r.ctx, arena_sz((size_t)n * 24), 0);
    if (rc) return rc;
    r.rect_dev = reinterpret_cast<double *>(r.ctx->hold);
    if (!defer_rect) {   // (a chain projects all its scans in one launch)
        rc = modest_project_velo_to_rect(r.ctx, r.pts_dev, n, stride, P->V2C, P->R0, r.rect_dev, stream_);
        if (rc) return rc;
    }
    r.coff.resize((size_t)n_lab + 1);
    for (int c = 0; c <= n_lab; ++c) r.coff[(size_t)c] = r.off[(size_t)c + 1];
    r.active = true;
    return MODEST_OK;
}

// rectangle_at_angle's tail + get_obj up to the lowest-point search; best / ext: this scan's clusters
int boxes_tail(BoxRun &r, const modest_boxes_params *P, const int32_t *best, const double *ext) {
    const int n_lab = r.n_lab;
    r.boxes6.resize((size_t)n_lab * 6);
    r.area.resize((size_t)n_lab);
    r.cx.resize((size_t)n_lab);
    r.cz.resize((size_t)n_lab);
    for (int c = 0; c < n_lab; ++c) {
        const int b = best[c];
        MODEST_REQUIRE(b >= 0 && b < P->n_angles, "bad heading index");
        const double *e = ext + 8 * (size_t)c;
        double angle = P->angles[b], co = P->cossin[2 * b], si = P->cossin[2 * b + 1];
        double min_x = e[0], max_x = e[1], min_y = e[2], max_y = e[3];
        if ((max_x - min_x) < (max_y - min_y)) {
            angle = P->angles[b] + M_PI / 2;
            co = P->cossin90[2 * b];
            si = P->cossin90[2 * b + 1];
            min_x = e[4], max_x = e[5], min_y = e[6], max_y = e[7];
        }
        r.area[(size_t)c] = (max_x - min_x) * (max_y - min_y);
        // rval @ components, components = [[c, s], [-s, c]]
        const double rv[4][2] = {{max_x, min_y}, {min_x, min_y}, {min_x, max_y}, {max_x, max_y}};
        double cor[4][2];
        for (int i = 0; i < 4; ++i) {
            cor[i][0] = dot2(rv[i][0], co, rv[i][1], -si);
            cor[i][1] = dot2(rv[i][0], si, rv[i][1], co);
        }
        const double ry = angle * -1;
        const double d01x = cor[0][0] - cor[1][0], d01y = cor[0][1] - cor[1][1];
        const double d03x = cor[0][0] - cor[3][0], d03y = cor[0][1] - cor[3][1];
        const double l = std::sqrt(dot2(d01x, d01x, d01y, d01y)), w = std::sqrt(dot2(d03x, d03x, d03y, d03y));
        r.cx[(size_t)c] = (cor[0][0] + cor[2][0]) / 2;
        r.cz[(size_t)c] = (cor[0][1] + cor[2][1]) / 2;
        double *o = r.objs_out + 8 * (size_t)c;
        o[3] = l;
        o[4] = w;
        o[6] = ry;
        double *bx = r.boxes6.data() + 6 * (size_t)c;
        bx[0] = r.cx[(size_t)c];
        bx[1] = r.cz[(size_t)c];
        bx[2] = l;
        bx[3] = w;
        bx[4] = co;    // cos(ry) = cos(-angle)
        bx[5] = -si;   // sin(ry)
    }
    return MODEST_OK;
}

// get_obj's remaining fields, the volume gate and the relabelling (generate_mask.py:91-103)
void boxes_finish(BoxRun &r, const modest_boxes_params *P, const double *bottom) {
    const int n_lab = r.n_lab;
    for (int c = 0; c < n_lab; ++c)
        if (std::isinf(bottom[c])) {   // numpy raises on the empty maximum: the host statement reports it
            r.info_out[1] = 2;
            return;
        }
    int n_keep = 0;
    for (int c = 0; c < n_lab; ++c) {
        double *o = r.objs_out + 8 * (size_t)c;
        const double h = bottom[c] - r.miny[(size_t)c];
        o[0] = r.cx[(size_t)c];
        o[1] = bottom[c];
        o[2] = r.cz[(size_t)c];
        o[5] = h;
        o[7] = r.area[(size_t)c] * h;
        r.keep_out[c] = (o[7] > P->min_volume && o[7] < P->max_volume) ? 1 : 0;
        n_keep += r.keep_out[c];
    }
    const bool has_zero = n_keep != n_lab || r.any_zero;
    std::vector<int64_t> table((size_t)n_lab + 1, 0);
    int64_t next = has_zero ? 1 : 0;
    for (int c = 0; c < n_lab; ++c)
        if (r.keep_out[c]) table[(size_t)c + 1] = next++;
    if (r.members) {   // (labels outside the list are 0 and stay 0)
        for (int k = 0; k < r.n_members; ++k) r.labels[r.members[k]] = table[(size_t)r.labels[r.members[k]]];
    } else {
        for (int i = 0; i < r.n; ++i) r.labels[i] = table[(size_t)r.labels[i]];
    }
    r.info_out[0] = n_keep;
}
}  // namespace

extern "C" int modest_scan_boxes(modest_ctx *ctx, const float *pts_dev, const float *pts_host, int n, int stride,
                                 int64_t *labels_inout, int n_lab, const modest_boxes_params *P, double *objs_out,
                                 int32_t *keep_out, int32_t *info_out, void *stream_) {
    MODEST_REQUIRE(ctx && pts_dev && pts_host && labels_inout && P && objs_out && keep_out && info_out, "NULL argument");
    MODEST_REQUIRE(n >= 1 && (stride == 3 || stride == 4) && n_lab >= 0, "bad scan");
    MODEST_REQUIRE(P->angles && P->cossin && P->cossin90 && P->n_angles >= 1, "angle tables missing");
    MODEST_HIP_CHECK(hipSetDevice(ctx->device));
    BoxRun r;
    r.ctx = ctx, r.pts_dev = pts_dev, r.pts_host = pts_host, r.n = n, r.stride = stride, r.n_lab = n_lab;
    r.labels = labels_inout, r.objs_out = objs_out, r.keep_out = keep_out, r.info_out = info_out;
    int rc = boxes_begin(r, P, stream_);
    if (rc || !r.active) return rc;
    std::vector<int32_t> best((size_t)n_lab);
    std::vector<double> ext((size_t)n_lab * 8);
    rc = modest_fit_boxes_closeness_host(ctx, r.xz.data(), r.coff.data(), n_lab, P->cossin, P->n_angles, P->d0, best.data(),
                                         P->cossin90, ext.data(), stream_);
    if (rc) {   // a cluster too large for the extents kernel: the caller's host statement takes the scan
        info_out[1] = 1;
        return MODEST_OK;
    }
    rc = boxes_tail(r, P, best.data(), ext.data());
    if (rc) return rc;
    std::vector<double> bottom((size_t)n_lab);
    rc = modest_lowest_point(ctx, r.rect_dev, n, r.boxes6.data(), n_lab, bottom.data(), stream_);
    if (rc) return rc;
    boxes_finish(r, P, bottom.data());
    return MODEST_OK;
}

// The same for a chain of scans: host phases per scan, ONE closeness launch over all clusters of the chain and ONE
// lowest-point launch over all boxes (two round trips per chain).  Every scan in its own context (its rect-frame
// copy lives there); the shared launches run in the first scan's.  Results are those of separate calls.
extern "C" int modest_scan_boxes_batch(const modest_boxes_scan *scans, int n_scans, const modest_boxes_params *P,
                                       void *stream_) {
    MODEST_REQUIRE(scans && P && n_scans >= 1 && n_scans <= 64, "bad chain");
    MODEST_REQUIRE(P->angles && P->cossin && P->cossin90 && P->n_angles >= 1, "angle tables missing");
    static const bool trace = getenv("MODEST_CHAIN_TRACE") != nullptr;
    auto now = [] {
        timespec ts;
        clock_gettime(CLOCK_MONOTONIC, &ts);
        return 1e3 * (double)ts.tv_sec + 1e-6 * (double)ts.tv_nsec;
    };
    double tt[6] = {now(), 0, 0, 0, 0, 0};
    std::vector<BoxRun> R((size_t)n_scans);
    int total_lab = 0, total_m = 0;
    for (int s = 0; s < n_scans; ++s) {
        const modest_boxes_scan &q = scans[s];
        MODEST_REQUIRE(q.ctx && q.pts_dev && q.pts_host && q.labels_inout && q.objs_out && q.keep_out && q.info_out, "NULL argument");
        MODEST_REQUIRE(q.n >= 1 && (q.stride == 3 || q.stride == 4) && q.n_lab >= 0, "bad scan");
        for (int k = 0; k < s; ++k) MODEST_REQUIRE(scans[k].ctx != q.ctx, "every scan of a chain needs its own context");
        MODEST_HIP_CHECK(hipSetDevice(q.ctx->device));
        BoxRun &r = R[(size_t)s];
        r.ctx = q.ctx, r.pts_dev = q.pts_dev, r.pts_host = q.pts_host, r.n = q.n, r.stride = q.stride, r.n_lab = q.n_lab;
        r.labels = q.labels_inout, r.objs_out = q.objs_out, r.keep_out = q.keep_out, r.info_out = q.info_out;
        r.members = q.members, r.n_members = q.n_members;
        int rc = boxes_begin(r, P, stream_, true);
        if (rc) return rc;
        if (r.active) {
            total_lab += r.n_lab;
            total_m += r.m;
        }
    }
    if (total_lab == 0) return MODEST_OK;
    tt[1] = now();
    {   // the rect-frame copies of all active scans: one launch
        std::vector<const float *> ins;
        std::vector<double *> outs;
        std::vector<int> ns, ss;
        for (BoxRun &r : R)
            if (r.active) {
                ins.push_back(r.pts_dev);
                outs.push_back(r.rect_dev);
                ns.push_back(r.n);
                ss.push_back(r.stride);
            }
        int rc = modest_project_velo_to_rect_multi(R[0].ctx, ins.data(), ns.data(), ss.data(), outs.data(), (int)ins.size(), P->V2C,
                                                   P->R0, stream_);
        if (rc) return rc;
    }
    // all clusters of the chain as one list
    std::vector<double> xz((size_t)total_m * 2);
    std::vector<int32_t> coff((size_t)total_lab + 1, 0), best((size_t)total_lab);
    std::vector<double> ext((size_t)total_lab * 8);
    {
        int c0 = 0, m0 = 0;
        for (BoxRun &r : R) {
            if (!r.active) continue;
            memcpy(xz.data() + 2 * (size_t)m0, r.xz.data(), (size_t)r.m * 16);
            for (int c = 0; c <= r.n_lab; ++c) coff[(size_t)c0 + c] = m0 + r.coff[(size_t)c];
            c0 += r.n_lab;
            m0 += r.m;
        }
    }
    modest_ctx *ctx0 = nullptr;
    for (BoxRun &r : R)
        if (r.active) {
            ctx0 = r.ctx;
            break;
        }
    tt[2] = now();
    int rc = modest_fit_boxes_closeness_host(ctx0, xz.data(), coff.data(), total_lab, P->cossin, P->n_angles, P->d0, best.data(),
                                             P->cossin90, ext.data(), stream_);
    tt[3] = now();
    if (rc) {   // a cluster too large for the extents kernel somewhere: every scan reports it, the caller goes scan by scan
        for (BoxRun &r : R)
            if (r.active) r.info_out[1] = 1;
        return MODEST_OK;
    }
    std::vector<double> boxes6((size_t)total_lab * 6), bottom((size_t)total_lab);
    std::vector<const double *> src((size_t)total_lab);
    std::vector<int> nsrc((size_t)total_lab);
    {
        int c0 = 0;
        for (BoxRun &r : R) {
            if (!r.active) continue;
            rc = boxes_tail(r, P, best.data() + c0, ext.data() + 8 * (size_t)c0);
            if (rc) return rc;
            memcpy(boxes6.data() + 6 * (size_t)c0, r.boxes6.data(), (size_t)r.n_lab * 48);
            for (int c = 0; c < r.n_lab; ++c) {
                src[(size_t)c0 + c] = r.rect_dev;
                nsrc[(size_t)c0 + c] = r.n;
            }
            c0 += r.n_lab;
        }
    }
    tt[4] = now();
    rc = modest_lowest_point_multi(ctx0, src.data(), nsrc.data(), boxes6.data(), total_lab, bottom.data(), stream_);
    if (rc) return rc;
    tt[5] = now();
    {
        int c0 = 0;
        for (BoxRun &r : R) {
            if (!r.active) continue;
            boxes_finish(r, P, bottom.data() + c0);
            c0 += r.n_lab;
        }
    }
    if (trace)
        fprintf(stderr, "[scan_boxes_batch %d scans, %d clusters, %d members] begin %.3f | rect + pack %.3f | closeness %.3f | tail %.3f | lowest %.3f | finish %.3f ms\n",
                n_scans, total_lab, total_m, tt[1] - tt[0], tt[2] - tt[1], tt[3] - tt[2], tt[4] - tt[3], tt[5] - tt[4], now() - tt[5]);
    return MODEST_OK;
}

extern "C" int modest_objs_iou(modest_ctx *ctx, const double *objs8, int k, float *iou_out, void *stream_) {
    MODEST_REQUIRE(ctx && k >= 0, "bad argument");
    if (k == 0) return MODEST_OK;
    MODEST_REQUIRE(objs8 && iou_out, "NULL buffer");
    // [t0, t2, 0, l, w, h, -ry] as float32 (pointcloud_utils.py:322-324: float64 rows, then .float())
    std::vector<float> boxes((size_t)k * 7);
    for (int i = 0; i < k; ++i) {
        const double *o = objs8 + 8 * (size_t)i;
        float *b = boxes.data() + 7 * (size_t)i;
        b[0] = (float)o[0];
        b[1] = (float)o[2];
        b[2] = 0.f;
        b[3] = (float)o[3];
        b[4] = (float)o[4];
        b[5] = (float)o[5];
        b[6] = (float)(-o[6]);
    }
    return modest_boxes_iou_bev_host(ctx, boxes.data(), k, boxes.data(), k, iou_out, stream_);
}

// modest_objs_iou for the box sets of a chain of scans: one launch, one round trip
extern "C" int modest_objs_iou_batch(modest_ctx *ctx, const double *const *objs8, const int32_t *k, int n_sets,
                                     float *const *iou_out, void *stream_) {
    MODEST_REQUIRE(ctx && objs8 && k && iou_out && n_sets >= 1 && n_sets <= 64, "bad arguments");
    std::vector<std::vector<float>> boxes((size_t)n_sets);
    std::vector<const float *> bp((size_t)n_sets);
    std::vector<int> kv((size_t)n_sets);
    for (int s = 0; s < n_sets; ++s) {
        MODEST_REQUIRE(k[s] >= 0 && (k[s] == 0 || (objs8[s] && iou_out[s])), "bad box set");
        boxes[(size_t)s].resize((size_t)k[s] * 7);
        for (int i = 0; i < k[s]; ++i) {   // [t0, t2, 0, l, w, h, -ry] as float32 (pointcloud_utils.py:322-324)
            const double *o = objs8[s] + 8 * (size_t)i;
            float *b = boxes[(size_t)s].data() + 7 * (size_t)i;
            b[0] = (float)o[0];
            b[1] = (float)o[2];
            b[2] = 0.f;
            b[3] = (float)o[3];
            b[4] = (float)o[4];
            b[5] = (float)o[5];
            b[6] = (float)(-o[6]);
        }
        bp[(size_t)s] = boxes[(size_t)s].data();
        kv[(size_t)s] = k[s];
    }
    return modest_boxes_self_iou_bev_host_batch(ctx, bp.data(), kv.data(), n_sets, iou_out, stream_);
}

extern "C" int modest_label_lines(const double *objs8, const double *cossin_ry, int k, const int64_t *order,
                                  const float *iou, const modest_labels_params *P, int32_t *kept_out, int32_t *n_kept_out,
                                  char *text_out, int32_t text_cap, int32_t *text_len_out) {
    MODEST_REQUIRE(P && n_kept_out && text_len_out && k >= 0 && text_cap >= 1, "bad argument");
    *n_kept_out = 0;
    *text_len_out = 0;
    text_out[0] = 0;
    if (k == 0) return MODEST_OK;
    MODEST_REQUIRE(objs8 && cossin_ry && kept_out && text_out, "NULL buffer");
    std::vector<char> mask((size_t)k, 1);
    if (P->nms_enable) {   // objs_nms (pointcloud_utils.py:329-343): original order is kept
        MODEST_REQUIRE(order && iou, "NMS needs the walk order and the IoU matrix");
        const float thr = P->nms_threshold;
        for (int q = 0; q < k; ++q) {
            const int64_t idx = order[q];
            MODEST_REQUIRE(idx >= 0 && idx < k, "order out of range");
            if (!mask[(size_t)idx]) continue;
            const float *row = iou + (size_t)idx * k;
            for (int j = 0; j < k; ++j)
                if (row[j] > thr) mask[(size_t)j] = 0;
            mask[(size_t)idx] = 1;
        }
    }
    const double *Pm = P->P;
    int nk = 0, len = 0;
    for (int i = 0; i < k; ++i) {
        if (!mask[(size_t)i]) continue;
        const double *o = objs8 + 8 * (size_t)i;
        const double t0 = o[0], t1 = o[1], t2 = o[2], l = o[3], w = o[4], h = o[5], ry = o[6];
        if (P->fov_only) {   // is_within_fov (:373-379)
            const double cen[4] = {t0, t1 - h / 2, t2, 1.0};
            const double u = dot4(cen, Pm) / dot4(cen, Pm + 8), v = dot4(cen, Pm + 4) / dot4(cen, Pm + 8);
            if (!(u < P->image_w && u >= 0 && v < P->image_h && v >= 0 && cen[2] > 0)) continue;
        }
        // compute_box_3d (kitti_util.py:453-488): R = roty(ry), corners, + t, project_to_image
        const double c = cossin_ry[2 * i], s = cossin_ry[2 * i + 1];
        const double xc[8] = {l / 2, l / 2, -l / 2, -l / 2, l / 2, l / 2, -l / 2, -l / 2};
        const double yc[8] = {0, 0, 0, 0, -h, -h, -h, -h};
        const double zc[8] = {w / 2, -w / 2, -w / 2, w / 2, w / 2, -w / 2, -w / 2, w / 2};
        double lo_u = INFINITY, lo_v = INFINITY, hi_u = -INFINITY, hi_v = -INFINITY;
        for (int j = 0; j < 8; ++j) {
            const double X[4] = {dot3(c, xc[j], 0.0, yc[j], s, zc[j]) + t0, dot3(0.0, xc[j], 1.0, yc[j], 0.0, zc[j]) + t1,
                                 dot3(-s, xc[j], 0.0, yc[j], c, zc[j]) + t2, 1.0};
            const double d = dot4(X, Pm + 8), u = dot4(X, Pm) / d, v = dot4(X, Pm + 4) / d;
            lo_u = std::min(lo_u, u);
            hi_u = std::max(hi_u, u);
            lo_v = std::min(lo_v, v);
            hi_v = std::max(hi_v, v);
        }
        const double alpha = -std::atan2(t0, t2) + ry;
        const double f[12] = {alpha, lo_u, lo_v, hi_u, hi_v, h, w, l, t0, t1, t2, ry};
        // a projected corner next to the image plane prints hundreds of digits with %.4f: 12 x (1 + 309 + 5) + the prefix
        char line[4096];
        int p = snprintf(line, sizeof(line), "%sDynamic -1 -1", nk ? "\n" : "");
        for (int q = 0; q < 12; ++q) {
            const int w_ = snprintf(line + p, sizeof(line) - (size_t)p, " %.4f", f[q]);
            if (w_ < 0 || (size_t)(p + w_) >= sizeof(line)) {   // (cannot happen for finite doubles; never read past the buffer)
                modest_set_error("modest_label_lines: a label line does not fit %zu bytes", sizeof(line));
                return MODEST_ERR_CAPACITY;
            }
            p += w_;
        }
        if (len + p + 1 > text_cap) {
            modest_set_error("modest_label_lines: text buffer of %d bytes is too small", text_cap);
            return MODEST_ERR_CAPACITY;
        }
        memcpy(text_out + len, line, (size_t)p);
        len += p;
        kept_out[nk++] = i;
    }
    text_out[len] = 0;
    *n_kept_out = nk;
    *text_len_out = len;
    return MODEST_OK;
}
