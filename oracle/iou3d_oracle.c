/* ORACLE — test infrastructure only, never linked into the product.
 *
 * Plain-C restatement of the reference's rotated BEV IoU (the arithmetic shared
 * by generate_cluster_mask/utils/iou3d_nms/src/iou3d_cpu.cpp:128-229 and the
 * CUDA kernel src/iou3d_nms_kernel.cu:104-234): float32 operations in the
 * reference's order, glibc cosf/sinf/atan2f.  Build with -ffp-contract=off
 * (oracle/Makefile).  Validated against the reference's own iou3d_cpu.cpp
 * compiled where it lies (oracle/build_ref.py -> oracle/_ref/iou3d_ref.so) and
 * against tests/golden/boxes_iou.npz in tests/test_oracle_mask.py (test_iou_oracle_*).
 */
#include <math.h>
#include <stdint.h>

#define EPSF 1e-8f

typedef struct { float x, y; } P2;

static float cross2(P2 a, P2 b) { return a.x * b.y - a.y * b.x; }                 /* cpp:58-60 */
static float cross3(P2 p1, P2 p2, P2 p0) {                                       /* cpp:62-64 */
    return (p1.x - p0.x) * (p2.y - p0.y) - (p2.x - p0.x) * (p1.y - p0.y);
}
static float fmin2(float a, float b) { return a > b ? b : a; }                    /* cpp:33-39 */
static float fmax2(float a, float b) { return a > b ? a : b; }

static int rect_cross(P2 p1, P2 p2, P2 q1, P2 q2) {                               /* cpp:66-72 */
    return fmin2(p1.x, p2.x) <= fmax2(q1.x, q2.x) && fmin2(q1.x, q2.x) <= fmax2(p1.x, p2.x) &&
           fmin2(p1.y, p2.y) <= fmax2(q1.y, q2.y) && fmin2(q1.y, q2.y) <= fmax2(p1.y, p2.y);
}

static int in_box2d(const float *box, P2 p) {                                     /* cpp:74-84 */
    const float MARGIN = 1e-2f;
    float cx = box[0], cy = box[1];
    float c = cosf(-box[6]), s = sinf(-box[6]);
    float rx = (p.x - cx) * c + (p.y - cy) * (-s);
    float ry = (p.x - cx) * s + (p.y - cy) * c;
    return fabsf(rx) < box[3] / 2 + MARGIN && fabsf(ry) < box[4] / 2 + MARGIN;
}

static int seg_x(P2 p1, P2 p0, P2 q1, P2 q0, P2 *ans) {                           /* cpp:86-115 */
    if (!rect_cross(p0, p1, q0, q1)) return 0;
    float s1 = cross3(q0, p1, p0), s2 = cross3(p1, q1, p0);
    float s3 = cross3(p0, q1, q0), s4 = cross3(q1, p1, q0);
    if (!(s1 * s2 > 0 && s3 * s4 > 0)) return 0;
    float s5 = cross3(q1, p1, p0);
    if (fabsf(s5 - s1) > EPSF) {
        ans->x = (s5 * q0.x - s1 * q1.x) / (s5 - s1);
        ans->y = (s5 * q0.y - s1 * q1.y) / (s5 - s1);
    } else {
        float a0 = p0.y - p1.y, b0 = p1.x - p0.x, c0 = p0.x * p1.y - p1.x * p0.y;
        float a1 = q0.y - q1.y, b1 = q1.x - q0.x, c1 = q0.x * q1.y - q1.x * q0.y;
        float D = a0 * b1 - a1 * b0;
        ans->x = (b0 * c1 - b1 * c0) / D;
        ans->y = (a1 * c0 - a0 * c1) / D;
    }
    return 1;
}

static void rot(P2 ctr, float c, float s, P2 *p) {                                /* cpp:117-121 */
    float nx = (p->x - ctr.x) * c + (p->y - ctr.y) * (-s) + ctr.x;
    float ny = (p->x - ctr.x) * s + (p->y - ctr.y) * c + ctr.y;
    p->x = nx; p->y = ny;
}

float modest_oracle_box_overlap(const float *a, const float *b) {                 /* cpp:128-220 */
    float aa = a[6], ba = b[6];
    float adx = a[3] / 2, bdx = b[3] / 2, ady = a[4] / 2, bdy = b[4] / 2;
    P2 ca = {a[0], a[1]}, cb = {b[0], b[1]};
    P2 A[5] = {{a[0] - adx, a[1] - ady}, {a[0] + adx, a[1] - ady}, {a[0] + adx, a[1] + ady}, {a[0] - adx, a[1] + ady}};
    P2 B[5] = {{b[0] - bdx, b[1] - bdy}, {b[0] + bdx, b[1] - bdy}, {b[0] + bdx, b[1] + bdy}, {b[0] - bdx, b[1] + bdy}};
    float ac = cosf(aa), as = sinf(aa), bc = cosf(ba), bs = sinf(ba);
    for (int k = 0; k < 4; k++) { rot(ca, ac, as, &A[k]); rot(cb, bc, bs, &B[k]); }
    A[4] = A[0]; B[4] = B[0];
    P2 poly[24], ctr = {0.f, 0.f};
    int cnt = 0;
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++) {
            P2 x;
            if (seg_x(A[i + 1], A[i], B[j + 1], B[j], &x)) { ctr.x = ctr.x + x.x; ctr.y = ctr.y + x.y; poly[cnt++] = x; }
        }
    for (int k = 0; k < 4; k++) {
        if (in_box2d(a, B[k])) { ctr.x = ctr.x + B[k].x; ctr.y = ctr.y + B[k].y; poly[cnt++] = B[k]; }
        if (in_box2d(b, A[k])) { ctr.x = ctr.x + A[k].x; ctr.y = ctr.y + A[k].y; poly[cnt++] = A[k]; }
    }
    ctr.x /= cnt; ctr.y /= cnt;
    for (int j = 0; j < cnt - 1; j++)
        for (int i = 0; i < cnt - j - 1; i++)
            if (atan2f(poly[i].y - ctr.y, poly[i].x - ctr.x) > atan2f(poly[i + 1].y - ctr.y, poly[i + 1].x - ctr.x)) {
                P2 t = poly[i]; poly[i] = poly[i + 1]; poly[i + 1] = t;
            }
    float area = 0;
    for (int k = 0; k < cnt - 1; k++) {
        P2 u = {poly[k].x - poly[0].x, poly[k].y - poly[0].y};
        P2 v = {poly[k + 1].x - poly[0].x, poly[k + 1].y - poly[0].y};
        area += cross2(u, v);
    }
    return fabsf(area) / 2.0f;
}

float modest_oracle_iou_bev(const float *a, const float *b) {                     /* cpp:222-229 */
    float sa = a[3] * a[4], sb = b[3] * b[4];
    float so = modest_oracle_box_overlap(a, b);
    return so / fmaxf(sa + sb - so, EPSF);
}

/* boxes_iou_bev_cpu, cpp:232-252; mode 0 = IoU, 1 = raw overlap (kernel.cu:236-249) */
void modest_oracle_boxes_bev(const float *A, int na, const float *B, int nb, float *out, int mode) {
    for (int i = 0; i < na; i++)
        for (int j = 0; j < nb; j++)
            out[(long)i * nb + j] = mode ? modest_oracle_box_overlap(A + 7 * i, B + 7 * j)
                                         : modest_oracle_iou_bev(A + 7 * i, B + 7 * j);
}

static float iou_normal(const float *a, const float *b) {                         /* kernel.cu:314-325 */
    float left = fmaxf(a[0] - a[3] / 2, b[0] - b[3] / 2), right = fminf(a[0] + a[3] / 2, b[0] + b[3] / 2);
    float top = fmaxf(a[1] - a[4] / 2, b[1] - b[4] / 2), bottom = fminf(a[1] + a[4] / 2, b[1] + b[4] / 2);
    float w = fmaxf(right - left, 0.f), h = fmaxf(bottom - top, 0.f);
    float inter = w * h, sa = a[3] * a[4], sb = b[3] * b[4];
    return inter / fmaxf(sa + sb - inter, EPSF);
}

/* nms_gpu / nms_normal_gpu semantics (src/iou3d_nms.cpp:90-186): greedy over score-sorted boxes */
int modest_oracle_nms(const float *boxes, int n, float thresh, int rotated, int64_t *keep) {
    int kept = 0;
    for (int i = 0; i < n; i++) {
        int sup = 0;
        for (int k = 0; k < kept && !sup; k++) {
            const float *a = boxes + 7 * keep[k], *b = boxes + 7 * i;
            float v = rotated ? modest_oracle_iou_bev(a, b) : iou_normal(a, b);
            if (v > thresh) sup = 1;
        }
        if (!sup) keep[kept++] = i;
    }
    return kept;
}
