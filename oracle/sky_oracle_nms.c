/*
 * ORACLE -- TEST INFRASTRUCTURE ONLY.  Not part of the product path.
 *
 * Plain-C restatement of the reference's post-processing wrapper
 *     non_max_suppression   skyeye/utils/metrics.py:361-457
 * for ONE image, plus the greedy suppression it delegates to
 *     torchvision.ops.nms   call site metrics.py:442
 * torchvision is a third-party dependency that is absent from /root/reference
 * and from this image (requirements.txt:2 / setup.py:25 say only
 * "torchvision>=0.8.1", no pin, no lock file).  Its published algorithm is
 * restated here: visit boxes by descending score; keep a box unless an already
 * kept box overlaps it with IoU > threshold (strict); IoU = inter / (area_i +
 * area_j - inter), area = (x2-x1)*(y2-y1), inter = max(0,.)*max(0,.), all in
 * fp32.  Tie order among equal scores is unspecified upstream; this build
 * fixes it as "lower input index first".
 *
 * PARITY STATUS: the wrapper (filtering, column assembly, caps, offsets) is
 * pinned by tests/golden/nms.npz, produced by executing the reference wrapper
 * with this same greedy rule injected as metrics.torchvision
 * (tests/golden/make_golden.py).  The suppression core itself has no reference
 * vector to check against: PARITY UNPINNED for torchvision.ops.nms.
 *
 * mode 0 ("literal")  = the file as written:
 *     boxes stay (cx,cy,w,h) and are fed to NMS as if they were corners,
 *     score = objectness only, class offset = column 5 * 4096 (the class
 *     *confidence* for nc>1), output rows are 7 wide for nc>1, 6 for nc==1.
 * mode 1 ("corrected") = the YOLOv5 semantics the reference imitates:
 *     conf = obj*cls, boxes converted to corners, offset = class id * 4096,
 *     rows [x1,y1,x2,y2,conf,cls].  Build-defined, no reference counterpart.
 * mode 2 ("rows") = the input rows are already boxes (x1,y1,x2,y2,conf,cls,...;
 *     stride nc+5): the cross-tile stage of tiled inference over the survivors of
 *     the per-tile NMS.  Build-defined, no reference counterpart.
 *
 * Compile with -ffp-contract=off: every product and sum must round exactly as
 * the fp32 tensor ops of the reference do.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

typedef struct {
    float score;
    int idx;
} sk_t;

static int cmp_desc(const void* a, const void* b)
{
    const sk_t* p = (const sk_t*)a;
    const sk_t* q = (const sk_t*)b;
    if (p->score > q->score) return -1;
    if (p->score < q->score) return 1;
    return p->idx < q->idx ? -1 : (p->idx > q->idx ? 1 : 0);
}

/*
 * pred  [n, nc+5] fp32 rows (cx,cy,w,h,obj,cls...)
 * out   [max_det, 7] fp32 (only the first `cols` columns of each row are meaningful,
 *       rows are packed with stride `cols`)
 * returns the number of rows written; *cols_out = 6 or 7
 */
int sky_oracle_nms_image(const float* pred, int n, int nc, float conf_thres, float iou_thres,
                         const int* classes, int n_classes, int agnostic, int multi_label, int max_det,
                         int max_nms, float max_wh, int mode, float* out, int* cols_out)
{
    const int no = nc + 5;
    const int cols = (mode == 0 && nc > 1) ? 7 : 6;
    *cols_out = cols;
    if (mode == 0) multi_label = multi_label && nc > 1;          /* metrics.py:396 */
    /* ---- candidate rows x[m, 7] = (b0,b1,b2,b3, c4, c5, c6) ---- */
    size_t cap = 1024, m = 0;
    float* x = (float*)malloc(cap * 7 * sizeof(float));
#define PUSH(B0, B1, B2, B3, C4, C5, C6)                                   \
    do {                                                                    \
        if (m == cap) {                                                     \
            cap *= 2;                                                       \
            x = (float*)realloc(x, cap * 7 * sizeof(float));                \
        }                                                                   \
        float* r_ = x + m * 7;                                              \
        r_[0] = (B0); r_[1] = (B1); r_[2] = (B2); r_[3] = (B3);             \
        r_[4] = (C4); r_[5] = (C5); r_[6] = (C6);                           \
        ++m;                                                                \
    } while (0)
    for (int i = 0; i < n; ++i) {
        const float* p = pred + (size_t)i * no;
        if (!(p[4] > conf_thres)) continue;                      /* metrics.py:389,402 */
        if (mode == 2) {
            PUSH(p[0], p[1], p[2], p[3], p[4], p[5], 0.0f);
            continue;
        }
        if (mode == 0) {
            if (nc > 1) {
                if (multi_label) {                               /* metrics.py:412-413 */
                    for (int j = 0; j < nc; ++j)
                        if (p[5 + j] > conf_thres) PUSH(p[0], p[1], p[2], p[3], p[4], p[5 + j], (float)j);
                } else {                                         /* metrics.py:416-417 */
                    int bj = 0;
                    float bc = p[5];
                    for (int j = 1; j < nc; ++j)
                        if (p[5 + j] > bc) { bc = p[5 + j]; bj = j; }   /* first maximal index */
                    if (bc > conf_thres) PUSH(p[0], p[1], p[2], p[3], p[4], bc, (float)bj);
                }
            } else {                                             /* metrics.py:420-422 */
                PUSH(p[0], p[1], p[2], p[3], p[4], 0.0f, 0.0f);
            }
        } else {
            const float hw = p[2] / 2.0f, hh = p[3] / 2.0f;
            const float x1 = p[0] - hw, y1 = p[1] - hh, x2 = p[0] + hw, y2 = p[1] + hh;
            if (multi_label && nc > 1) {
                for (int j = 0; j < nc; ++j) {
                    const float c = p[5 + j] * p[4];
                    if (c > conf_thres) PUSH(x1, y1, x2, y2, c, (float)j, 0.0f);
                }
            } else {
                int bj = 0;
                float bc = p[5] * p[4];
                for (int j = 1; j < nc; ++j) {
                    const float c = p[5 + j] * p[4];
                    if (c > bc) { bc = c; bj = j; }
                }
                if (bc > conf_thres) PUSH(x1, y1, x2, y2, bc, (float)bj, 0.0f);
            }
        }
    }
#undef PUSH
    /* ---- class filter: compares column 5 with the requested ids (metrics.py:425-426) ---- */
    if (classes && n_classes > 0) {
        size_t k = 0;
        for (size_t i = 0; i < m; ++i) {
            int hit = 0;
            for (int c = 0; c < n_classes; ++c) hit |= (x[i * 7 + 5] == (float)classes[c]);
            if (hit) {
                if (k != i) memcpy(x + k * 7, x + i * 7, 7 * sizeof(float));
                ++k;
            }
        }
        m = k;
    }
    if (m == 0) {
        free(x);
        return 0;
    }
    /* ---- order by score (column 4), descending, ties by candidate index;
     *      keep the best max_nms (metrics.py:434-435) ---- */
    sk_t* ord = (sk_t*)malloc(m * sizeof(sk_t));
    for (size_t i = 0; i < m; ++i) { ord[i].score = x[i * 7 + 4]; ord[i].idx = (int)i; }
    qsort(ord, m, sizeof(sk_t), cmp_desc);
    if ((long)m > (long)max_nms) m = (size_t)max_nms;
    /* ---- greedy suppression on boxes offset per class (metrics.py:438-444) ---- */
    float* kb = (float*)malloc((size_t)max_det * 5 * sizeof(float));  /* x1,y1,x2,y2,area of kept */
    int kept = 0;
    for (size_t s = 0; s < m && kept < max_det; ++s) {
        const float* r = x + (size_t)ord[s].idx * 7;
        const float c = agnostic ? r[5] * 0.0f : r[5] * max_wh;      /* metrics.py:438 */
        const float x1 = r[0] + c, y1 = r[1] + c, x2 = r[2] + c, y2 = r[3] + c;
        const float area = (x2 - x1) * (y2 - y1);
        int dead = 0;
        for (int k = 0; k < kept && !dead; ++k) {
            const float* q = kb + (size_t)k * 5;
            float w = fminf(q[2], x2) - fmaxf(q[0], x1);
            float h = fminf(q[3], y2) - fmaxf(q[1], y1);
            w = w > 0.0f ? w : 0.0f;
            h = h > 0.0f ? h : 0.0f;
            const float inter = w * h;
            const float iou = inter / (q[4] + area - inter);
            dead = iou > iou_thres;
        }
        if (dead) continue;
        float* q = kb + (size_t)kept * 5;
        q[0] = x1; q[1] = y1; q[2] = x2; q[3] = y2; q[4] = area;
        memcpy(out + (size_t)kept * cols, r, (size_t)cols * sizeof(float));
        ++kept;
    }
    free(kb);
    free(ord);
    free(x);
    return kept;
}
