/*
 * glue.c — TEST INFRASTRUCTURE ONLY (see oracle.h).
 *
 * Plain-C restatement of the host-side glue the reference crate itself implements around the network:
 * SSD anchors, box decode, sigmoid, thresholding, weighted NMS, letterbox removal, landmark projection and
 * the ROI maths.  Dtypes and evaluation order follow the Rust source literally (f32 where Rust uses f32,
 * f64 where it uses f64, no fused multiply-add: build with -ffp-contract=off).
 * Paths below are relative to /root/reference/src/face_detection_lite/.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "oracle.h"

/* ---------------------------------------------------------------- SSD anchors: face_detection.rs:28-86, 366-413 */
typedef struct {
    int num_layers, input_h, input_w;
    float off_x, off_y;
    int strides[4];
    float interp_ratio;
} ssd_opts;

static int ssd_options(int kind, ssd_opts *o) {
    switch (kind) {
    case ORC_FD_FRONT: /* SSDOptions::new_front, face_detection.rs:39-49 */
    case ORC_FD_SHORT: /* SSDOptions::new_short, face_detection.rs:63-73 */
        *o = (ssd_opts){4, 128, 128, 0.5f, 0.5f, {8, 16, 16, 16}, 1.0f};
        return 0;
    case ORC_FD_BACK: /* SSDOptions::new_back, face_detection.rs:51-61 */
        *o = (ssd_opts){4, 256, 256, 0.5f, 0.5f, {16, 32, 32, 32}, 1.0f};
        return 0;
    case ORC_FD_FULL: /* SSDOptions::new_full, face_detection.rs:75-85 (shared by Full and FullSparse, 176-183) */
    case ORC_FD_FULL_SPARSE:
        *o = (ssd_opts){1, 192, 192, 0.5f, 0.5f, {4, 0, 0, 0}, 0.0f};
        return 0;
    default:
        return -1;
    }
}

int orc_fd_input_size(int kind) {
    ssd_opts o;
    return ssd_options(kind, &o) == 0 ? o.input_h : -1;
}

int orc_ssd_anchors(int kind, float *out, int cap) {
    ssd_opts o;
    if (ssd_options(kind, &o) != 0) return -1;
    int n = 0, layer_id = 0;
    while (layer_id < o.num_layers) { /* face_detection.rs:378-405 */
        int last = layer_id, repeats = 0;
        while (last < o.num_layers && o.strides[last] == o.strides[layer_id]) {
            last += 1;
            repeats += (o.interp_ratio == 1.0f) ? 2 : 1;
        }
        int stride = o.strides[layer_id];
        int fm_h = o.input_h / stride, fm_w = o.input_w / stride;
        for (int y = 0; y < fm_h; y++) {
            float y_center = ((float)y + o.off_y) / (float)fm_h;
            for (int x = 0; x < fm_w; x++) {
                float x_center = ((float)x + o.off_x) / (float)fm_w;
                for (int r = 0; r < repeats; r++) {
                    if (out && n < cap) { out[2 * n] = x_center; out[2 * n + 1] = y_center; }
                    n++;
                }
            }
        }
        layer_id = last;
    }
    return n;
}

/* ---------------------------------------------------------------- decode: face_detection.rs:269-296 */
void orc_decode_boxes(const float *raw, const float *anchors, int n, float scale, float *out) {
    for (int a = 0; a < n; a++) {
        float b[16];
        for (int k = 0; k < 16; k++) b[k] = raw[16 * a + k] / scale; /* mapv(|x| x / scale), 274 */
        float ax = anchors[2 * a], ay = anchors[2 * a + 1];
        b[0] += ax; b[1] += ay;                                       /* boxes[:,0,:] += anchors, 276-277 */
        for (int i = 2; i < 8; i++) { b[2 * i] += ax; b[2 * i + 1] += ay; } /* 279-282 */
        float cx = b[0], cy = b[1];                                   /* center, 284 */
        float hx = b[2] / 2.0f, hy = b[3] / 2.0f;                     /* half_size, 285 */
        b[0] = cx - hx; b[1] = cy - hy;                               /* 286-289 */
        b[2] = cx + hx; b[3] = cy + hy;                               /* 290-293 */
        memcpy(out + 16 * a, b, sizeof(b));
    }
}

/* ---------------------------------------------------------------- sigmoid: face_detection.rs:300-314, transform.rs:111-113 */
static float sigmoid_f32(float x) { return 1.0f / (1.0f + expf(-x)); }

void orc_sigmoid_scores(const float *raw, int n, float *out) {
    const float lim = 80.0f; /* RAW_SCORE_LIMIT, face_detection.rs:133 */
    for (int i = 0; i < n; i++) {
        float x = raw[i];
        if (x < -lim) x = -lim; else if (x > lim) x = lim;
        out[i] = sigmoid_f32(x);
    }
}

/* ---------------------------------------------------------------- convert_to_detections: face_detection.rs:317-362 */
int orc_convert_to_detections(const float *boxes, const float *scores, int n, orc_detection *out) {
    int m = 0;
    for (int a = 0; a < n; a++) {
        if (!(scores[a] > 0.5f)) continue;                     /* MIN_SCORE, 326 */
        const float *b = boxes + 16 * a;
        if (!(b[2] > b[0] && b[3] > b[1])) continue;           /* is_valid: row1 > row0 elementwise, 318-323 */
        memcpy(out[m].data, b, 16 * sizeof(float));
        out[m].score = scores[a];
        m++;
    }
    return m;
}

/* ---------------------------------------------------------------- IoU: nms.rs:5-17, types.rs:99-159 */
typedef struct { double xmin, ymin, xmax, ymax; } bbox64;

static bbox64 det_bbox(const orc_detection *d) { /* Detection::bbox, types.rs:219-225 */
    bbox64 b = {(double)d->data[0], (double)d->data[1], (double)d->data[2], (double)d->data[3]};
    return b;
}
static double bbox_area(const bbox64 *b) { /* BBox::area/empty, types.rs:123-145 */
    double w = b->xmax - b->xmin, h = b->ymax - b->ymin;
    if (w <= 0.0 || h <= 0.0) return 0.0;
    return w * h;
}
static double overlap_similarity(const bbox64 *b1, const bbox64 *b2) {
    bbox64 i = {fmax(b1->xmin, b2->xmin), fmax(b1->ymin, b2->ymin), fmin(b1->xmax, b2->xmax), fmin(b1->ymax, b2->ymax)};
    if (!(i.xmin < i.xmax && i.ymin < i.ymax)) return 0.0; /* BBox::intersect -> None, types.rs:148-159 */
    double ia = bbox_area(&i);
    double den = bbox_area(b1) + bbox_area(b2) - ia;
    return den > 0.0 ? ia / den : 0.0;
}

typedef struct { int index; float score; } iscore;

/* stable descending sort by score (nms.rs:137: sort_by(|a,b| b.1.partial_cmp(&a.1)) is a stable merge sort) */
static void stable_sort_desc(iscore *v, int n) {
    iscore *tmp = (iscore *)malloc((size_t)(n > 0 ? n : 1) * sizeof(iscore));
    for (int w = 1; w < n; w *= 2) {
        for (int lo = 0; lo < n; lo += 2 * w) {
            int mid = lo + w < n ? lo + w : n, hi = lo + 2 * w < n ? lo + 2 * w : n;
            int i = lo, j = mid, k = lo;
            while (i < mid && j < hi) tmp[k++] = (v[j].score > v[i].score) ? v[j++] : v[i++]; /* ties keep left */
            while (i < mid) tmp[k++] = v[i++];
            while (j < hi) tmp[k++] = v[j++];
        }
        memcpy(v, tmp, (size_t)n * sizeof(iscore));
    }
    free(tmp);
}

int orc_weighted_nms(const orc_detection *dets, int n, float min_supp, int has_min_score, float min_score,
                     orc_detection *out) {
    if (n <= 0) return 0;
    iscore *rem = (iscore *)malloc((size_t)n * sizeof(iscore));
    iscore *next = (iscore *)malloc((size_t)n * sizeof(iscore));
    iscore *cand = (iscore *)malloc((size_t)n * sizeof(iscore));
    for (int i = 0; i < n; i++) { rem[i].index = i; rem[i].score = dets[i].score; } /* nms.rs:130-134 */
    stable_sort_desc(rem, n);
    int nrem = n, nout = 0;
    const double thr = (double)min_supp; /* `min_suppression_threshold as f64`, nms.rs:87 */
    while (nrem > 0) { /* nms.rs:65-121 */
        const orc_detection *head = &dets[rem[0].index];
        if (has_min_score && head->score < min_score) break; /* 69-73 */
        int num_prev = nrem, nnext = 0, ncand = 0;
        bbox64 hb = det_bbox(head);
        orc_detection wdet = *head; /* weighted_detection = detection.clone(), 81 */
        for (int k = 0; k < nrem; k++) { /* 83-92 */
            bbox64 rb = det_bbox(&dets[rem[k].index]);
            double sim = overlap_similarity(&rb, &hb);
            if (sim > thr) cand[ncand++] = rem[k]; else next[nnext++] = rem[k];
        }
        if (ncand > 0) { /* 94-112 */
            float w[16];
            for (int f = 0; f < 16; f++) w[f] = 0.0f;
            float total = 0.0f;
            for (int k = 0; k < ncand; k++) {
                float s = cand[k].score;
                total += s;
                const float *d = dets[cand[k].index].data;
                for (int f = 0; f < 16; f++) w[f] += d[f] * s; /* *w += d * score, 102-106 */
            }
            for (int f = 0; f < 16; f++) wdet.data[f] = w[f] / total; /* weighted /= total_score, 109 */
            wdet.score = head->score; /* Detection::new(w_det, detection.score), 112 */
        }
        out[nout++] = wdet;
        if (num_prev == nnext) break; /* 117-119 */
        iscore *t = rem; rem = next; next = t;
        nrem = nnext;
    }
    free(rem); free(next); free(cand);
    return nout;
}

int orc_plain_nms(const orc_detection *dets, int n, float min_supp, int has_min_score, float min_score,
                  orc_detection *out) { /* nms.rs:19-53 */
    if (n <= 0) return 0;
    iscore *order = (iscore *)malloc((size_t)n * sizeof(iscore));
    bbox64 *kept = (bbox64 *)malloc((size_t)n * sizeof(bbox64));
    for (int i = 0; i < n; i++) { order[i].index = i; order[i].score = dets[i].score; }
    stable_sort_desc(order, n);
    int nkept = 0, nout = 0;
    for (int k = 0; k < n; k++) {
        if (has_min_score && order[k].score < min_score) break;
        bbox64 b = det_bbox(&dets[order[k].index]);
        int suppressed = 0;
        for (int j = 0; j < nkept; j++)
            if (overlap_similarity(&kept[j], &b) > (double)min_supp) { suppressed = 1; break; }
        if (!suppressed) { out[nout++] = dets[order[k].index]; kept[nkept++] = b; }
    }
    free(order); free(kept);
    return nout;
}

/* ---------------------------------------------------------------- letterbox removal: transform.rs:115-142 */
int orc_letterbox_removal(orc_detection *dets, int n, const double padding[4]) {
    double left = padding[0], top = padding[1], right = padding[2], bottom = padding[3];
    double h_scale = 1.0 - (left + right), v_scale = 1.0 - (top + bottom);
    if (!(h_scale > 2.220446049250313e-16) || !(v_scale > 2.220446049250313e-16)) return -1; /* assert!, 121-122 */
    float l = (float)left, t = (float)top, hs = (float)h_scale, vs = (float)v_scale; /* `as f32`, 141 */
    for (int i = 0; i < n; i++)
        for (int r = 0; r < 8; r++) {
            dets[i].data[2 * r] = (dets[i].data[2 * r] - l) / hs;
            dets[i].data[2 * r + 1] = (dets[i].data[2 * r + 1] - t) / vs;
        }
    return 0;
}

/* ---------------------------------------------------------------- FaceDetection::infer post-network chain: face_detection.rs:259-265 */
int orc_fd_postprocess(const float *raw_boxes, const float *raw_scores, const float *anchors, int n, float scale,
                       const double padding[4], orc_detection *out, int cap) {
    float *boxes = (float *)malloc((size_t)n * 16 * sizeof(float));
    float *scores = (float *)malloc((size_t)n * sizeof(float));
    orc_detection *dets = (orc_detection *)malloc((size_t)n * sizeof(orc_detection));
    orc_detection *pruned = (orc_detection *)malloc((size_t)n * sizeof(orc_detection));
    orc_decode_boxes(raw_boxes, anchors, n, scale, boxes);
    orc_sigmoid_scores(raw_scores, n, scores);
    int nd = orc_convert_to_detections(boxes, scores, n, dets);
    /* MIN_SUPPRESSION_THRESHOLD 0.3, Some(MIN_SCORE 0.5), weighted = true (face_detection.rs:136-139, 263) */
    int np = orc_weighted_nms(dets, nd, 0.3f, 1, 0.5f, pruned);
    static const double zero[4] = {0, 0, 0, 0};
    int rc = orc_letterbox_removal(pruned, np, padding ? padding : zero);
    int m = np < cap ? np : cap;
    if (rc == 0) memcpy(out, pruned, (size_t)m * sizeof(orc_detection)); else np = -1;
    free(boxes); free(scores); free(dets); free(pruned);
    return np;
}

/* ---------------------------------------------------------------- face flag: face_landmark.rs:292-296 */
int orc_face_flag_passes(float raw_flag) { return sigmoid_f32(raw_flag) <= 0.5f ? 0 : 1; }

/* ---------------------------------------------------------------- Rect::scaled: types.rs:62-77 */
static orc_rect rect_scaled(const orc_rect *r, double sw, double sh, int normalize) {
    if ((r->normalized != 0) == (normalize != 0)) return *r;
    double sx = normalize ? 1.0 / sw : sw, sy = normalize ? 1.0 / sh : sh;
    orc_rect o = {r->x_center * sx, r->y_center * sy, r->width * sx, r->height * sy, r->rotation, normalize};
    return o;
}

/* ---------------------------------------------------------------- project_landmarks: transform.rs:351-432 */
void orc_project_landmarks(const float *raw, int n, int tensor_w, int tensor_h, int image_w, int image_h,
                           const double padding[4], const orc_rect *roi, int flip_horizontal, double *out) {
    float wf = (float)tensor_w, hf = (float)tensor_h;
    static const double zero[4] = {0, 0, 0, 0};
    if (!padding) padding = zero;
    int has_pad = !(padding[0] == 0.0 && padding[1] == 0.0 && padding[2] == 0.0 && padding[3] == 0.0); /* 374 */
    double left = padding[0], top = padding[1];
    double h_scale = 1.0 - (padding[0] + padding[2]), v_scale = 1.0 - (padding[1] + padding[3]);
    orc_rect nr = {0};
    float m00 = 0, m01 = 0, m10 = 0, m11 = 0;
    if (roi) {
        nr = rect_scaled(roi, (double)image_w, (double)image_h, 1); /* 390 */
        double s = sin(nr.rotation), c = cos(nr.rotation);
        m00 = (float)c; m01 = (float)s; m10 = (float)(-s); m11 = (float)c; /* matrix rows 0,1; row 2 = [1,1,1], 393 */
    }
    for (int i = 0; i < n; i++) {
        float x = raw[3 * i] / wf, y = raw[3 * i + 1] / hf, z = raw[3 * i + 2] / wf; /* 360-368 */
        if (flip_horizontal) x = x * -1.0f + 1.0f;                                     /* 370-372 */
        if (has_pad) {                                                                 /* 374-387 */
            x = (float)(((double)x - left) / h_scale);
            y = (float)(((double)y - top) / v_scale);
            z = (float)(((double)z - 0.) / h_scale);
        }
        if (roi) { /* 389-423 */
            x = x - 0.5f; y = y - 0.5f; z = z - 0.0f;
            float rz = z * 0.0f;                              /* rotated_points z *= 0.0, 399-402 */
            float rx = x * m00 + y * m10 + rz * 1.0f;         /* rotated = [x y 0]·matrix, 403 */
            float ry = x * m01 + y * m11 + rz * 1.0f;
            float rzz = x * 0.0f + y * 0.0f + rz * 1.0f;
            x = x * 0.f + rx; y = y * 0.f + ry; z = z * 1.f + rzz; /* 405-409 */
            x = (float)((double)x * nr.width + nr.x_center);  /* 411-419 */
            y = (float)((double)y * nr.height + nr.y_center);
            z = (float)((double)z * nr.width + 0.);
        }
        out[3 * i] = (double)x; out[3 * i + 1] = (double)y; out[3 * i + 2] = (double)z; /* Landmark::new, 421/428 */
    }
}

/* ---------------------------------------------------------------- ROI maths: transform.rs:44-109 */
static int bbox_normalized(const double b[4]) { return b[0] >= -1.0 && b[2] < 2.0 && b[1] >= -1.0; } /* types.rs:133-135 */

int orc_bbox_to_roi(const double bbox[4], int image_w, int image_h, const double *kp, double scale_x, double scale_y,
                    int size_mode, orc_rect *out) {
    if (!bbox_normalized(bbox)) return -1; /* transform.rs:51-53 */
    /* select_roi_size, 87-109: abs_box = bbox.absolute(image_size) (types.rs:168-173) */
    double ax0 = bbox[0] * (double)image_w, ay0 = bbox[1] * (double)image_h;
    double ax1 = bbox[2] * (double)image_w, ay1 = bbox[3] * (double)image_h;
    double width = ax1 - ax0, height = ay1 - ay0;
    double iw = (double)image_w, ih = (double)image_h;
    if (size_mode == 1) { double l = fmax(width, height); width = l / iw; height = l / ih; }
    else if (size_mode == 2) { double s = fmin(width, height); width = s / iw; height = s / ih; }
    width *= scale_x; height *= scale_y; /* 58 */
    double bw = bbox[2] - bbox[0], bh = bbox[3] - bbox[1];
    double cx = bbox[0] + bw / 2.0, cy = bbox[1] + bh / 2.0; /* 59-60 */
    double rotation = 0.0;
    if (kp) { /* 62-75 */
        double x0 = kp[0], y0 = kp[1], x1 = kp[2], y1 = kp[3];
        double angle = -atan2(y0 - y1, x1 - x0);
        double two_pi = 2.0 * M_PI;
        rotation = angle - two_pi * floor((angle + M_PI) / two_pi);
    }
    *out = (orc_rect){cx, cy, width, height, rotation, 1};
    return 0;
}

int orc_face_detection_to_roi(const orc_detection *det, int image_w, int image_h, orc_rect *out) {
    /* face_landmark.rs:180-198; Detection::scaled_by_image_size types.rs:237-245 (f32 multiply) */
    float lx = det->data[4] * (float)image_w, ly = det->data[5] * (float)image_h; /* keypoint 0 = LeftEye */
    float rx = det->data[6] * (float)image_w, ry = det->data[7] * (float)image_h; /* keypoint 1 = RightEye */
    double kp[4] = {(double)lx, (double)ly, (double)rx, (double)ry};
    double bbox[4] = {(double)det->data[0], (double)det->data[1], (double)det->data[2], (double)det->data[3]};
    return orc_bbox_to_roi(bbox, image_w, image_h, kp, 1.5, 1.5, 1, out); /* ROI_SCALE, SquareLong */
}

int orc_iris_rois_from_face_landmarks(const double *lm, int image_w, int image_h, orc_rect *left, orc_rect *right) {
    /* iris_landmark.rs:268-292; bbox_from_landmarks transform.rs:146-165 */
    static const int idx[2][2] = {{33, 133}, {362, 263}};
    orc_rect *outs[2] = {left, right};
    for (int e = 0; e < 2; e++) {
        const double *a = lm + 3 * idx[e][0], *b = lm + 3 * idx[e][1];
        double bbox[4] = {fmin(a[0], b[0]), fmin(a[1], b[1]), fmax(a[0], b[0]), fmax(a[1], b[1])};
        double kp[4] = {a[0], a[1], b[0], b[1]};
        if (orc_bbox_to_roi(bbox, image_w, image_h, kp, 2.3, 2.3, 1, outs[e]) != 0) return -1;
    }
    return 0;
}
