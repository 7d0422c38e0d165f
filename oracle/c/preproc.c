/*
 * preproc.c — TEST INFRASTRUCTURE ONLY (see oracle.h).
 *
 * Restatement of transform::image_to_tensor (/root/reference/src/face_detection_lite/transform.rs:188-309) and of
 * the OpenCV primitives it calls through the third-party `opencv` crate 0.93.1 (Cargo.toml:12; OpenCV itself is
 * not under /root/reference): getPerspectiveTransform + warpPerspective(INTER_LINEAR, BORDER_CONSTANT 0),
 * copyMakeBorder, resize(INTER_LINEAR) and flip, for 8UC3 images.  The OpenCV parts follow OpenCV 4.x's
 * published generic (non-IPP) 8-bit fixed-point algorithms:
 *   warp:   source coords in 1/32 px (INTER_BITS = 5), 2x2 weights as 15-bit shorts (INTER_REMAP_COEF_BITS),
 *           dst = (sum w*p + 2^14) >> 15; the (0,0) table entry saturates to {32767,0,0,1}.
 *   resize: 11-bit coefficient pairs (INTER_RESIZE_COEF_BITS), horizontal pass to int, vertical pass
 *           ((b0*(S0>>4))>>16) + ((b1*(S1>>4))>>16) + 2) >> 2; exact 2x shrink switches to INTER_AREA; equal size copies.
 * A real OpenCV build may dispatch to IPP/vendor HALs whose rounding differs by 1 LSB, so parity on this row is
 * stated as +-1 LSB of the u8 image.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "oracle.h"

typedef struct { int w, h; uint8_t *p; } img8; /* 3 channels, row-major */

static img8 img_new(int w, int h) {
    img8 i = {w, h, (uint8_t *)calloc((size_t)(w > 0 ? w : 1) * (size_t)(h > 0 ? h : 1) * 3, 1)};
    return i;
}

/* ---- cv::getPerspectiveTransform: 8x8 linear system in double (OpenCV solves it by SVD; the solution is unique) */
static int solve8(double a[8][9]) {
    for (int c = 0; c < 8; c++) {
        int piv = c;
        for (int r = c + 1; r < 8; r++) if (fabs(a[r][c]) > fabs(a[piv][c])) piv = r;
        if (fabs(a[piv][c]) < 1e-300) return -1;
        if (piv != c) for (int k = 0; k < 9; k++) { double t = a[c][k]; a[c][k] = a[piv][k]; a[piv][k] = t; }
        for (int r = 0; r < 8; r++) {
            if (r == c) continue;
            double f = a[r][c] / a[c][c];
            if (f != 0.0) for (int k = c; k < 9; k++) a[r][k] -= f * a[c][k];
        }
    }
    for (int r = 0; r < 8; r++) a[r][8] /= a[r][r];
    return 0;
}

static int get_perspective_transform(const float src[4][2], const float dst[4][2], double M[9]) {
    double a[8][9];
    memset(a, 0, sizeof(a));
    for (int i = 0; i < 4; i++) {
        double sx = src[i][0], sy = src[i][1], dx = dst[i][0], dy = dst[i][1];
        a[i][0] = a[i + 4][3] = sx;
        a[i][1] = a[i + 4][4] = sy;
        a[i][2] = a[i + 4][5] = 1;
        a[i][6] = -sx * dx; a[i][7] = -sy * dx;
        a[i + 4][6] = -sx * dy; a[i + 4][7] = -sy * dy;
        a[i][8] = dx; a[i + 4][8] = dy;
    }
    if (solve8(a) != 0) return -1;
    for (int i = 0; i < 8; i++) M[i] = a[i][8];
    M[8] = 1.0;
    return 0;
}

static int invert3(const double m[9], double o[9]) {
    double c0 = m[4] * m[8] - m[5] * m[7], c1 = m[5] * m[6] - m[3] * m[8], c2 = m[3] * m[7] - m[4] * m[6];
    double det = m[0] * c0 + m[1] * c1 + m[2] * c2;
    if (det == 0.0) return -1;
    double id = 1.0 / det;
    o[0] = c0 * id; o[1] = (m[2] * m[7] - m[1] * m[8]) * id; o[2] = (m[1] * m[5] - m[2] * m[4]) * id;
    o[3] = c1 * id; o[4] = (m[0] * m[8] - m[2] * m[6]) * id; o[5] = (m[2] * m[3] - m[0] * m[5]) * id;
    o[6] = c2 * id; o[7] = (m[1] * m[6] - m[0] * m[7]) * id; o[8] = (m[0] * m[4] - m[1] * m[3]) * id;
    return 0;
}

static int sat_int(double v) { /* saturate_cast<int>(double) = cvRound (round half to even) */
    if (v < -2147483648.0) v = -2147483648.0;
    if (v > 2147483647.0) v = 2147483647.0;
    return (int)lrint(v);
}

/* ---- cv::warpPerspective(src, M (src->dst), dsize, INTER_LINEAR, BORDER_CONSTANT, 0) for 8UC3 */
static img8 warp_perspective(const img8 *src, const double Mfwd[9], int dw, int dh) {
    img8 dst = img_new(dw, dh);
    double M[9];
    if (invert3(Mfwd, M) != 0) return dst;
    const int BLOCK = 32;
    int bh0 = BLOCK / 2 < dh ? BLOCK / 2 : dh;
    int bw0 = (bh0 > 0 && BLOCK * BLOCK / bh0 < dw) ? BLOCK * BLOCK / bh0 : dw;
    bh0 = (bw0 > 0 && BLOCK * BLOCK / bw0 < dh) ? BLOCK * BLOCK / bw0 : dh;
    for (int y = 0; y < dh; y += bh0)
        for (int x = 0; x < dw; x += bw0) {
            int bw = bw0 < dw - x ? bw0 : dw - x, bh = bh0 < dh - y ? bh0 : dh - y;
            for (int y1 = 0; y1 < bh; y1++) {
                double X0 = M[0] * x + M[1] * (y + y1) + M[2];
                double Y0 = M[3] * x + M[4] * (y + y1) + M[5];
                double W0 = M[6] * x + M[7] * (y + y1) + M[8];
                for (int x1 = 0; x1 < bw; x1++) {
                    double W = W0 + M[6] * x1;
                    W = W != 0.0 ? 32.0 / W : 0.0;
                    int X = sat_int((X0 + M[0] * x1) * W), Y = sat_int((Y0 + M[3] * x1) * W);
                    int sx = X >> 5, sy = Y >> 5, ax = X & 31, ay = Y & 31;
                    /* saturate_cast<short> of the integer coordinates */
                    if (sx < -32768) sx = -32768; if (sx > 32767) sx = 32767;
                    if (sy < -32768) sy = -32768; if (sy > 32767) sy = 32767;
                    int w00 = (32 - ay) * (32 - ax) * 32, w01 = (32 - ay) * ax * 32, w10 = ay * (32 - ax) * 32, w11 = ay * ax * 32;
                    if (ax == 0 && ay == 0) { w00 = 32767; w11 = 1; }
                    uint8_t *d = dst.p + ((size_t)(y + y1) * dw + (x + x1)) * 3;
                    if (sx >= src->w || sx + 1 < 0 || sy >= src->h || sy + 1 < 0) { d[0] = d[1] = d[2] = 0; continue; }
                    for (int c = 0; c < 3; c++) {
                        int p00 = 0, p01 = 0, p10 = 0, p11 = 0;
                        int x0in = sx >= 0 && sx < src->w, x1in = sx + 1 >= 0 && sx + 1 < src->w;
                        int y0in = sy >= 0 && sy < src->h, y1in = sy + 1 >= 0 && sy + 1 < src->h;
                        if (y0in && x0in) p00 = src->p[((size_t)sy * src->w + sx) * 3 + c];
                        if (y0in && x1in) p01 = src->p[((size_t)sy * src->w + sx + 1) * 3 + c];
                        if (y1in && x0in) p10 = src->p[((size_t)(sy + 1) * src->w + sx) * 3 + c];
                        if (y1in && x1in) p11 = src->p[((size_t)(sy + 1) * src->w + sx + 1) * 3 + c];
                        int v = (p00 * w00 + p01 * w01 + p10 * w10 + p11 * w11 + (1 << 14)) >> 15;
                        d[c] = (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v));
                    }
                }
            }
        }
    return dst;
}

/* ---- cv::resize(src, dsize, INTER_LINEAR) for 8UC3 */
static short sat_short_f(float v) {
    long r = lrintf(v);
    return (short)(r < -32768 ? -32768 : (r > 32767 ? 32767 : r));
}

static img8 resize_linear(const img8 *src, int dw, int dh) {
    img8 dst = img_new(dw, dh);
    if (dw == src->w && dh == src->h) { memcpy(dst.p, src->p, (size_t)dw * dh * 3); return dst; }
    double inv_sx = (double)dw / src->w, inv_sy = (double)dh / src->h;
    double scale_x = 1.0 / inv_sx, scale_y = 1.0 / inv_sy;
    int iscale_x = sat_int(scale_x), iscale_y = sat_int(scale_y);
    int area_fast = fabs(scale_x - iscale_x) < 2.220446049250313e-16 && fabs(scale_y - iscale_y) < 2.220446049250313e-16;
    if (area_fast && iscale_x == 2 && iscale_y == 2) { /* INTER_LINEAR -> INTER_AREA fast path */
        for (int y = 0; y < dh; y++)
            for (int x = 0; x < dw; x++)
                for (int c = 0; c < 3; c++) {
                    const uint8_t *s = src->p + ((size_t)(2 * y) * src->w + 2 * x) * 3 + c;
                    dst.p[((size_t)y * dw + x) * 3 + c] = (uint8_t)((s[0] + s[3] + s[(size_t)src->w * 3] + s[(size_t)src->w * 3 + 3] + 2) >> 2);
                }
        return dst;
    }
    int *xofs = (int *)malloc((size_t)dw * sizeof(int)), *yofs = (int *)malloc((size_t)dh * sizeof(int));
    short *alpha = (short *)malloc((size_t)dw * 2 * sizeof(short)), *beta = (short *)malloc((size_t)dh * 2 * sizeof(short));
    for (int dx = 0; dx < dw; dx++) {
        float fx = (float)((dx + 0.5) * scale_x - 0.5);
        int sx = (int)floorf(fx);
        fx -= (float)sx;
        if (sx < 0) { fx = 0; sx = 0; }
        if (sx >= src->w - 1) { fx = 0; sx = src->w - 1; }
        xofs[dx] = sx;
        alpha[2 * dx] = sat_short_f((1.f - fx) * 2048.f);
        alpha[2 * dx + 1] = sat_short_f(fx * 2048.f);
    }
    for (int dy = 0; dy < dh; dy++) {
        float fy = (float)((dy + 0.5) * scale_y - 0.5);
        int sy = (int)floorf(fy);
        fy -= (float)sy;
        yofs[dy] = sy;
        beta[2 * dy] = sat_short_f((1.f - fy) * 2048.f);
        beta[2 * dy + 1] = sat_short_f(fy * 2048.f);
    }
    int *row0 = (int *)malloc((size_t)dw * 3 * sizeof(int)), *row1 = (int *)malloc((size_t)dw * 3 * sizeof(int));
    for (int dy = 0; dy < dh; dy++) {
        int sy0 = yofs[dy], sy1 = yofs[dy] + 1; /* rows clipped to [0, h-1] (resizeGeneric_Invoker: clip(sy, 0, ssize.height)) */
        if (sy0 < 0) sy0 = 0; if (sy0 > src->h - 1) sy0 = src->h - 1;
        if (sy1 < 0) sy1 = 0; if (sy1 > src->h - 1) sy1 = src->h - 1;
        const uint8_t *s0 = src->p + (size_t)sy0 * src->w * 3, *s1 = src->p + (size_t)sy1 * src->w * 3;
        for (int dx = 0; dx < dw; dx++) {
            int sx = xofs[dx], sx1 = sx + 1 < src->w ? sx + 1 : sx;
            int a0 = alpha[2 * dx], a1 = alpha[2 * dx + 1];
            for (int c = 0; c < 3; c++) {
                if (sx + 1 < src->w) {
                    row0[dx * 3 + c] = s0[sx * 3 + c] * a0 + s0[sx1 * 3 + c] * a1;
                    row1[dx * 3 + c] = s1[sx * 3 + c] * a0 + s1[sx1 * 3 + c] * a1;
                } else { /* dx >= xmax: D[dx] = S[sx]*ONE */
                    row0[dx * 3 + c] = s0[sx * 3 + c] * 2048;
                    row1[dx * 3 + c] = s1[sx * 3 + c] * 2048;
                }
            }
        }
        int b0 = beta[2 * dy], b1 = beta[2 * dy + 1];
        uint8_t *d = dst.p + (size_t)dy * dw * 3;
        for (int i = 0; i < dw * 3; i++) {
            int v = (((b0 * (row0[i] >> 4)) >> 16) + ((b1 * (row1[i] >> 4)) >> 16) + 2) >> 2;
            d[i] = (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v));
        }
    }
    free(xofs); free(yofs); free(alpha); free(beta); free(row0); free(row1);
    return dst;
}

static img8 copy_make_border(const img8 *src, int top, int bottom, int left, int right) {
    img8 dst = img_new(src->w + left + right, src->h + top + bottom);
    for (int y = 0; y < src->h; y++)
        memcpy(dst.p + ((size_t)(y + top) * dst.w + left) * 3, src->p + (size_t)y * src->w * 3, (size_t)src->w * 3);
    return dst;
}

/* ---- Rect::points, types.rs:80-96 */
static void rect_points(const orc_rect *r, double pts[4][2]) {
    double x = r->x_center, y = r->y_center, w = r->width / 2.0, h = r->height / 2.0;
    double p[4][2] = {{x - w, y - h}, {x + w, y - h}, {x + w, y + h}, {x - w, y + h}};
    if (r->rotation != 0.0) {
        double s = sin(r->rotation), c = cos(r->rotation);
        for (int i = 0; i < 4; i++) {
            double dx = p[i][0] - x, dy = p[i][1] - y;
            p[i][0] = x + dx * c - dy * s;
            p[i][1] = y + dx * s + dy * c;
        }
    }
    memcpy(pts, p, sizeof(p));
}

int orc_image_to_tensor(const uint8_t *image, int image_w, int image_h, const orc_rect *roi_in, int out_w, int out_h,
                        int keep_aspect_ratio, double range_min, double range_max, int flip_horizontal, float *out,
                        double padding_out[4]) {
    /* transform.rs:190-199: default ROI = whole image; roi.scaled(size, false) (types.rs:62-77) */
    orc_rect roi = roi_in ? *roi_in : (orc_rect){0.5, 0.5, 1.0, 1.0, 0.0, 1};
    if (roi.normalized) {
        roi.x_center *= (double)image_w; roi.y_center *= (double)image_h;
        roi.width *= (double)image_w; roi.height *= (double)image_h;
        roi.normalized = 0;
    }
    /* 203-207: Rect::size truncates absolute sizes (types.rs:52-59) */
    int width = keep_aspect_ratio ? (int)(double)(int)roi.width : out_w;
    int height = keep_aspect_ratio ? (int)(double)(int)roi.height : out_h;
    if (width <= 0 || height <= 0) return -1;
    double pts[4][2];
    rect_points(&roi, pts);
    float src[4][2], dst[4][2] = {{0, 0}, {(float)width, 0}, {(float)width, (float)height}, {0, (float)height}};
    for (int i = 0; i < 4; i++) { src[i][0] = (float)pts[i][0]; src[i][1] = (float)pts[i][1]; } /* 210-213 */
    double M[9];
    if (get_perspective_transform(src, dst, M) != 0) return -1; /* 222 */
    img8 im = {image_w, image_h, (uint8_t *)image};
    img8 roi_image = warp_perspective(&im, M, width, height); /* 225-234 */
    double pad_x = 0.0, pad_y = 0.0;
    if (keep_aspect_ratio) { /* 239-280 */
        double out_aspect = (double)(out_h / out_w); /* integer division, 240 */
        double roi_aspect = roi.height / roi.width;
        int new_width = (int)roi.width, new_height = (int)roi.height;
        if (out_aspect > roi_aspect) { new_height = (int)(roi.width * out_aspect); pad_y = (1.0 - roi_aspect / out_aspect) / 2.0; }
        else { new_width = (int)(roi.height / out_aspect); pad_x = (1.0 - out_aspect / roi_aspect) / 2.0; }
        if (new_width != (int)roi.width || new_height != (int)roi.height) {
            int pad_h = (int)(pad_x * (double)new_width), pad_v = (int)(pad_y * (double)new_height);
            img8 padded = copy_make_border(&roi_image, pad_v, pad_v, pad_h, pad_h);
            img8 resized = resize_linear(&padded, new_width, new_height);
            free(padded.p); free(roi_image.p);
            roi_image = resized;
        }
        img8 resized = resize_linear(&roi_image, out_w, out_h); /* 277 */
        free(roi_image.p);
        roi_image = resized;
    }
    /* 282-286 flip; 292-301 normalise: (pixel as f64 * (max - min) / 255.0 + min) as f32.  The reference's fill loop
     * indexes [y < width][x < height], which is only meaningful for square outputs. */
    for (int y = 0; y < roi_image.h; y++)
        for (int x = 0; x < roi_image.w; x++) {
            int sx = flip_horizontal ? roi_image.w - 1 - x : x;
            for (int c = 0; c < 3; c++) {
                double p = (double)roi_image.p[((size_t)y * roi_image.w + sx) * 3 + c];
                out[((size_t)y * roi_image.w + x) * 3 + c] = (float)(p * (range_max - range_min) / 255.0 + range_min);
            }
        }
    free(roi_image.p);
    if (padding_out) { padding_out[0] = pad_x; padding_out[1] = pad_y; padding_out[2] = pad_x; padding_out[3] = pad_y; }
    return 0;
}
