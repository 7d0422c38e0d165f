/*
 * interp.c — TEST INFRASTRUCTURE ONLY (see oracle.h).
 *
 * Straightforward op-by-op f32 interpreter for the builtin operators that appear in the frozen MediaPipe graphs
 * (SURVEY.md Appendix A/B/C).  It restates what `interpreter.invoke()` does at the reference's call sites
 * (/root/reference/src/face_detection_lite/face_detection.rs:207-235, face_landmark.rs:233-265,
 * iris_landmark.rs:161-203).  The arithmetic itself lives in the third-party TensorFlow-Lite runtime behind the
 * `tflite` crate 0.9.8 (not under /root/reference); semantics follow the published TFLite float reference kernels:
 *   CONV_2D / DEPTHWISE_CONV_2D: total = sum(input * filter) over (ky,kx,ic); out = act(total + bias);
 *       TF "SAME" padding: out = ceil(in/stride), total_pad = max(0,(out-1)*stride + k - in), before = total_pad/2.
 *   MAX_POOL_2D, ADD, RELU, PRELU (x >= 0 ? x : alpha[c]*x), PAD (zero), RESHAPE, CONCATENATION,
 *   RESIZE_BILINEAR (half_pixel_centers), DEQUANTIZE (f16 -> f32 widening), DENSIFY, DEPTH_TO_SPACE.
 * One frame at a time (the reference is single-image); batches are independent frames spread over OpenMP threads.
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "oracle.h"
#include "tfl_reader.h"

static __thread char g_err[512];
const char *orc_last_error(void) { return g_err; }

struct orc_model {
    orc_graph g;
    float **consts;     /* per tensor: f32 constant (dequantized / densified / repacked) or NULL */
    int32_t **iconsts;  /* per tensor: i32 constant or NULL */
    float **repacked;   /* per op: conv filter transposed to [KH][KW][I][O] */
    size_t *elems;      /* per tensor: element count per frame */
    size_t arena_elems; /* sum over activation tensors */
    size_t *offset;     /* per tensor: offset into a frame arena (activations) */
};

static float f16_to_f32(uint16_t h) {
    uint32_t sign = (uint32_t)(h & 0x8000u) << 16, exp = (h >> 10) & 0x1f, man = h & 0x3ffu, bits;
    if (exp == 0) {
        if (man == 0) bits = sign;
        else {
            int e = -1;
            do { e++; man <<= 1; } while (!(man & 0x400u));
            bits = sign | ((uint32_t)(127 - 15 - e) << 23) | ((man & 0x3ffu) << 13);
        }
    } else if (exp == 31) bits = sign | 0x7f800000u | (man << 13);
    else bits = sign | ((exp + 112) << 23) | (man << 13);
    float f;
    memcpy(&f, &bits, 4);
    return f;
}

static size_t tensor_elems(const orc_tensor *t) {
    size_t n = 1;
    for (int d = 0; d < t->rank; d++) n *= (size_t)t->shape[d];
    return n;
}

static void same_pads(int in, int k, int stride, int *before, int *out) {
    int o = (in + stride - 1) / stride;
    int total = (o - 1) * stride + k - in;
    if (total < 0) total = 0;
    *before = total / 2;
    *out = o;
}

static float *const_as_f32(const orc_model *m, int ti) {
    const orc_graph *g = &m->g;
    const orc_tensor *t = &g->tensors[ti];
    if (!t->data) return NULL;
    size_t n = tensor_elems(t);
    float *dst = (float *)calloc(n ? n : 1, sizeof(float));
    if (t->has_sparsity) {
        if (t->type == TT_F32) {
            orc_graph_densify(g, t, (uint8_t *)dst, 4);
        } else if (t->type == TT_F16) {
            uint16_t *tmp = (uint16_t *)calloc(n ? n : 1, 2);
            orc_graph_densify(g, t, (uint8_t *)tmp, 2);
            for (size_t i = 0; i < n; i++) dst[i] = f16_to_f32(tmp[i]);
            free(tmp);
        }
        return dst;
    }
    if (t->type == TT_F32) {
        memcpy(dst, t->data, n * 4 <= t->nbytes ? n * 4 : t->nbytes);
    } else if (t->type == TT_F16) {
        for (size_t i = 0; i < n && 2 * i + 1 < t->nbytes; i++) {
            uint16_t h;
            memcpy(&h, t->data + 2 * i, 2);
            dst[i] = f16_to_f32(h);
        }
    } else {
        free(dst);
        return NULL;
    }
    return dst;
}

int orc_model_load_bytes(const uint8_t *bytes, size_t n, orc_model **out) {
    orc_model *m = (orc_model *)calloc(1, sizeof(orc_model));
    if (orc_graph_parse(bytes, n, &m->g, g_err, sizeof(g_err)) != 0) { free(m); return -1; }
    orc_graph *g = &m->g;
    m->consts = (float **)calloc((size_t)g->ntensors, sizeof(float *));
    m->iconsts = (int32_t **)calloc((size_t)g->ntensors, sizeof(int32_t *));
    m->repacked = (float **)calloc((size_t)g->nops ? (size_t)g->nops : 1, sizeof(float *));
    m->elems = (size_t *)calloc((size_t)g->ntensors, sizeof(size_t));
    m->offset = (size_t *)calloc((size_t)g->ntensors, sizeof(size_t));
    for (int i = 0; i < g->ntensors; i++) {
        orc_tensor *t = &g->tensors[i];
        m->elems[i] = tensor_elems(t);
        if (t->data) {
            if (t->type == TT_I32) {
                m->iconsts[i] = (int32_t *)malloc(t->nbytes);
                memcpy(m->iconsts[i], t->data, t->nbytes);
            } else m->consts[i] = const_as_f32(m, i);
        }
    }
    /* DEQUANTIZE / DENSIFY of constants are folded at load: their outputs become constants. */
    for (int i = 0; i < g->nops; i++) {
        orc_op *op = &g->ops[i];
        if ((op->code == OP_DEQUANTIZE || op->code == OP_DENSIFY) && m->consts[op->in[0]]) {
            size_t n2 = m->elems[op->out[0]];
            m->consts[op->out[0]] = (float *)malloc((n2 ? n2 : 1) * 4);
            memcpy(m->consts[op->out[0]], m->consts[op->in[0]], n2 * 4);
        }
    }
    for (int i = 0; i < g->nops; i++) {
        orc_op *op = &g->ops[i];
        if (op->code == OP_CONV_2D) {
            const orc_tensor *w = &g->tensors[op->in[1]];
            const float *src = m->consts[op->in[1]];
            if (!src || w->rank != 4) { snprintf(g_err, sizeof(g_err), "op %d: conv filter not constant", i); orc_model_free(m); return -1; }
            int O = w->shape[0], KH = w->shape[1], KW = w->shape[2], I = w->shape[3];
            float *r = (float *)malloc((size_t)O * KH * KW * I * 4);
            for (int o = 0; o < O; o++)
                for (int ky = 0; ky < KH; ky++)
                    for (int kx = 0; kx < KW; kx++)
                        for (int c = 0; c < I; c++)
                            r[(((size_t)ky * KW + kx) * I + c) * O + o] = src[(((size_t)o * KH + ky) * KW + kx) * I + c];
            m->repacked[i] = r;
        }
    }
    size_t off = 0;
    for (int i = 0; i < g->ntensors; i++) {
        if (!m->consts[i] && !m->iconsts[i]) { m->offset[i] = off; off += (m->elems[i] + 15) & ~(size_t)15; }
    }
    m->arena_elems = off;
    if (g->ninputs < 1 || g->tensors[g->inputs[0]].rank != 4) {
        snprintf(g_err, sizeof(g_err), "model input must be rank-4 NHWC");
        orc_model_free(m);
        return -1;
    }
    *out = m;
    return 0;
}

int orc_model_load(const char *path, orc_model **out) {
    FILE *fp = fopen(path, "rb");
    if (!fp) { snprintf(g_err, sizeof(g_err), "cannot open %s", path); return -1; }
    fseek(fp, 0, SEEK_END);
    long n = ftell(fp);
    fseek(fp, 0, SEEK_SET);
    uint8_t *buf = (uint8_t *)malloc((size_t)n);
    if (fread(buf, 1, (size_t)n, fp) != (size_t)n) { fclose(fp); free(buf); snprintf(g_err, sizeof(g_err), "short read %s", path); return -1; }
    fclose(fp);
    int rc = orc_model_load_bytes(buf, (size_t)n, out);
    free(buf);
    return rc;
}

void orc_model_free(orc_model *m) {
    if (!m) return;
    for (int i = 0; i < m->g.ntensors; i++) { if (m->consts) free(m->consts[i]); if (m->iconsts) free(m->iconsts[i]); }
    for (int i = 0; i < m->g.nops; i++) if (m->repacked) free(m->repacked[i]);
    free(m->consts); free(m->iconsts); free(m->repacked); free(m->elems); free(m->offset);
    orc_graph_free(&m->g);
    free(m);
}

void orc_model_input_dims(const orc_model *m, int dims[4]) {
    const orc_tensor *t = &m->g.tensors[m->g.inputs[0]];
    for (int d = 0; d < 4; d++) dims[d] = t->shape[d];
}
int orc_model_num_outputs(const orc_model *m) { return m->g.noutputs; }
int orc_model_output_dims(const orc_model *m, int idx, int dims[4]) {
    const orc_tensor *t = &m->g.tensors[m->g.outputs[idx]];
    for (int d = 0; d < 4; d++) dims[d] = d < t->rank ? t->shape[d] : 1;
    return t->rank;
}
size_t orc_model_output_elems(const orc_model *m, int idx) { return m->elems[m->g.outputs[idx]]; }
int orc_model_num_tensors(const orc_model *m) { return m->g.ntensors; }
int orc_model_num_ops(const orc_model *m) { return m->g.nops; }

static inline float apply_act(float v, int act) {
    if (act == 1) return v > 0.f ? v : 0.f;
    if (act == 3) return v < 0.f ? 0.f : (v > 6.f ? 6.f : v);
    return v;
}

#define MAX_OC 2048

static void conv2d(const float *in, int H, int W, int C, const float *wt, int KH, int KW, int O, const float *bias,
                   int sh, int sw, int padding, int act, float *out, int Ho, int Wo) {
    int pt = 0, pl = 0, ho, wo;
    if (padding == 0) { same_pads(H, KH, sh, &pt, &ho); same_pads(W, KW, sw, &pl, &wo); }
    (void)ho; (void)wo;
    float acc[MAX_OC];
    for (int oy = 0; oy < Ho; oy++)
        for (int ox = 0; ox < Wo; ox++) {
            for (int o = 0; o < O; o++) acc[o] = 0.f;
            for (int ky = 0; ky < KH; ky++) {
                int iy = oy * sh - pt + ky;
                if (iy < 0 || iy >= H) continue;
                for (int kx = 0; kx < KW; kx++) {
                    int ix = ox * sw - pl + kx;
                    if (ix < 0 || ix >= W) continue;
                    const float *ip = in + ((size_t)iy * W + ix) * C;
                    const float *wp = wt + ((size_t)ky * KW + kx) * C * O;
                    for (int c = 0; c < C; c++) {
                        float x = ip[c];
                        const float *wr = wp + (size_t)c * O;
                        for (int o = 0; o < O; o++) acc[o] += x * wr[o];
                    }
                }
            }
            float *op = out + ((size_t)oy * Wo + ox) * O;
            for (int o = 0; o < O; o++) op[o] = apply_act(acc[o] + (bias ? bias[o] : 0.f), act);
        }
}

static void dwconv2d(const float *in, int H, int W, int C, const float *w, int KH, int KW, const float *bias, int sh,
                     int sw, int padding, int act, float *out, int Ho, int Wo) {
    int pt = 0, pl = 0, ho, wo;
    if (padding == 0) { same_pads(H, KH, sh, &pt, &ho); same_pads(W, KW, sw, &pl, &wo); }
    (void)ho; (void)wo;
    float acc[MAX_OC];
    for (int oy = 0; oy < Ho; oy++)
        for (int ox = 0; ox < Wo; ox++) {
            for (int c = 0; c < C; c++) acc[c] = 0.f;
            for (int ky = 0; ky < KH; ky++) {
                int iy = oy * sh - pt + ky;
                if (iy < 0 || iy >= H) continue;
                for (int kx = 0; kx < KW; kx++) {
                    int ix = ox * sw - pl + kx;
                    if (ix < 0 || ix >= W) continue;
                    const float *ip = in + ((size_t)iy * W + ix) * C;
                    const float *wp = w + ((size_t)ky * KW + kx) * C;
                    for (int c = 0; c < C; c++) acc[c] += ip[c] * wp[c];
                }
            }
            float *op = out + ((size_t)oy * Wo + ox) * C;
            for (int c = 0; c < C; c++) op[c] = apply_act(acc[c] + (bias ? bias[c] : 0.f), act);
        }
}

static void maxpool2d(const float *in, int H, int W, int C, int FH, int FW, int sh, int sw, int padding, float *out,
                      int Ho, int Wo) {
    int pt = 0, pl = 0, ho, wo;
    if (padding == 0) { same_pads(H, FH, sh, &pt, &ho); same_pads(W, FW, sw, &pl, &wo); }
    (void)ho; (void)wo;
    for (int oy = 0; oy < Ho; oy++)
        for (int ox = 0; ox < Wo; ox++) {
            float *op = out + ((size_t)oy * Wo + ox) * C;
            for (int c = 0; c < C; c++) op[c] = -INFINITY;
            for (int ky = 0; ky < FH; ky++) {
                int iy = oy * sh - pt + ky;
                if (iy < 0 || iy >= H) continue;
                for (int kx = 0; kx < FW; kx++) {
                    int ix = ox * sw - pl + kx;
                    if (ix < 0 || ix >= W) continue;
                    const float *ip = in + ((size_t)iy * W + ix) * C;
                    for (int c = 0; c < C; c++) if (ip[c] > op[c]) op[c] = ip[c];
                }
            }
        }
}

/* TFLite reference ResizeBilinear with half_pixel_centers (ComputeInterpolationValues). */
static void resize_bilinear(const float *in, int H, int W, int C, float *out, int Ho, int Wo, int align_corners,
                            int half_pixel) {
    float hs = (align_corners && Ho > 1) ? (float)(H - 1) / (float)(Ho - 1) : (float)H / (float)Ho;
    float ws = (align_corners && Wo > 1) ? (float)(W - 1) / (float)(Wo - 1) : (float)W / (float)Wo;
    for (int y = 0; y < Ho; y++) {
        float iy = half_pixel ? ((float)y + 0.5f) * hs - 0.5f : (float)y * hs;
        float fy = floorf(iy);
        int y0 = (int)fy < 0 ? 0 : (int)fy;
        int y1 = (int)ceilf(iy) > H - 1 ? H - 1 : (int)ceilf(iy);
        float dy = iy - (float)y0; /* TFLite: input_y - y0 with y0 the clamped lower bound */
        for (int x = 0; x < Wo; x++) {
            float ixf = half_pixel ? ((float)x + 0.5f) * ws - 0.5f : (float)x * ws;
            float fx = floorf(ixf);
            int x0 = (int)fx < 0 ? 0 : (int)fx;
            int x1 = (int)ceilf(ixf) > W - 1 ? W - 1 : (int)ceilf(ixf);
            float dx = ixf - (float)x0;
            const float *p00 = in + ((size_t)y0 * W + x0) * C, *p01 = in + ((size_t)y0 * W + x1) * C;
            const float *p10 = in + ((size_t)y1 * W + x0) * C, *p11 = in + ((size_t)y1 * W + x1) * C;
            float *op = out + ((size_t)y * Wo + x) * C;
            for (int c = 0; c < C; c++)
                op[c] = p00[c] * (1 - dy) * (1 - dx) + p10[c] * dy * (1 - dx) + p01[c] * (1 - dy) * dx + p11[c] * dy * dx;
        }
    }
}

/* Run one frame. arena holds every activation tensor at m->offset[]. Returns 0 or -1. */
static int run_frame(const orc_model *m, const float *in, float *arena, char *err, size_t errlen) {
    const orc_graph *g = &m->g;
#define T(i) (m->consts[i] ? m->consts[i] : arena + m->offset[i])
    memcpy(arena + m->offset[g->inputs[0]], in, m->elems[g->inputs[0]] * 4);
    for (int oi = 0; oi < g->nops; oi++) {
        const orc_op *op = &g->ops[oi];
        int o0 = op->out[0];
        const orc_tensor *to = &g->tensors[o0];
        switch (op->code) {
        case OP_DEQUANTIZE:
        case OP_DENSIFY:
            if (!m->consts[o0]) { snprintf(err, errlen, "op %d: runtime DEQUANTIZE/DENSIFY unsupported", oi); return -1; }
            break;
        case OP_CONV_2D: {
            const orc_tensor *ti = &g->tensors[op->in[0]], *tw = &g->tensors[op->in[1]];
            if (tw->shape[0] > MAX_OC) { snprintf(err, errlen, "op %d: too many output channels", oi); return -1; }
            const float *bias = (op->nin > 2 && op->in[2] >= 0) ? m->consts[op->in[2]] : NULL;
            conv2d(T(op->in[0]), ti->shape[1], ti->shape[2], ti->shape[3], m->repacked[oi], tw->shape[1], tw->shape[2],
                   tw->shape[0], bias, op->stride_h, op->stride_w, op->padding, op->act, T(o0), to->shape[1], to->shape[2]);
            break;
        }
        case OP_DEPTHWISE_CONV_2D: {
            const orc_tensor *ti = &g->tensors[op->in[0]], *tw = &g->tensors[op->in[1]];
            if (op->depth_multiplier != 1 || tw->shape[3] > MAX_OC) { snprintf(err, errlen, "op %d: depthwise config unsupported", oi); return -1; }
            const float *bias = (op->nin > 2 && op->in[2] >= 0) ? m->consts[op->in[2]] : NULL;
            dwconv2d(T(op->in[0]), ti->shape[1], ti->shape[2], ti->shape[3], m->consts[op->in[1]], tw->shape[1], tw->shape[2],
                     bias, op->stride_h, op->stride_w, op->padding, op->act, T(o0), to->shape[1], to->shape[2]);
            break;
        }
        case OP_MAX_POOL_2D: {
            const orc_tensor *ti = &g->tensors[op->in[0]];
            maxpool2d(T(op->in[0]), ti->shape[1], ti->shape[2], ti->shape[3], op->filter_h, op->filter_w, op->stride_h,
                      op->stride_w, op->padding, T(o0), to->shape[1], to->shape[2]);
            break;
        }
        case OP_ADD: {
            const float *a = T(op->in[0]), *b = T(op->in[1]);
            float *d = T(o0);
            size_t n = m->elems[o0];
            if (m->elems[op->in[0]] != n || m->elems[op->in[1]] != n) { snprintf(err, errlen, "op %d: broadcasting ADD unsupported", oi); return -1; }
            for (size_t i = 0; i < n; i++) d[i] = apply_act(a[i] + b[i], op->act);
            break;
        }
        case OP_RELU: {
            const float *a = T(op->in[0]);
            float *d = T(o0);
            size_t n = m->elems[o0];
            for (size_t i = 0; i < n; i++) d[i] = a[i] > 0.f ? a[i] : 0.f;
            break;
        }
        case OP_PRELU: {
            const float *a = T(op->in[0]), *al = m->consts[op->in[1]];
            float *d = T(o0);
            size_t n = m->elems[o0], C = m->elems[op->in[1]];
            if (!al || C != (size_t)to->shape[to->rank - 1]) { snprintf(err, errlen, "op %d: PRELU alpha must be per-channel constant", oi); return -1; }
            for (size_t i = 0; i < n; i++) d[i] = a[i] >= 0.f ? a[i] : al[i % C] * a[i];
            break;
        }
        case OP_PAD: {
            const orc_tensor *ti = &g->tensors[op->in[0]];
            const int32_t *p = m->iconsts[op->in[1]];
            if (!p || ti->rank != 4) { snprintf(err, errlen, "op %d: PAD needs constant paddings, rank 4", oi); return -1; }
            float *d = T(o0);
            const float *a = T(op->in[0]);
            memset(d, 0, m->elems[o0] * 4);
            int H = ti->shape[1], W = ti->shape[2], C = ti->shape[3];
            int Wo = to->shape[2], Co = to->shape[3];
            for (int y = 0; y < H; y++)
                for (int x = 0; x < W; x++)
                    memcpy(d + (((size_t)(y + p[2]) * Wo) + (x + p[4])) * Co + p[6], a + ((size_t)y * W + x) * C, (size_t)C * 4);
            break;
        }
        case OP_RESHAPE:
            memcpy(T(o0), T(op->in[0]), m->elems[o0] * 4);
            break;
        case OP_CONCATENATION: {
            /* outer = prod(shape[:axis]); each input contributes inner_i = prod(shape_i[axis:]) contiguous floats */
            int axis = op->axis < 0 ? op->axis + to->rank : op->axis;
            size_t outer = 1;
            for (int d = 0; d < axis; d++) outer *= (size_t)to->shape[d];
            size_t inner_out = m->elems[o0] / outer, off = 0;
            float *d = T(o0);
            for (int k = 0; k < op->nin; k++) {
                size_t inner = m->elems[op->in[k]] / outer;
                const float *a = T(op->in[k]);
                for (size_t r = 0; r < outer; r++) memcpy(d + r * inner_out + off, a + r * inner, inner * 4);
                off += inner;
            }
            break;
        }
        case OP_RESIZE_BILINEAR: {
            const orc_tensor *ti = &g->tensors[op->in[0]];
            resize_bilinear(T(op->in[0]), ti->shape[1], ti->shape[2], ti->shape[3], T(o0), to->shape[1], to->shape[2],
                            op->align_corners, op->half_pixel_centers);
            break;
        }
        case OP_DEPTH_TO_SPACE: {
            const orc_tensor *ti = &g->tensors[op->in[0]];
            int H = ti->shape[1], W = ti->shape[2], C = ti->shape[3], bs = op->block_size, Co = C / (bs * bs);
            const float *a = T(op->in[0]);
            float *d = T(o0);
            for (int y = 0; y < H; y++)
                for (int x = 0; x < W; x++)
                    for (int by = 0; by < bs; by++)
                        for (int bx = 0; bx < bs; bx++)
                            memcpy(d + (((size_t)(y * bs + by) * (W * bs)) + (x * bs + bx)) * Co,
                                   a + ((size_t)y * W + x) * C + ((size_t)by * bs + bx) * Co, (size_t)Co * 4);
            break;
        }
        default:
            snprintf(err, errlen, "op %d: builtin code %d unsupported", oi, op->code);
            return -1;
        }
    }
#undef T
    return 0;
}

int orc_model_run(const orc_model *m, const float *in, int batch, float *const *outs, int nthreads) {
    const orc_graph *g = &m->g;
    size_t in_elems = m->elems[g->inputs[0]];
    int failed = 0;
    char err[256] = {0};
    if (nthreads < 1) nthreads = 1;
#pragma omp parallel num_threads(nthreads)
    {
        /* per-thread arena kept across calls: re-faulting ~60 MB per thread per call dominated multi-threaded runs */
        static __thread float *t_arena = NULL;
        static __thread size_t t_cap = 0;
        size_t need = (m->arena_elems ? m->arena_elems : 1);
        if (need > t_cap) { free(t_arena); t_arena = (float *)malloc(need * sizeof(float)); t_cap = need; }
        float *arena = t_arena;
        char lerr[256];
#pragma omp for schedule(dynamic, 1)
        for (int b = 0; b < batch; b++) {
            if (run_frame(m, in + (size_t)b * in_elems, arena, lerr, sizeof(lerr)) != 0) {
#pragma omp critical
                { failed = 1; memcpy(err, lerr, sizeof(err)); }
                continue;
            }
            for (int k = 0; k < g->noutputs; k++) {
                int ti = g->outputs[k];
                const float *src = m->consts[ti] ? m->consts[ti] : arena + m->offset[ti];
                memcpy(outs[k] + (size_t)b * m->elems[ti], src, m->elems[ti] * 4);
            }
        }
    }
    if (failed) { snprintf(g_err, sizeof(g_err), "%s", err); return -1; }
    return 0;
}

long orc_model_run_tensor(const orc_model *m, const float *in, int tensor_index, float *dst, size_t cap) {
    if (tensor_index < 0 || tensor_index >= m->g.ntensors) return -1;
    float *arena = (float *)malloc((m->arena_elems ? m->arena_elems : 1) * sizeof(float));
    char err[256];
    if (run_frame(m, in, arena, err, sizeof(err)) != 0) { free(arena); snprintf(g_err, sizeof(g_err), "%s", err); return -1; }
    size_t n = m->elems[tensor_index];
    if (n > cap) n = cap;
    const float *src = m->consts[tensor_index] ? m->consts[tensor_index] : arena + m->offset[tensor_index];
    memcpy(dst, src, n * 4);
    free(arena);
    return (long)n;
}
