/*
 * tfl_reader.c — TEST INFRASTRUCTURE ONLY (see oracle.h).
 *
 * Dependency-free reader for TFL3 flatbuffers.  Stands in for `FlatBufferModel::build_from_file`
 * of the third-party `tflite` crate 0.9.8 (call sites /root/reference/src/face_detection_lite/
 * face_detection.rs:188, face_landmark.rs:216, iris_landmark.rs:150).  Field numbers follow the
 * published TFLite schema v3.
 */
#include "tfl_reader.h"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>

typedef struct {
    const uint8_t *b;
    size_t n;
    int bad;
} fb_t;

static uint32_t rd_u32(fb_t *f, size_t o) {
    if (o + 4 > f->n) { f->bad = 1; return 0; }
    uint32_t v; memcpy(&v, f->b + o, 4); return v;
}
static int32_t rd_i32(fb_t *f, size_t o) { return (int32_t)rd_u32(f, o); }
static uint16_t rd_u16(fb_t *f, size_t o) {
    if (o + 2 > f->n) { f->bad = 1; return 0; }
    uint16_t v; memcpy(&v, f->b + o, 2); return v;
}
static uint8_t rd_u8(fb_t *f, size_t o) {
    if (o + 1 > f->n) { f->bad = 1; return 0; }
    return f->b[o];
}
static size_t indirect(fb_t *f, size_t o) { return o + rd_u32(f, o); }

/* absolute offset of field k of a table, 0 when absent */
static size_t field(fb_t *f, size_t table, int k) {
    size_t vt = table - (size_t)(int64_t)rd_i32(f, table);
    uint16_t vsize = rd_u16(f, vt);
    int slot = 4 + 2 * k;
    if (slot + 2 > vsize) return 0;
    uint16_t off = rd_u16(f, vt + slot);
    return off ? table + off : 0;
}
static size_t vec(fb_t *f, size_t table, int k, uint32_t *len) {
    size_t fo = field(f, table, k);
    if (!fo) { *len = 0; return 0; }
    size_t v = indirect(f, fo);
    *len = rd_u32(f, v);
    return v + 4;
}
static int32_t scalar_i32(fb_t *f, size_t table, int k, int32_t def) {
    size_t fo = field(f, table, k);
    return fo ? rd_i32(f, fo) : def;
}
static int32_t scalar_i8(fb_t *f, size_t table, int k, int32_t def) {
    size_t fo = field(f, table, k);
    return fo ? (int8_t)rd_u8(f, fo) : def;
}

int orc_graph_parse(const uint8_t *bytes, size_t n, orc_graph *g, char *err, size_t errlen) {
    memset(g, 0, sizeof(*g));
    if (n < 8 || memcmp(bytes + 4, "TFL3", 4) != 0) {
        snprintf(err, errlen, "not a TFL3 flatbuffer");
        return -1;
    }
    g->blob = (uint8_t *)malloc(n);
    memcpy(g->blob, bytes, n);
    g->nblob = n;
    fb_t fbs = {g->blob, n, 0};
    fb_t *f = &fbs;
    size_t model = indirect(f, 0);

    uint32_t ncodes;
    size_t codes = vec(f, model, 1, &ncodes);
    int *opcodes = (int *)calloc(ncodes ? ncodes : 1, sizeof(int));
    for (uint32_t i = 0; i < ncodes; i++) {
        size_t oc = indirect(f, codes + 4 * i);
        int a = scalar_i8(f, oc, 0, 0), b = scalar_i32(f, oc, 3, 0);
        opcodes[i] = a > b ? a : b;
    }
    uint32_t nbuf;
    size_t bufs = vec(f, model, 4, &nbuf);
    uint32_t nsg;
    size_t sgs = vec(f, model, 2, &nsg);
    if (nsg < 1) {
        snprintf(err, errlen, "model has no subgraph");
        free(opcodes);
        return -1;
    }
    size_t sg = indirect(f, sgs);

    uint32_t nt;
    size_t ts = vec(f, sg, 0, &nt);
    g->ntensors = (int)nt;
    g->tensors = (orc_tensor *)calloc(nt ? nt : 1, sizeof(orc_tensor));
    for (uint32_t i = 0; i < nt; i++) {
        size_t tt = indirect(f, ts + 4 * i);
        orc_tensor *t = &g->tensors[i];
        uint32_t rank;
        size_t sh = vec(f, tt, 0, &rank);
        if (rank > 6) { snprintf(err, errlen, "tensor %u rank %u unsupported", i, rank); free(opcodes); return -1; }
        t->rank = (int)rank;
        for (uint32_t d = 0; d < rank; d++) t->shape[d] = rd_i32(f, sh + 4 * d);
        t->type = scalar_i8(f, tt, 1, 0);
        uint32_t bidx = (uint32_t)scalar_i32(f, tt, 2, 0);
        if (bidx < nbuf) {
            size_t bt = indirect(f, bufs + 4 * bidx);
            uint32_t blen;
            size_t bd = vec(f, bt, 0, &blen);
            if (blen) { t->data = g->blob + bd; t->nbytes = blen; }
        }
        size_t spf = field(f, tt, 6);
        if (spf) { t->has_sparsity = 1; t->sparsity_off = indirect(f, spf); }
    }

    uint32_t nin, nout;
    size_t ins = vec(f, sg, 1, &nin), outs = vec(f, sg, 2, &nout);
    g->ninputs = (int)(nin > 4 ? 4 : nin);
    g->noutputs = (int)(nout > 8 ? 8 : nout);
    for (int i = 0; i < g->ninputs; i++) g->inputs[i] = rd_i32(f, ins + 4 * i);
    for (int i = 0; i < g->noutputs; i++) g->outputs[i] = rd_i32(f, outs + 4 * i);

    uint32_t no;
    size_t os = vec(f, sg, 3, &no);
    g->nops = (int)no;
    g->ops = (orc_op *)calloc(no ? no : 1, sizeof(orc_op));
    for (uint32_t i = 0; i < no; i++) {
        size_t ot = indirect(f, os + 4 * i);
        orc_op *op = &g->ops[i];
        uint32_t oci = (uint32_t)scalar_i32(f, ot, 0, 0);
        op->code = oci < ncodes ? opcodes[oci] : -1;
        uint32_t ni, nn;
        size_t iv = vec(f, ot, 1, &ni), ov = vec(f, ot, 2, &nn);
        if (ni > 8 || nn > 4) { snprintf(err, errlen, "op %u arity unsupported", i); free(opcodes); return -1; }
        op->nin = (int)ni; op->nout = (int)nn;
        for (uint32_t k = 0; k < ni; k++) op->in[k] = rd_i32(f, iv + 4 * k);
        for (uint32_t k = 0; k < nn; k++) op->out[k] = rd_i32(f, ov + 4 * k);
        op->stride_w = op->stride_h = 1;
        size_t of = field(f, ot, 4);
        size_t o = of ? indirect(f, of) : 0;
        if (o) {
            switch (op->code) {
            case OP_CONV_2D:
                op->padding = scalar_i8(f, o, 0, 0); op->stride_w = scalar_i32(f, o, 1, 1);
                op->stride_h = scalar_i32(f, o, 2, 1); op->act = scalar_i8(f, o, 3, 0);
                break;
            case OP_DEPTHWISE_CONV_2D:
                op->padding = scalar_i8(f, o, 0, 0); op->stride_w = scalar_i32(f, o, 1, 1);
                op->stride_h = scalar_i32(f, o, 2, 1); op->depth_multiplier = scalar_i32(f, o, 3, 1);
                op->act = scalar_i8(f, o, 4, 0);
                break;
            case OP_MAX_POOL_2D:
                op->padding = scalar_i8(f, o, 0, 0); op->stride_w = scalar_i32(f, o, 1, 1);
                op->stride_h = scalar_i32(f, o, 2, 1); op->filter_w = scalar_i32(f, o, 3, 1);
                op->filter_h = scalar_i32(f, o, 4, 1); op->act = scalar_i8(f, o, 5, 0);
                break;
            case OP_ADD: op->act = scalar_i8(f, o, 0, 0); break;
            case OP_CONCATENATION: op->axis = scalar_i32(f, o, 0, 0); op->act = scalar_i8(f, o, 1, 0); break;
            case OP_RESIZE_BILINEAR:
                op->align_corners = scalar_i8(f, o, 2, 0); op->half_pixel_centers = scalar_i8(f, o, 3, 0);
                break;
            case OP_DEPTH_TO_SPACE: op->block_size = scalar_i32(f, o, 0, 1); break;
            default: break;
            }
        }
    }
    free(opcodes);
    if (f->bad) {
        snprintf(err, errlen, "truncated or corrupt flatbuffer");
        orc_graph_free(g);
        return -1;
    }
    return 0;
}

void orc_graph_free(orc_graph *g) {
    free(g->blob); free(g->tensors); free(g->ops);
    memset(g, 0, sizeof(*g));
}

/* ---- DENSIFY: TFLite sparse format converter (published algorithm): walk dim_metadata in traversal order. ---- */
typedef struct {
    int format, dense_size;
    int seg_type, idx_type;         /* 0 none, 1 i32, 2 u16, 3 u8 */
    size_t seg, idx;                /* data offsets */
    uint32_t nseg, nidx;
} dim_meta;

static int64_t sv_get(fb_t *f, int type, size_t base, uint32_t i) {
    switch (type) {
    case 1: return rd_i32(f, base + 4 * (size_t)i);
    case 2: return rd_u16(f, base + 2 * (size_t)i);
    case 3: return rd_u8(f, base + i);
    default: return 0;
    }
}

typedef struct {
    fb_t *f;
    int nlev, nd, nblock;
    int order[12], block_map[6], block_size[6];
    dim_meta dm[12];
    int idx[12];
    const orc_tensor *t;
    const uint8_t *values;
    uint8_t *dst;
    size_t esize;
} dens_ctx;

static void dens_emit(dens_ctx *c, int64_t vidx) {
    int coord[12];
    for (int lv = 0; lv < c->nlev; lv++) coord[c->order[lv]] = c->idx[lv];
    int orig[6];
    for (int d = 0; d < c->nd; d++) orig[d] = coord[d];
    for (int b = 0; b < c->nblock; b++) orig[c->block_map[b]] = orig[c->block_map[b]] * c->block_size[b] + coord[c->nd + b];
    size_t lin = 0;
    for (int d = 0; d < c->nd; d++) lin = lin * (size_t)c->t->shape[d] + (size_t)orig[d];
    memcpy(c->dst + lin * c->esize, c->values + (size_t)vidx * c->esize, c->esize);
}

static void dens_rec(dens_ctx *c, int level, int64_t prev) {
    if (level == c->nlev) { dens_emit(c, prev); return; }
    dim_meta *d = &c->dm[level];
    if (d->format == 0) {
        for (int i = 0; i < d->dense_size; i++) { c->idx[level] = i; dens_rec(c, level + 1, prev * d->dense_size + i); }
    } else {
        int64_t s = sv_get(c->f, d->seg_type, d->seg, (uint32_t)prev), e = sv_get(c->f, d->seg_type, d->seg, (uint32_t)prev + 1);
        for (int64_t p = s; p < e; p++) { c->idx[level] = (int)sv_get(c->f, d->idx_type, d->idx, (uint32_t)p); dens_rec(c, level + 1, p); }
    }
}

int orc_graph_densify(const orc_graph *g, const orc_tensor *t, uint8_t *dst, size_t esize) {
    if (!t->has_sparsity || !t->data) return -1;
    fb_t fbs = {g->blob, g->nblob, 0};
    dens_ctx c;
    memset(&c, 0, sizeof(c));
    c.f = &fbs; c.t = t; c.values = t->data; c.dst = dst; c.esize = esize; c.nd = t->rank;
    size_t sp = t->sparsity_off;
    uint32_t n;
    size_t v = vec(c.f, sp, 0, &n);
    if (n > 12) return -1;
    c.nlev = (int)n;
    for (uint32_t i = 0; i < n; i++) c.order[i] = rd_i32(c.f, v + 4 * i);
    v = vec(c.f, sp, 1, &n);
    if (n > 6) return -1;
    c.nblock = (int)n;
    for (uint32_t i = 0; i < n; i++) c.block_map[i] = rd_i32(c.f, v + 4 * i);
    v = vec(c.f, sp, 2, &n);
    if ((int)n != c.nlev) return -1;
    for (uint32_t i = 0; i < n; i++) {
        size_t dmt = indirect(c.f, v + 4 * i);
        dim_meta *d = &c.dm[i];
        d->format = scalar_i8(c.f, dmt, 0, 0);
        d->dense_size = scalar_i32(c.f, dmt, 1, 0);
        d->seg_type = (uint8_t)scalar_i8(c.f, dmt, 2, 0);
        d->idx_type = (uint8_t)scalar_i8(c.f, dmt, 4, 0);
        if (d->seg_type) { size_t tb = indirect(c.f, field(c.f, dmt, 3)); d->seg = vec(c.f, tb, 0, &d->nseg); }
        if (d->idx_type) { size_t tb = indirect(c.f, field(c.f, dmt, 5)); d->idx = vec(c.f, tb, 0, &d->nidx); }
    }
    for (int b = 0; b < c.nblock; b++) c.block_size[b] = c.dm[c.nd + b].dense_size;
    dens_rec(&c, 0, 0);
    return fbs.bad ? -1 : 0;
}
