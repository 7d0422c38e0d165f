/*
 * jpeg.c — TEST INFRASTRUCTURE ONLY (see oracle.h).
 *
 * CPU restatement of what `convert_image_to_mat` (/root/reference/src/face_detection_lite/utils.rs:8-21) computes:
 * `imdecode(bytes, IMREAD_COLOR)` followed by `cvt_color(BGR2RGB)`, i.e. the decoded picture as 8-bit RGB.
 * The arithmetic lives in a dependency that is absent from /root/reference: the `opencv` crate 0.93.1
 * (Cargo.lock:1001-1004) -> system OpenCV 4.x -> its bundled libjpeg-turbo (libjpeg 6b API: ISLOW integer IDCT, "fancy"
 * triangle-filter chroma upsampling, 16-bit fixed-point YCbCr -> RGB).  This file restates those published algorithms
 * (ITU-T T.81 for the bit stream; jidctint.c / jdsample.c / jdcolor.c for the sample arithmetic) for baseline / extended
 * sequential and (round 3) progressive 8-bit Huffman JPEGs with 1 or 3 components and h1v1 / h2v1 / h2v2 chroma sampling — the
 * reference's test_data/ images are all three baseline, 4:2:0; `imdecode` takes progressive files the same way (jdphuff.c).
 *
 * PINNED against libjpeg-turbo itself as present in this image (Pillow's decoder, libjpeg-turbo with the 6.2 API, the same
 * library family OpenCV bundles): bit-exact on the reference's three test JPEGs (tests/test_jpeg.py).
 *
 * Deliberately simple: one bit at a time, canonical-code search, no lookup tables, whole planes in memory.
 */
#include <stdlib.h>
#include <string.h>

#include "oracle.h"

typedef struct {
    const uint8_t *p, *end;
    uint32_t bitbuf;
    int bits;
    int marker;  /* a marker met inside the entropy-coded data (0 = none) */
    int fake;    /* the byte in bitbuf is padding behind a marker / the end of the data, not stream data */
    int insufficient; /* a padding bit was consumed: jdhuff.c's insufficient_data (the MCUs that follow stay zero) */
} bitreader;

typedef struct {
    int present;
    uint8_t counts[17];
    uint8_t symbols[256];
    int mincode[17], maxcode[18], valptr[17];
} hufftab;

typedef struct {
    int id, h, v, tq, td, ta;
    int bw, bh;      /* blocks per row / column, padded to whole MCUs */
    int16_t *coef;   /* [bh][bw][64], natural order, quantised */
    uint8_t *plane;  /* [bh*8][bw*8] */
    int dw, dh;      /* downsampled (real) size: ceil(W * h / hmax), ceil(H * v / vmax) */
    int pred;
} component;

static const uint8_t zigzag[64] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48,
                                   41, 34, 27, 20, 13, 6,  7,  14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23,
                                   30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

static void build_huff(hufftab *h) { /* T.81 Annex C / F.2.2.3 */
    int code = 0, k = 0;
    for (int l = 1; l <= 16; l++) {
        h->valptr[l] = k;
        h->mincode[l] = code;
        code += h->counts[l];
        k += h->counts[l];
        h->maxcode[l] = h->counts[l] ? code - 1 : -1;
        code <<= 1;
    }
    h->maxcode[17] = 0x7fffffff;
}

static int get_bit(bitreader *br) {
    if (br->bits == 0) {
        int byte = 0;
        br->fake = 1;
        if (br->marker == 0 && br->p < br->end) {
            br->fake = 0;
            byte = *br->p++;
            if (byte == 0xFF) {
                int nx = br->p < br->end ? *br->p : 0xD9;
                if (nx == 0x00) br->p++;             /* stuffed zero */
                else { br->marker = nx; byte = 0; br->p--; br->fake = 1; } /* marker: feed zeros from here on (libjpeg does the same) */
            }
        }
        br->bitbuf = (uint32_t)byte;
        br->bits = 8;
    }
    br->bits--;
    if (br->fake) br->insufficient = 1; /* jdhuff.c jpeg_fill_bit_buffer: "if (nbits > bits_left) ... insufficient_data = TRUE" */
    return (int)((br->bitbuf >> br->bits) & 1);
}
static int get_bits(bitreader *br, int n) {
    int v = 0;
    while (n--) v = (v << 1) | get_bit(br);
    return v;
}
static int decode_symbol(bitreader *br, const hufftab *h) {
    int code = 0;
    for (int l = 1; l <= 16; l++) {
        code = (code << 1) | get_bit(br);
        if (h->maxcode[l] >= 0 && code <= h->maxcode[l] && code >= h->mincode[l]) return h->symbols[h->valptr[l] + code - h->mincode[l]];
    }
    return 0; /* corrupt data: libjpeg warns and uses 0 */
}
static int extend(int v, int t) { return t == 0 ? 0 : (v < (1 << (t - 1)) ? v - (1 << t) + 1 : v); } /* F.2.2.1 */

/* jidctint.c jpeg_idct_islow: CONST_BITS 13, PASS1_BITS 2 */
#define FIX_0_298631336 2446
#define FIX_0_390180644 3196
#define FIX_0_541196100 4433
#define FIX_0_765366865 6270
#define FIX_0_899976223 7373
#define FIX_1_175875602 9633
#define FIX_1_501321110 12299
#define FIX_1_847759065 15137
#define FIX_1_961570560 16069
#define FIX_2_053119869 16819
#define FIX_2_562915447 20995
#define FIX_3_072711026 25172
#define DESCALE(x, n) (((x) + (1L << ((n)-1))) >> (n))

static void idct_1d(const long in[8], long out[8], int shift) {
    long z2 = in[2], z3 = in[6];
    long z1 = (z2 + z3) * FIX_0_541196100;
    long tmp2 = z1 + z3 * (-FIX_1_847759065);
    long tmp3 = z1 + z2 * FIX_0_765366865;
    long tmp0 = (in[0] + in[4]) << 13, tmp1 = (in[0] - in[4]) << 13;
    long tmp10 = tmp0 + tmp3, tmp13 = tmp0 - tmp3, tmp11 = tmp1 + tmp2, tmp12 = tmp1 - tmp2;
    long t0 = in[7], t1 = in[5], t2 = in[3], t3 = in[1];
    long z1b = t0 + t3, z2b = t1 + t2, z3b = t0 + t2, z4b = t1 + t3;
    long z5 = (z3b + z4b) * FIX_1_175875602;
    t0 *= FIX_0_298631336; t1 *= FIX_2_053119869; t2 *= FIX_3_072711026; t3 *= FIX_1_501321110;
    z1b *= -FIX_0_899976223; z2b *= -FIX_2_562915447; z3b *= -FIX_1_961570560; z4b *= -FIX_0_390180644;
    z3b += z5; z4b += z5;
    t0 += z1b + z3b; t1 += z2b + z4b; t2 += z2b + z3b; t3 += z1b + z4b;
    out[0] = DESCALE(tmp10 + t3, shift); out[7] = DESCALE(tmp10 - t3, shift);
    out[1] = DESCALE(tmp11 + t2, shift); out[6] = DESCALE(tmp11 - t2, shift);
    out[2] = DESCALE(tmp12 + t1, shift); out[5] = DESCALE(tmp12 - t1, shift);
    out[3] = DESCALE(tmp13 + t0, shift); out[4] = DESCALE(tmp13 - t0, shift);
}

static void idct_block(const int16_t *coef, const uint16_t *q, uint8_t *dst, int stride) {
    long ws[64];
    for (int c = 0; c < 8; c++) { /* columns; the all-AC-zero shortcut of the library gives the same numbers */
        long in[8], out[8];
        for (int r = 0; r < 8; r++) in[r] = (long)coef[r * 8 + c] * q[r * 8 + c];
        idct_1d(in, out, 13 - 2);
        for (int r = 0; r < 8; r++) ws[r * 8 + c] = out[r];
    }
    for (int r = 0; r < 8; r++) {
        long out[8];
        idct_1d(&ws[r * 8], out, 13 + 2 + 3);
        for (int c = 0; c < 8; c++) {
            /* range_limit[(x) & RANGE_MASK] of the library: the 10-bit wrap, then +128 and clamp */
            long x = ((out[c] + 512) & 1023) - 512 + 128;
            dst[r * stride + c] = (uint8_t)(x < 0 ? 0 : (x > 255 ? 255 : x));
        }
    }
}

static int clamp255(int v) { return v < 0 ? 0 : (v > 255 ? 255 : v); }

/* Decoded size without decoding the scans. Returns 0 on success. */
int orc_jpeg_info(const uint8_t *data, size_t n, int *width, int *height) {
    for (size_t i = 2; i + 8 < n;) {
        if (data[i] != 0xFF) return -1;
        int m = data[i + 1];
        size_t len = ((size_t)data[i + 2] << 8) | data[i + 3];
        if (m == 0xC0 || m == 0xC1 || m == 0xC2) {
            *height = (data[i + 5] << 8) | data[i + 6];
            *width = (data[i + 7] << 8) | data[i + 8];
            return 0;
        }
        i += 2 + len;
    }
    return -1;
}

/* Samples from the quantised coefficients of every component: dequantise + ISLOW IDCT, fancy upsampling, YCbCr -> RGB. */
static void reconstruct(component *comp, int ncomp, int W, int H, int hmax, int vmax, uint16_t (*qt)[64], uint8_t *rgb) {
    /* ---- samples: dequantise + ISLOW IDCT */
    for (int c = 0; c < ncomp; c++)
        for (int by = 0; by < comp[c].bh; by++)
            for (int bx = 0; bx < comp[c].bw; bx++)
                idct_block(comp[c].coef + ((size_t)by * comp[c].bw + bx) * 64, qt[comp[c].tq], comp[c].plane + ((size_t)by * 8 * comp[c].bw + bx) * 8,
                           comp[c].bw * 8);
    /* ---- fancy upsampling (jdsample.c) + YCbCr -> RGB (jdcolor.c) */
    for (int y = 0; y < H; y++)
        for (int x = 0; x < W; x++) {
            int Y = comp[0].plane[(size_t)y * comp[0].bw * 8 + x];
            uint8_t *o = rgb + ((size_t)y * W + x) * 3;
            if (ncomp == 1) { o[0] = o[1] = o[2] = (uint8_t)Y; continue; }
            int cc[2];
            for (int c = 1; c <= 2; c++) {
                const uint8_t *pl = comp[c].plane;
                int st = comp[c].bw * 8, dw = comp[c].dw, dh = comp[c].dh;
                if (hmax == 1 && vmax == 1) {
                    cc[c - 1] = pl[(size_t)y * st + x];
                } else if (vmax == 1) { /* h2v1_fancy_upsample */
                    int cx = x >> 1, t = pl[(size_t)y * st + cx];
                    if (dw <= 2) { cc[c - 1] = t; } /* the library falls back to replication for planes this narrow */
                    else if ((x & 1) == 0) cc[c - 1] = cx == 0 ? t : (3 * t + pl[(size_t)y * st + cx - 1] + 1) >> 2;
                    else cc[c - 1] = cx == dw - 1 ? t : (3 * t + pl[(size_t)y * st + cx + 1] + 2) >> 2;
                } else { /* h2v2_fancy_upsample: 3/4 nearer + 1/4 further row, then the same along the row, 4-bit rounding */
                    int cx = x >> 1, cy = y >> 1;
                    if (dw <= 2) { cc[c - 1] = pl[(size_t)cy * st + cx]; continue; }
                    int fy = (y & 1) ? cy + 1 : cy - 1; /* context rows: the edge rows replicate */
                    if (fy < 0) fy = 0;
                    if (fy > dh - 1) fy = dh - 1;
                    int tc = 3 * pl[(size_t)cy * st + cx] + pl[(size_t)fy * st + cx];
                    if ((x & 1) == 0) {
                        if (cx == 0) cc[c - 1] = (tc * 4 + 8) >> 4;
                        else cc[c - 1] = (3 * tc + 3 * pl[(size_t)cy * st + cx - 1] + pl[(size_t)fy * st + cx - 1] + 8) >> 4;
                    } else {
                        if (cx == dw - 1) cc[c - 1] = (tc * 4 + 7) >> 4;
                        else cc[c - 1] = (3 * tc + 3 * pl[(size_t)cy * st + cx + 1] + pl[(size_t)fy * st + cx + 1] + 7) >> 4;
                    }
                }
            }
            int cb = cc[0] - 128, cr = cc[1] - 128;
            /* SCALEBITS 16: FIX(1.40200) = 91881, FIX(1.77200) = 116130, FIX(0.71414) = 46802, FIX(0.34414) = 22554 */
            int r = Y + (int)((91881L * cr + 32768) >> 16);
            int g = Y + (int)((-22554L * cb + 32768 - 46802L * cr) >> 16);
            int b = Y + (int)((116130L * cb + 32768) >> 16);
            o[0] = (uint8_t)clamp255(r); o[1] = (uint8_t)clamp255(g); o[2] = (uint8_t)clamp255(b);
        }
}

/* ---- progressive scans (T.81 Annex G; the control flow of libjpeg's jdphuff.c: decode_mcu_DC_first / _AC_first / _DC_refine /
 * _AC_refine).  One scan adds a band [Ss, Se] of coefficients, or one more bit (Al) of a band, of one component (AC) or of several
 * interleaved ones (DC).  eobrun: blocks still covered by an end-of-band run. */
typedef struct { int nc, ci[3], Ss, Se, Ah, Al; } scan_hdr;

static void restart_marker(bitreader *br) {
    br->bits = 0;
    if (br->marker >= 0xD0 && br->marker <= 0xD7) { br->p += 2; br->marker = 0; br->insufficient = 0; }
    else if (br->marker == 0 && br->p + 1 < br->end && br->p[0] == 0xFF && br->p[1] >= 0xD0 && br->p[1] <= 0xD7) { br->p += 2; br->insufficient = 0; }
}

static void prog_block(bitreader *br, const scan_hdr *sh, component *cp, const hufftab *dct, const hufftab *act, int16_t *blk, int *eobrun) {
    const int Al = sh->Al;
    if (sh->Ss == 0) { /* DC */
        if (sh->Ah == 0) {
            int t = decode_symbol(br, dct);
            cp->pred += extend(get_bits(br, t), t);
            blk[0] = (int16_t)(cp->pred * (1 << Al));
        } else if (get_bit(br)) {
            blk[0] |= (int16_t)(1 << Al);
        }
        return;
    }
    if (sh->Ah == 0) { /* AC first */
        if (*eobrun > 0) { (*eobrun)--; return; }
        for (int k = sh->Ss; k <= sh->Se; k++) {
            int rs = decode_symbol(br, act), r = rs >> 4, sz = rs & 15;
            if (sz) {
                k += r;
                int v = extend(get_bits(br, sz), sz);
                if (k <= 63) blk[zigzag[k]] = (int16_t)(v * (1 << Al));
            } else if (r == 15) {
                k += 15;
            } else {
                *eobrun = 1 << r;
                if (r) *eobrun += get_bits(br, r);
                (*eobrun)--;
                break;
            }
        }
        return;
    }
    /* AC refinement */
    const int p1 = 1 << Al, m1 = -(1 << Al);
    int k = sh->Ss;
    if (*eobrun == 0) {
        for (; k <= sh->Se; k++) {
            int rs = decode_symbol(br, act), r = rs >> 4, sz = rs & 15, val = 0;
            if (sz) {
                val = get_bit(br) ? p1 : m1; /* sz must be 1 */
            } else if (r != 15) {
                *eobrun = 1 << r;
                if (r) *eobrun += get_bits(br, r);
                break; /* the rest of the band is handled as part of the run */
            }
            do { /* skip r still-zero coefficients; every already-nonzero one on the way takes a correction bit */
                int16_t *c = &blk[zigzag[k]];
                if (*c != 0) {
                    if (get_bit(br) && (*c & p1) == 0) *c = (int16_t)(*c + (*c >= 0 ? p1 : m1));
                } else if (--r < 0) {
                    break;
                }
                k++;
            } while (k <= sh->Se);
            if (sz && k <= 63) blk[zigzag[k]] = (int16_t)val;
        }
    }
    if (*eobrun > 0) {
        for (; k <= sh->Se; k++) {
            int16_t *c = &blk[zigzag[k]];
            if (*c != 0 && get_bit(br) && (*c & p1) == 0) *c = (int16_t)(*c + (*c >= 0 ? p1 : m1));
        }
        (*eobrun)--;
    }
}

/* One progressive scan; returns the offset of the first byte behind its entropy-coded data. */
static size_t prog_scan(const uint8_t *data, size_t n, size_t ecs, const scan_hdr *sh, component *comp, int W, int H, int hmax, int vmax, int restart,
                        const hufftab *dc, const hufftab *ac) {
    bitreader br = {data + ecs, data + n, 0, 0, 0, 0, 0};
    int eobrun = 0, mcus = 0;
    for (int s = 0; s < sh->nc; s++) comp[sh->ci[s]].pred = 0;
    if (sh->nc == 1) { /* non-interleaved: the component's own blocks in raster order, not padded to whole MCUs */
        component *cp = &comp[sh->ci[0]];
        int wib = (cp->dw + 7) / 8, hib = (cp->dh + 7) / 8;
        for (int by = 0; by < hib; by++)
            for (int bx = 0; bx < wib; bx++) {
                if (restart && mcus > 0 && mcus % restart == 0) { restart_marker(&br); cp->pred = 0; eobrun = 0; }
                mcus++;
                if (br.insufficient) continue;
                prog_block(&br, sh, cp, &dc[cp->td], &ac[cp->ta], cp->coef + ((size_t)by * cp->bw + bx) * 64, &eobrun);
            }
    } else { /* interleaved (DC scans only) */
        int mcux = (W + 8 * hmax - 1) / (8 * hmax), mcuy = (H + 8 * vmax - 1) / (8 * vmax);
        for (int my = 0; my < mcuy; my++)
            for (int mx = 0; mx < mcux; mx++) {
                if (restart && mcus > 0 && mcus % restart == 0) { restart_marker(&br); for (int s = 0; s < sh->nc; s++) comp[sh->ci[s]].pred = 0; }
                mcus++;
                if (br.insufficient) continue;
                for (int s = 0; s < sh->nc; s++) {
                    component *cp = &comp[sh->ci[s]];
                    for (int by = 0; by < cp->v; by++)
                        for (int bx = 0; bx < cp->h; bx++)
                            prog_block(&br, sh, cp, &dc[cp->td], &ac[cp->ta], cp->coef + ((size_t)(my * cp->v + by) * cp->bw + mx * cp->h + bx) * 64, &eobrun);
                }
            }
    }
    /* the reader stops in front of the marker that ends the scan (or at the end of the data) */
    size_t pos = (size_t)(br.p - data);
    return pos;
}

/* imdecode(IMREAD_COLOR) + BGR2RGB (utils.rs:8-21): rgb = [H][W][3]. Returns 0, or a negative code (-2 = unsupported). */
int orc_jpeg_decode_rgb(const uint8_t *data, size_t n, uint8_t *rgb, int cap_w, int cap_h) {
    if (n < 4 || data[0] != 0xFF || data[1] != 0xD8) return -1;
    uint16_t qt[4][64];
    hufftab dc[4], ac[4];
    memset(dc, 0, sizeof dc);
    memset(ac, 0, sizeof ac);
    component comp[3];
    memset(comp, 0, sizeof comp);
    int ncomp = 0, W = 0, H = 0, hmax = 1, vmax = 1, restart = 0, rc = -1, progressive = 0, scans = 0;
    int coef_bits[3][64];
    memset(coef_bits, 0xff, sizeof coef_bits); /* -1: never sent */
    size_t i = 2;
    while (i + 4 <= n) {
        if (data[i] != 0xFF) { i++; continue; }
        int m = data[i + 1];
        if (m == 0xFF) { i++; continue; }
        if (m == 0xD9) break;
        size_t len = ((size_t)data[i + 2] << 8) | data[i + 3];
        const uint8_t *seg = data + i + 4;
        if (i + 2 + len > n) goto done;
        if (m == 0xDB) { /* DQT */
            for (size_t k = 0; k + 1 < len - 2;) {
                int pq = seg[k] >> 4, tq = seg[k] & 15;
                k++;
                if (tq > 3) goto done;
                for (int z = 0; z < 64; z++) {
                    qt[tq][zigzag[z]] = pq ? (uint16_t)((seg[k] << 8) | seg[k + 1]) : seg[k];
                    k += pq ? 2 : 1;
                }
            }
        } else if (m == 0xC4) { /* DHT */
            for (size_t k = 0; k + 17 <= len - 2;) {
                int tc = seg[k] >> 4, th = seg[k] & 15;
                if (th > 3 || tc > 1) goto done;
                hufftab *h = tc ? &ac[th] : &dc[th];
                int total = 0;
                for (int l = 1; l <= 16; l++) { h->counts[l] = seg[k + l]; total += seg[k + l]; }
                if (total > 256 || k + 17 + (size_t)total > len - 2) goto done;
                memcpy(h->symbols, seg + k + 17, (size_t)total);
                h->present = 1;
                build_huff(h);
                k += 17 + (size_t)total;
            }
        } else if (m == 0xC0 || m == 0xC1 || m == 0xC2) { /* SOF0 / SOF1: sequential Huffman; SOF2: progressive Huffman */
            if (seg[0] != 8 || ncomp) { rc = -2; goto done; }
            progressive = m == 0xC2;
            H = (seg[1] << 8) | seg[2];
            W = (seg[3] << 8) | seg[4];
            ncomp = seg[5];
            if ((ncomp != 1 && ncomp != 3) || W <= 0 || H <= 0 || W > cap_w || H > cap_h) { rc = -2; goto done; }
            for (int c = 0; c < ncomp; c++) {
                comp[c].id = seg[6 + 3 * c];
                comp[c].h = seg[7 + 3 * c] >> 4;
                comp[c].v = seg[7 + 3 * c] & 15;
                comp[c].tq = seg[8 + 3 * c];
                if (comp[c].h > hmax) hmax = comp[c].h;
                if (comp[c].v > vmax) vmax = comp[c].v;
            }
            if (ncomp == 1) { comp[0].h = comp[0].v = 1; hmax = vmax = 1; } /* a single component is never interleaved */
            if (progressive) { /* the coefficient arrays live across the scans */
                if (ncomp == 3 && !((comp[0].h == 1 || comp[0].h == 2) && (comp[0].v == 1 || comp[0].v == 2) && comp[1].h == 1 && comp[1].v == 1 &&
                                    comp[2].h == 1 && comp[2].v == 1 && !(comp[0].h == 1 && comp[0].v == 2))) { rc = -2; goto done; }
                int mcux = (W + 8 * hmax - 1) / (8 * hmax), mcuy = (H + 8 * vmax - 1) / (8 * vmax);
                for (int c = 0; c < ncomp; c++) {
                    comp[c].bw = mcux * comp[c].h;
                    comp[c].bh = mcuy * comp[c].v;
                    comp[c].dw = (W * comp[c].h + hmax - 1) / hmax;
                    comp[c].dh = (H * comp[c].v + vmax - 1) / vmax;
                    comp[c].coef = (int16_t *)calloc((size_t)comp[c].bw * comp[c].bh * 64, sizeof(int16_t));
                    comp[c].plane = (uint8_t *)malloc((size_t)comp[c].bw * comp[c].bh * 64);
                    if (!comp[c].coef || !comp[c].plane) goto done;
                }
            }
        } else if (m >= 0xC3 && m <= 0xCF && m != 0xC4 && m != 0xC8 && m != 0xCC) {
            rc = -2; /* lossless / arithmetic / hierarchical: outside this restatement */
            goto done;
        } else if (m == 0xDD) {
            restart = (seg[0] << 8) | seg[1];
        } else if (m == 0xDA && progressive) { /* SOS of a progressive frame: one of several scans */
            scan_hdr sh;
            sh.nc = seg[0];
            if (ncomp == 0 || sh.nc < 1 || sh.nc > ncomp || len < (size_t)(6 + 2 * sh.nc)) goto done;
            for (int s2 = 0; s2 < sh.nc; s2++) {
                int cid = seg[1 + 2 * s2], c = 0;
                while (c < ncomp && comp[c].id != cid) c++;
                if (c == ncomp) goto done;
                sh.ci[s2] = c;
                comp[c].td = seg[2 + 2 * s2] >> 4;
                comp[c].ta = seg[2 + 2 * s2] & 15;
                if (comp[c].td > 3 || comp[c].ta > 3) goto done;
            }
            sh.Ss = seg[1 + 2 * sh.nc]; sh.Se = seg[2 + 2 * sh.nc]; sh.Ah = seg[3 + 2 * sh.nc] >> 4; sh.Al = seg[3 + 2 * sh.nc] & 15;
            if (sh.Ss > sh.Se || sh.Se > 63 || (sh.Ss == 0 && sh.Se != 0) || (sh.Ss > 0 && sh.nc != 1) || sh.Al > 13 || sh.Ah > 13) goto done;
            for (int s2 = 0; s2 < sh.nc; s2++) /* jdphuff.c start_pass_phuff_decoder: the Al of the last scan that carried coefficient k */
                for (int k = sh.Ss; k <= sh.Se; k++) coef_bits[sh.ci[s2]][k] = sh.Al;
            i = prog_scan(data, n, i + 2 + len, &sh, comp, W, H, hmax, vmax, restart, dc, ac);
            scans++;
            continue;
        } else if (m == 0xDA) { /* SOS: one interleaved scan holding every component */
            if (ncomp == 0 || seg[0] != ncomp) { rc = -2; goto done; }
            for (int s = 0; s < ncomp; s++) {
                int cid = seg[1 + 2 * s], c = 0;
                while (c < ncomp && comp[c].id != cid) c++;
                if (c == ncomp) goto done;
                comp[c].td = seg[2 + 2 * s] >> 4;
                comp[c].ta = seg[2 + 2 * s] & 15;
            }
            if (ncomp == 3 && !((comp[0].h == 1 || comp[0].h == 2) && (comp[0].v == 1 || comp[0].v == 2) && comp[1].h == 1 && comp[1].v == 1 &&
                                comp[2].h == 1 && comp[2].v == 1 && !(comp[0].h == 1 && comp[0].v == 2))) { rc = -2; goto done; }
            int mcux = (W + 8 * hmax - 1) / (8 * hmax), mcuy = (H + 8 * vmax - 1) / (8 * vmax);
            for (int c = 0; c < ncomp; c++) {
                comp[c].bw = mcux * comp[c].h;
                comp[c].bh = mcuy * comp[c].v;
                comp[c].dw = (W * comp[c].h + hmax - 1) / hmax;
                comp[c].dh = (H * comp[c].v + vmax - 1) / vmax;
                comp[c].coef = (int16_t *)calloc((size_t)comp[c].bw * comp[c].bh * 64, sizeof(int16_t));
                comp[c].plane = (uint8_t *)malloc((size_t)comp[c].bw * comp[c].bh * 64);
                if (!comp[c].coef || !comp[c].plane) goto done;
            }
            bitreader br = {data + i + 2 + len, data + n, 0, 0, 0, 0, 0};
            int mcus = 0;
            for (int my = 0; my < mcuy; my++)
                for (int mx = 0; mx < mcux; mx++) {
                    if (restart && mcus > 0 && mcus % restart == 0) { /* RSTn: byte-align, skip the marker, reset the predictors */
                        br.bits = 0;
                        if (br.marker >= 0xD0 && br.marker <= 0xD7) { br.p += 2; br.marker = 0; br.insufficient = 0; } /* already met by the bit reader */
                        else if (br.marker == 0 && br.p + 1 < br.end && br.p[0] == 0xFF && br.p[1] >= 0xD0 && br.p[1] <= 0xD7) { br.p += 2; br.insufficient = 0; }
                        for (int c = 0; c < ncomp; c++) comp[c].pred = 0;
                    }
                    mcus++;
                    /* jdhuff.c decode_mcu: "If we've run out of data, just leave the MCU set to zeroes": the MCU in which the data
                     * ran out is finished on padding zeros, the ones behind it are not decoded (uniform grey) until a restart
                     * marker is found again (process_restart clears the flag only then) */
                    if (br.insufficient) continue;
                    for (int c = 0; c < ncomp; c++)
                        for (int by = 0; by < comp[c].v; by++)
                            for (int bx = 0; bx < comp[c].h; bx++) {
                                int16_t *blk = comp[c].coef + ((size_t)(my * comp[c].v + by) * comp[c].bw + mx * comp[c].h + bx) * 64;
                                int t = decode_symbol(&br, &dc[comp[c].td]);
                                comp[c].pred += extend(get_bits(&br, t), t);
                                blk[0] = (int16_t)comp[c].pred;
                                for (int k = 1; k < 64;) {
                                    int rs = decode_symbol(&br, &ac[comp[c].ta]);
                                    int r = rs >> 4, s = rs & 15;
                                    if (s == 0) {
                                        if (r != 15) break; /* EOB */
                                        k += 16;
                                        continue;
                                    }
                                    k += r;
                                    if (k > 63) break;
                                    blk[zigzag[k]] = (int16_t)extend(get_bits(&br, s), s);
                                    k++;
                                }
                            }
                }
            reconstruct(comp, ncomp, W, H, hmax, vmax, qt, rgb);
            rc = 0;
            goto done;
        }
        i += 2 + len;
    }
    if (progressive && scans > 0) { /* EOI (or the end of the data): every scan that was there has been added */
        /* jdcoefct.c smoothing_ok(): with every component's DC known and one of the first AC coefficients (1..9 in libjpeg-turbo >= 2.1)
         * missing or not refined to its last bit, libjpeg decodes through decompress_smooth_data (inter-block smoothing).  That pass is
         * outside this restatement: such an INCOMPLETE progressive stream is refused (-3), never decoded differently. */
        int dc_known = 1, low_ac_open = 0;
        for (int c = 0; c < ncomp; c++) {
            if (coef_bits[c][0] < 0) dc_known = 0;
            for (int k = 1; k <= 9; k++) if (coef_bits[c][k] != 0) low_ac_open = 1;
        }
        if (dc_known && low_ac_open) { rc = -3; goto done; }
        reconstruct(comp, ncomp, W, H, hmax, vmax, qt, rgb);
        rc = 0;
    }
done:
    for (int c = 0; c < 3; c++) { free(comp[c].coef); free(comp[c].plane); }
    return rc;
}
