// jpeg.cpp — host half of the JPEG path (see jpeg.hpp): marker parsing and Huffman decoding (ITU-T T.81 Annex F.2).
// Entropy-coded data is one serial bit stream per restart interval; it stays on the CPU.  Everything after it — the
// per-block and per-pixel arithmetic — runs on the GPU (jpeg_kernels.hip).
#include "jpeg.hpp"

#include <algorithm>
#include <cstring>
#include <stdexcept>

namespace mi {
namespace {

const uint8_t kZigzag[64] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48,
                             41, 34, 27, 20, 13, 6,  7,  14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23,
                             30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

constexpr int64_t kMaxPixels = int64_t(1) << 27;  // bounds the coefficient buffer (2 B x 3 planes per pixel) before it is allocated
constexpr int kLook = 9;  // codes up to 9 bits resolve with one table read

struct Huff {
    bool present = false;
    uint8_t symbols[256] = {};
    uint16_t look[1 << kLook] = {};  // (length << 8) | symbol, 0 = longer code
    int maxcode[18] = {}, valoff[17] = {};
    // false = the length counts over-subscribe the code space (a code of length l would not fit in l bits): libjpeg's
    // jdhuff.c rejects such a DHT ("bad Huffman table"); accepting it would index look[] / symbols[] out of range.
    bool build(const uint8_t counts[17]) {
        int code = 0, k = 0;
        present = false;
        std::memset(look, 0, sizeof look);
        for (int l = 1; l <= 16; l++) {
            // jdhuff.c jpeg_make_d_derived_tbl: after the codes of length l "code is now 1 more than the last code used ...
            // it must still fit in si bits, since no code is allowed to be all ones": code + counts[l] >= 2^l is a bad table
            if (counts[l] && code + counts[l] >= (1 << l)) return false;
            valoff[l] = k - code;
            for (int i = 0; i < counts[l]; i++, code++, k++)
                if (l <= kLook)
                    for (int fill = 0; fill < (1 << (kLook - l)); fill++) look[(code << (kLook - l)) | fill] = static_cast<uint16_t>((l << 8) | symbols[k]);
            maxcode[l] = counts[l] ? code - 1 : -1;
            code <<= 1;
        }
        maxcode[17] = 0x7fffffff;
        present = true;
        return true;
    }
};

struct Bits {
    const uint8_t *p, *end;
    uint64_t buf = 0;
    int n = 0;
    int real = 0;               // how many of the n buffered bits are stream data (the rest is padding behind a marker / the end)
    bool insufficient = false;  // a padding bit was consumed: jdhuff.c's insufficient_data
    bool hit_marker = false;
    void fill() {  // keep at least 32 bits; past a marker (or the end) the stream reads as zeros, like libjpeg
        while (n <= 56) {
            unsigned byte = 0;
            if (!hit_marker && p < end) {
                byte = *p++;
                real += 8;
                if (byte == 0xFF) {
                    unsigned nx = p < end ? *p : 0xD9u;
                    if (nx == 0) p++;
                    else { hit_marker = true; byte = 0; p--; real -= 8; }
                }
            }
            buf |= static_cast<uint64_t>(byte) << (56 - n);
            n += 8;
        }
    }
    unsigned peek(int k) { return static_cast<unsigned>(buf >> (64 - k)); }
    void skip(int k) {
        buf <<= k; n -= k;
        if ((real -= k) < 0) { real = 0; insufficient = true; }
    }
    int get(int k) {
        if (k == 0) return 0;
        if (n < k) fill();
        unsigned v = peek(k);
        skip(k);
        return static_cast<int>(v);
    }
    int symbol(const Huff& h) {
        if (n < 16) fill();
        unsigned e = h.look[peek(kLook)];
        if (e) { skip(static_cast<int>(e >> 8)); return static_cast<int>(e & 255); }
        int code = static_cast<int>(peek(kLook));
        for (int l = kLook + 1; l <= 16; l++) {
            code = static_cast<int>(peek(l));
            if (h.maxcode[l] >= 0 && code <= h.maxcode[l]) {
                skip(l);
                const int at = code + h.valoff[l];
                return at >= 0 && at < 256 ? h.symbols[at] : 0;
            }
        }
        skip(16);
        return 0;  // corrupt stream: libjpeg warns and substitutes 0
    }
    void restart() {  // byte-align and step over RSTn; data found again ends the run of skipped MCUs (jdhuff.c process_restart)
        n = 0; buf = 0; real = 0;
        if (hit_marker) { if (p + 1 < end && p[1] >= 0xD0 && p[1] <= 0xD7) { p += 2; hit_marker = false; insufficient = false; } }
        else if (p + 1 < end && p[0] == 0xFF && p[1] >= 0xD0 && p[1] <= 0xD7) { p += 2; insufficient = false; }
    }
};

inline int extend(int v, int t) { return v < (1 << (t - 1)) ? v - (1 << t) + 1 : v; }

[[noreturn]] void bad(const char* what) { throw std::runtime_error(std::string("jpeg: ") + what); }

struct ScanHdr {  // progressive scans: which components, which band of coefficients [Ss, Se], which bit (Ah: previous, Al: this)
    int nc = 0, ci[3] = {0, 0, 0}, Ss = 0, Se = 0, Ah = 0, Al = 0;
};

struct Parser {
    const uint8_t* d;
    size_t n;
    JpegFrame* f;
    Huff dc[4], ac[4];
    int restart_interval = 0;
    bool have_frame = false;

    // walks the markers from `start` up to SOS; returns the offset of the entropy-coded data (0 when stop_at_frame and a frame header
    // was found, or — behind the first scan of a progressive frame, `sh` given — when EOI / the end of the data comes first)
    size_t headers(bool stop_at_frame, size_t start = 2, ScanHdr* sh = nullptr) {
        if (n < 4 || d[0] != 0xFF || d[1] != 0xD8) bad("not a JPEG stream (no SOI)");
        size_t i = start;
        while (i + 4 <= n) {
            if (d[i] != 0xFF) { i++; continue; }
            const int m = d[i + 1];
            if (m == 0xFF) { i++; continue; }
            if (m == 0xD9) break;
            const size_t len = (static_cast<size_t>(d[i + 2]) << 8) | d[i + 3];
            if (len < 2 || i + 2 + len > n) bad("truncated segment");
            const uint8_t* seg = d + i + 4;
            const size_t body = len - 2;
            if (m == 0xDB) {
                for (size_t k = 0; k < body;) {
                    const int pq = seg[k] >> 4, tq = seg[k] & 15;
                    k++;
                    if (tq > 3 || k + (pq ? 128u : 64u) > body) bad("bad quantisation table");
                    for (int z = 0; z < 64; z++) {
                        f->qt[tq][kZigzag[z]] = pq ? static_cast<uint16_t>((seg[k] << 8) | seg[k + 1]) : seg[k];
                        k += pq ? 2 : 1;
                    }
                }
            } else if (m == 0xC4) {
                for (size_t k = 0; k + 17 <= body;) {
                    const int tc = seg[k] >> 4, th = seg[k] & 15;
                    if (th > 3 || tc > 1) bad("bad Huffman table id");
                    uint8_t counts[17] = {};
                    int total = 0;
                    for (int l = 1; l <= 16; l++) { counts[l] = seg[k + l]; total += counts[l]; }
                    if (total > 256 || k + 17 + static_cast<size_t>(total) > body) bad("bad Huffman table");
                    Huff& h = tc ? ac[th] : dc[th];
                    std::memset(h.symbols, 0, sizeof h.symbols);
                    std::memcpy(h.symbols, seg + k + 17, static_cast<size_t>(total));
                    if (!h.build(counts)) bad("bad Huffman table");
                    k += 17 + static_cast<size_t>(total);
                }
            } else if (m == 0xC0 || m == 0xC1 || m == 0xC2) {
                if (have_frame) bad("second frame header");
                f->progressive = m == 0xC2;
                if (body < 6 || seg[0] != 8) bad("unsupported sample precision (8-bit only)");
                f->height = (seg[1] << 8) | seg[2];
                f->width = (seg[3] << 8) | seg[4];
                f->ncomp = seg[5];
                if (f->width <= 0 || f->height <= 0) bad("empty frame");
                if (static_cast<int64_t>(f->width) * f->height > kMaxPixels) bad("picture too large (more than 2^27 pixels)");
                if ((f->ncomp != 1 && f->ncomp != 3) || body < 6 + 3 * static_cast<size_t>(f->ncomp)) bad("unsupported component count (1 or 3)");
                f->hmax = f->vmax = 1;
                for (int c = 0; c < f->ncomp; c++) {
                    JpegComponent& cp = f->comp[c];
                    cp.id = seg[6 + 3 * c]; cp.h = seg[7 + 3 * c] >> 4; cp.v = seg[7 + 3 * c] & 15; cp.tq = seg[8 + 3 * c];
                    if (cp.tq > 3 || cp.h < 1 || cp.v < 1) bad("bad component");
                    f->hmax = cp.h > f->hmax ? cp.h : f->hmax;
                    f->vmax = cp.v > f->vmax ? cp.v : f->vmax;
                }
                if (f->ncomp == 1) f->comp[0].h = f->comp[0].v = f->hmax = f->vmax = 1;
                else {
                    const JpegComponent* c = f->comp;
                    const bool ok = c[1].h == 1 && c[1].v == 1 && c[2].h == 1 && c[2].v == 1 && ((c[0].h == 1 && c[0].v == 1) || (c[0].h == 2 && c[0].v <= 2));
                    if (!ok) bad("unsupported chroma sampling (h1v1, h2v1, h2v2)");
                }
                have_frame = true;
                if (stop_at_frame) return 0;
            } else if (m >= 0xC3 && m <= 0xCF && m != 0xC4 && m != 0xC8 && m != 0xCC) {
                bad("unsupported coding process (baseline / extended sequential / progressive Huffman only)");
            } else if (m == 0xDD) {
                if (body < 2) bad("bad DRI");
                restart_interval = (seg[0] << 8) | seg[1];
            } else if (m == 0xDA && f->progressive) {
                if (!have_frame || !sh) bad("scan before frame header");
                sh->nc = body >= 1 ? seg[0] : 0;
                if (sh->nc < 1 || sh->nc > f->ncomp || body < 4 + 2 * static_cast<size_t>(sh->nc)) bad("bad scan header");
                for (int s = 0; s < sh->nc; s++) {
                    int c = 0;
                    while (c < f->ncomp && f->comp[c].id != seg[1 + 2 * s]) c++;
                    if (c == f->ncomp) bad("scan names an unknown component");
                    for (int q = 0; q < s; q++)
                        if (sh->ci[q] == c) bad("scan names a component twice");
                    sh->ci[s] = c;
                    f->comp[c].td = seg[2 + 2 * s] >> 4;
                    f->comp[c].ta = seg[2 + 2 * s] & 15;
                    if (f->comp[c].td > 3 || f->comp[c].ta > 3) bad("scan uses an undefined Huffman table");
                }
                const uint8_t* t = seg + 1 + 2 * sh->nc;
                sh->Ss = t[0]; sh->Se = t[1]; sh->Ah = t[2] >> 4; sh->Al = t[2] & 15;
                // jdphuff.c start_pass_phuff_decoder: the band and the bit positions a scan may name
                if (sh->Ss > sh->Se || sh->Se > 63 || (sh->Ss == 0 && sh->Se != 0) || (sh->Ss > 0 && sh->nc != 1) || sh->Al > 13 || sh->Ah > 13)
                    bad("bad progressive scan parameters");
                for (int s = 0; s < sh->nc; s++) {
                    const JpegComponent& cp = f->comp[sh->ci[s]];
                    if (sh->Ss == 0 ? (sh->Ah == 0 && !dc[cp.td].present) : !ac[cp.ta].present) bad("scan uses an undefined Huffman table");
                }
                return i + 2 + len;
            } else if (m == 0xDA) {
                if (!have_frame) bad("scan before frame header");
                if (body < 1 || seg[0] != f->ncomp || body < 1 + 2 * static_cast<size_t>(f->ncomp)) bad("unsupported scan layout (one interleaved scan)");
                for (int s = 0; s < f->ncomp; s++) {
                    int c = 0;
                    while (c < f->ncomp && f->comp[c].id != seg[1 + 2 * s]) c++;
                    if (c == f->ncomp) bad("scan names an unknown component");
                    f->comp[c].td = seg[2 + 2 * s] >> 4;
                    f->comp[c].ta = seg[2 + 2 * s] & 15;
                    if (f->comp[c].td > 3 || f->comp[c].ta > 3 || !dc[f->comp[c].td].present || !ac[f->comp[c].ta].present) bad("scan uses an undefined Huffman table");
                }
                return i + 2 + len;
            }
            i += 2 + len;
        }
        if (sh && start > 2) return 0;  // EOI, or the data ends: every scan that was there has been read
        bad(stop_at_frame ? "no frame header" : "no scan");
    }
};

// ---- progressive scans: the control flow of libjpeg's jdphuff.c (decode_mcu_DC_first / _AC_first / _DC_refine / _AC_refine)
void prog_block(Bits& br, const ScanHdr& sh, int& pred, const Huff& hd, const Huff& ha, int16_t* blk, int& eobrun) {
    const int Al = sh.Al;
    if (sh.Ss == 0) {
        if (sh.Ah == 0) {
            const int t = br.symbol(hd) & 15;
            if (t) pred += extend(br.get(t), t);
            blk[0] = static_cast<int16_t>(pred * (1 << Al));
        } else if (br.get(1)) {
            blk[0] = static_cast<int16_t>(blk[0] | (1 << Al));
        }
        return;
    }
    if (sh.Ah == 0) {  // first pass over a band of AC coefficients
        if (eobrun > 0) { eobrun--; return; }
        for (int k = sh.Ss; k <= sh.Se; k++) {
            const int rs = br.symbol(ha), r = rs >> 4, sz = rs & 15;
            if (sz) {
                k += r;
                const int v = extend(br.get(sz), sz);
                if (k <= 63) blk[kZigzag[k]] = static_cast<int16_t>(v * (1 << Al));
            } else if (r == 15) {
                k += 15;
            } else {
                eobrun = 1 << r;
                if (r) eobrun += br.get(r);
                eobrun--;
                break;
            }
        }
        return;
    }
    // refinement of a band: one more bit for the coefficients that are non-zero already, new +-1 << Al ones in between
    const int p1 = 1 << Al, m1 = -(1 << Al);
    auto correct = [&](int16_t& c) {
        if (br.get(1) && (c & p1) == 0) c = static_cast<int16_t>(c + (c >= 0 ? p1 : m1));
    };
    int k = sh.Ss;
    if (eobrun == 0) {
        for (; k <= sh.Se; k++) {
            const int rs = br.symbol(ha), sz = rs & 15;
            int r = rs >> 4, val = 0;
            if (sz) {
                val = br.get(1) ? p1 : m1;
            } else if (r != 15) {
                eobrun = 1 << r;
                if (r) eobrun += br.get(r);
                break;  // the rest of the band belongs to the run
            }
            do {
                int16_t& c = blk[kZigzag[k]];
                if (c != 0) correct(c);
                else if (--r < 0) break;
                k++;
            } while (k <= sh.Se);
            if (sz && k <= 63) blk[kZigzag[k]] = static_cast<int16_t>(val);
        }
    }
    if (eobrun > 0) {
        for (; k <= sh.Se; k++) {
            int16_t& c = blk[kZigzag[k]];
            if (c != 0) correct(c);
        }
        eobrun--;
    }
}

// one progressive scan; returns the offset at which the marker walk continues
size_t prog_scan(const uint8_t* data, size_t n, size_t ecs, const ScanHdr& sh, Parser& p, JpegFrame& f) {
    Bits br{data + ecs, data + n};
    int pred[3] = {0, 0, 0}, eobrun = 0, mcus = 0;
    auto at_restart = [&] {
        if (p.restart_interval && mcus > 0 && mcus % p.restart_interval == 0) {
            br.restart();
            pred[0] = pred[1] = pred[2] = 0;
            eobrun = 0;
        }
        mcus++;
    };
    if (sh.nc == 1) {  // non-interleaved: the component's own blocks in raster order, not padded to whole MCUs
        const JpegComponent& cp = f.comp[sh.ci[0]];
        const int wib = (cp.dw + 7) / 8, hib = (cp.dh + 7) / 8;
        for (int by = 0; by < hib; by++)
            for (int bx = 0; bx < wib; bx++) {
                at_restart();
                if (br.insufficient) continue;
                prog_block(br, sh, pred[0], p.dc[cp.td], p.ac[cp.ta], f.coef.data() + cp.coef_off + (static_cast<size_t>(by) * cp.bw + bx) * 64, eobrun);
            }
    } else {  // interleaved (DC scans)
        const int mcux = (f.width + 8 * f.hmax - 1) / (8 * f.hmax), mcuy = (f.height + 8 * f.vmax - 1) / (8 * f.vmax);
        for (int my = 0; my < mcuy; my++)
            for (int mx = 0; mx < mcux; mx++) {
                at_restart();
                if (br.insufficient) continue;
                for (int s = 0; s < sh.nc; s++) {
                    const JpegComponent& cp = f.comp[sh.ci[s]];
                    for (int by = 0; by < cp.v; by++)
                        for (int bx = 0; bx < cp.h; bx++)
                            prog_block(br, sh, pred[s], p.dc[cp.td], p.ac[cp.ta],
                                       f.coef.data() + cp.coef_off + (static_cast<size_t>(my * cp.v + by) * cp.bw + mx * cp.h + bx) * 64, eobrun);
                }
            }
    }
    return static_cast<size_t>(br.p - data);  // the reader stops in front of the marker that ends the scan (or at the end)
}

}  // namespace

void jpeg_parse_size(const uint8_t* data, size_t n, int* width, int* height) {
    JpegFrame f;
    Parser p{data, n, &f};
    p.headers(true);
    *width = f.width;
    *height = f.height;
}

void jpeg_entropy_decode(const uint8_t* data, size_t n, JpegFrame* out) {
    JpegFrame& f = *out;
    {
        const auto provide = f.coef.provide;   // (where the coefficients go survives the reset)
        void* const ctx = f.coef.ctx;
        f = JpegFrame();
        f.coef.provide = provide;
        f.coef.ctx = ctx;
    }
    Parser p{data, n, &f};
    ScanHdr sh;
    const size_t ecs = p.headers(false, 2, &sh);
    const int mcux = (f.width + 8 * f.hmax - 1) / (8 * f.hmax), mcuy = (f.height + 8 * f.vmax - 1) / (8 * f.vmax);
    size_t total = 0;
    for (int c = 0; c < f.ncomp; c++) {
        JpegComponent& cp = f.comp[c];
        cp.bw = mcux * cp.h; cp.bh = mcuy * cp.v;
        cp.dw = (f.width * cp.h + f.hmax - 1) / f.hmax;
        cp.dh = (f.height * cp.v + f.vmax - 1) / f.vmax;
        cp.coef_off = total;
        total += static_cast<size_t>(cp.bw) * cp.bh * 64;
    }
    // A short entropy-coded segment is NOT an error (libjpeg pads it with zeros, warns and returns a picture whose tail is grey:
    // the `insufficient` flag below).  What is refused is the allocation bomb: a few header bytes that promise a huge frame.
    // Every block costs at least two bits of entropy-coded data (one DC and one AC symbol); a header asking for more than 32 MiB
    // of coefficients that the remaining bytes cannot possibly hold is refused before the buffer is allocated.
    // (a progressive frame can spend a single bit per block — a DC-only first scan — and almost nothing on end-of-band runs)
    if (total * sizeof(int16_t) > (32u << 20) && total / 64 > (f.progressive ? 8 : 4) * (n - ecs) + 64)
        bad("truncated stream (fewer entropy-coded bytes than the frame header needs)");
    f.coef.assign(total, 0);
    if (f.progressive) {  // every scan adds a band, or a bit, to the coefficients
        // jdphuff.c start_pass_phuff_decoder: coef_bits[component][k] = the Al of the last scan that carried coefficient k (-1: never sent)
        int coef_bits[3][64];
        for (auto& cb : coef_bits)
            for (int& v : cb) v = -1;
        auto note_scan = [&](const ScanHdr& h) {
            for (int s = 0; s < h.nc; s++)
                for (int k = h.Ss; k <= h.Se; k++) coef_bits[h.ci[s]][k] = h.Al;
        };
        note_scan(sh);
        size_t pos = prog_scan(data, n, ecs, sh, p, f);
        for (int scans = 1; scans < 1024; scans++) {
            const size_t next = p.headers(false, std::max<size_t>(pos, 3), &sh);
            if (!next) break;
            note_scan(sh);
            pos = prog_scan(data, n, next, sh, p, f);
        }
        // An INCOMPLETE progressive stream (scans missing: the first AC coefficients never sent or not refined to their last bit) is
        // not decoded the plain way by libjpeg: with every component's DC known, jdcoefct.c smoothing_ok() turns on inter-block
        // smoothing (decompress_smooth_data estimates the missing low AC terms from the neighbours' DC values).  That pass is not
        // restated here, so such a stream is refused and the caller falls back to imdecode — silently
        // different pixels would be worse.  (AC coefficients 1..9: libjpeg-turbo >= 2.1; older versions look at 1..5.)
        bool dc_known = true, low_ac_open = false;
        for (int c = 0; c < f.ncomp; c++) {
            dc_known &= coef_bits[c][0] >= 0;
            for (int k = 1; k <= 9; k++) low_ac_open |= coef_bits[c][k] != 0;
        }
        if (dc_known && low_ac_open) bad("incomplete progressive stream (libjpeg would apply inter-block smoothing, which is not implemented)");
        return;
    }
    Bits br{data + ecs, data + n};
    int pred[3] = {0, 0, 0};
    int mcus = 0;
    for (int my = 0; my < mcuy; my++)
        for (int mx = 0; mx < mcux; mx++) {
            if (p.restart_interval && mcus > 0 && mcus % p.restart_interval == 0) {
                br.restart();
                pred[0] = pred[1] = pred[2] = 0;
            }
            mcus++;
            if (br.insufficient) continue;  // jdhuff.c decode_mcu: out of data -> "leave the MCU set to zeroes" until a restart marker is found
            for (int c = 0; c < f.ncomp; c++) {
                const JpegComponent& cp = f.comp[c];
                const Huff &hd = p.dc[cp.td], &ha = p.ac[cp.ta];
                for (int by = 0; by < cp.v; by++)
                    for (int bx = 0; bx < cp.h; bx++) {
                        int16_t* blk = f.coef.data() + cp.coef_off + (static_cast<size_t>(my * cp.v + by) * cp.bw + mx * cp.h + bx) * 64;
                        const int t = br.symbol(hd) & 15;
                        if (t) pred[c] += extend(br.get(t), t);
                        blk[0] = static_cast<int16_t>(pred[c]);
                        for (int k = 1; k < 64;) {
                            const int rs = br.symbol(ha), r = rs >> 4, s = rs & 15;
                            if (s == 0) {
                                if (r != 15) break;
                                k += 16;
                                continue;
                            }
                            k += r;
                            if (k > 63) break;
                            blk[kZigzag[k]] = static_cast<int16_t>(extend(br.get(s), s));
                            k++;
                        }
                    }
            }
        }
}

}  // namespace mi
