/* ORACLE (test infrastructure, never shipped, never on the product path).
 *
 * Sequential Huffman entropy decoding of ONE baseline JPEG scan (ITU-T T.81 Annex F.2.2, the algorithm libjpeg's
 * jdhuff.c implements): what sits between the file bytes and the quantised DCT coefficients when
 * cv2.VideoCapture.read / cv2.imread decode a frame in the reference (playaid/ai_runner.py:153,404-405,446;
 * playaid/manuscript.py:154-155). Plain C because a 1080p frame holds ~10^6 Huffman symbols -- far too many for a
 * Python loop; the arithmetic that follows (de-quantisation, integer IDCT, up-sampling, colour conversion) stays in
 * oracle/jpeg.py. Deliberately the textbook bit-serial form (DECODE procedure of figure F.16: one bit at a time
 * against MAXCODE / VALPTR), nothing shared with the HIP kernels' table-driven reader. Entropy decoding is lossless
 * and unambiguous, so the pin is on the whole decoder: oracle/jpeg.py::decode equals PIL.Image.open (live
 * libjpeg-turbo) byte for byte, tests/test_oracle_jpeg.py.
 *
 * Built on demand by oracle/_cbuild.py (gcc -O2 -shared) into oracle/_build/.
 */
#include <stddef.h>
#include <stdint.h>
#include <string.h>

typedef struct {
    int mincode[17], maxcode[18], valptr[17];
    uint8_t vals[256];
} htab;

static void build_table(htab* t, const uint8_t* counts, const uint8_t* syms) {
    /* Annex C: canonical code assignment; F.2.2.3: decoder tables */
    int code = 0, k = 0;
    for (int l = 1; l <= 16; ++l) {
        t->valptr[l] = k;
        t->mincode[l] = code;
        k += counts[l - 1];
        code += counts[l - 1];
        t->maxcode[l] = counts[l - 1] ? code - 1 : -1;
        code <<= 1;
    }
    t->maxcode[17] = 0x7fffffff;
    memcpy(t->vals, syms, 256);
}

typedef struct {
    const uint8_t* p;
    const uint8_t* end;
    uint32_t acc;
    int nbits;
    int marker; /* a marker was reached: feed zeros (T.81 F.2.2.5) */
} bitreader;

static int next_bit(bitreader* b) {
    if (b->nbits == 0) {
        int c = 0;
        if (!b->marker && b->p < b->end) {
            c = *b->p++;
            if (c == 0xff) {
                int c2 = b->p < b->end ? *b->p : 0xd9;
                if (c2 == 0) {
                    b->p++; /* stuffed zero */
                } else {
                    b->p--; /* leave the marker in place */
                    b->marker = 1;
                    c = 0;
                }
            }
        } else {
            b->marker = 1;
        }
        b->acc = (uint32_t)c;
        b->nbits = 8;
    }
    b->nbits--;
    return (int)((b->acc >> b->nbits) & 1u);
}

static int receive(bitreader* b, int n) {
    int v = 0;
    while (n--) v = (v << 1) | next_bit(b);
    return v;
}

static int extend(int v, int t) { return t == 0 ? 0 : (v < (1 << (t - 1)) ? v - (1 << t) + 1 : v); }

static int decode_symbol(bitreader* b, const htab* t) {
    int code = next_bit(b), l = 1;
    while (l <= 16 && code > t->maxcode[l]) {
        code = (code << 1) | next_bit(b);
        ++l;
    }
    if (l > 16) return -1;
    return t->vals[t->valptr[l] + code - t->mincode[l]];
}

static const uint8_t zigzag[64] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48,
                                   41, 34, 27, 20, 13, 6,  7,  14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23,
                                   30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

/* scan / scan_len: the entropy-coded segment (from the byte after the SOS header up to, at most, the end of the
 * file). ncomp components with sampling factors hs[c] x vs[c] and table selectors dc_tbl[c] / ac_tbl[c]; dht_counts /
 * dht_syms: [class 0 = DC, 1 = AC][table id 0..3][16 | 256]. The scan is interleaved (ncomp > 1) or holds one
 * component whose MCU is one block (ncomp == 1, T.81 A.2.2). coef: per component c a raster of blocks
 * [blocks_y[c]][blocks_x[c]][64] in NATURAL order at int16 offset comp_off[c]; blocks_x / blocks_y are the padded
 * counts (mcus * hs) for interleaved scans. Returns 0, or a negative code: -1 bad Huffman code, -2 missing / wrong
 * restart marker, -3 coefficient index overflow. *consumed receives the number of scan bytes read. */
int pa_oracle_jpeg_decode_scan(const uint8_t* scan, size_t scan_len, int ncomp, const int* hs, const int* vs, const int* dc_tbl,
                               const int* ac_tbl, const uint8_t* dht_counts, const uint8_t* dht_syms, int mcus_x, int mcus_y,
                               int restart_interval, int16_t* coef, const int64_t* comp_off, const int* blocks_x,
                               size_t* consumed) {
    htab tabs[2][4];
    for (int cls = 0; cls < 2; ++cls)
        for (int id = 0; id < 4; ++id) build_table(&tabs[cls][id], dht_counts + (cls * 4 + id) * 16, dht_syms + (cls * 4 + id) * 256);
    bitreader br = {scan, scan + scan_len, 0, 0, 0};
    int pred[4] = {0, 0, 0, 0};
    int todo = restart_interval, next_rst = 0;
    for (int my = 0; my < mcus_y; ++my)
        for (int mx = 0; mx < mcus_x; ++mx) {
            if (restart_interval && todo == 0) {
                /* F.2.2.4 / E.2.4: discard the partial byte, expect RSTm, reset the predictions */
                br.nbits = 0;
                br.marker = 0;
                if (br.p + 2 > br.end || br.p[0] != 0xff || br.p[1] != (0xd0 + next_rst)) return -2;
                br.p += 2;
                next_rst = (next_rst + 1) & 7;
                memset(pred, 0, sizeof pred);
                todo = restart_interval;
            }
            for (int c = 0; c < ncomp; ++c) {
                const int nh = ncomp == 1 ? 1 : hs[c], nv = ncomp == 1 ? 1 : vs[c];
                for (int v = 0; v < nv; ++v)
                    for (int h = 0; h < nh; ++h) {
                        int16_t* blk = coef + comp_off[c] + ((int64_t)(my * nv + v) * blocks_x[c] + (mx * nh + h)) * 64;
                        memset(blk, 0, 64 * sizeof(int16_t));
                        int t = decode_symbol(&br, &tabs[0][dc_tbl[c]]);
                        if (t < 0 || t > 11) return -1;
                        pred[c] += extend(receive(&br, t), t);
                        blk[0] = (int16_t)pred[c];
                        for (int k = 1; k < 64;) {
                            int rs = decode_symbol(&br, &tabs[1][ac_tbl[c]]);
                            if (rs < 0) return -1;
                            int r = rs >> 4, s = rs & 15;
                            if (s == 0) {
                                if (r != 15) break; /* EOB */
                                k += 16;            /* ZRL */
                                continue;
                            }
                            k += r;
                            if (k > 63) return -3;
                            blk[zigzag[k]] = (int16_t)extend(receive(&br, s), s);
                            ++k;
                        }
                    }
            }
            if (restart_interval) --todo;
        }
    if (consumed) *consumed = (size_t)(br.p - scan);
    return 0;
}
