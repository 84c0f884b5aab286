/*
 * iiv_oracle.c -- CPU restatement ("oracle") of the ][-Vision transcode hot path.
 *
 * TEST INFRASTRUCTURE ONLY (see iiv_oracle.h).  Plain C, gcc, optional OpenMP
 * for the table build.  Citations are reference file:line under transcoder/.
 *
 * PARITY UNPINNED for the CIE2000 matrix / Damerau-Levenshtein *values*: they
 * restate colormath==3.0.0 and weighted-levenshtein==0.2.2 (requirements.txt:6,32),
 * neither of which is vendored in the reference or installable here.  Everything
 * else is pinned by tests/golden (generated from the imported reference).
 */
#include "iiv_oracle.h"

#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------- */
/* mode constants                                                             */
/* ------------------------------------------------------------------------- */

int orc_masked_bits(int mode) { return mode == ORC_DHGR ? 13 : 14; } /* screen.py:615,886 */
int orc_masked_dots(int mode) { return mode == ORC_DHGR ? 10 : 18; } /* screen.py:624,890 */
int orc_num_offsets(int mode) { return mode == ORC_DHGR ? 4 : 2; }   /* len(BYTE_MASKS)  */

int orc_phase(int mode, int byte_offset)
{
    static const int hgr[2] = {1, 3};        /* screen.py:645 */
    static const int dhgr[4] = {1, 0, 3, 2}; /* screen.py:919 */
    return mode == ORC_DHGR ? dhgr[byte_offset] : hgr[byte_offset];
}

size_t orc_table_entries(int mode)
{
    int bits = orc_masked_bits(mode);
    return (size_t)orc_num_offsets(mode) << (2 * bits);
}

/* ------------------------------------------------------------------------- */
/* geometry (screen.py:16-69)                                                 */
/* ------------------------------------------------------------------------- */

int orc_y_to_base_addr(int y, int page)
{
    int a = y / 64;
    int d = y - 64 * a;
    int b = d / 8;
    int c = d - 8 * b;
    return 8192 * (page + 1) + 1024 * c + 128 * b + 40 * a; /* screen.py:23 */
}

void orc_screen_holes(uint8_t holes[8192])
{
    memset(holes, 1, 8192); /* screen.py:42 */
    for (int y = 0; y < 192; y++)
        for (int x = 0; x < 40; x++) {
            int y_base = orc_y_to_base_addr(y, 0);
            int page = y_base >> 8;
            int offset = y_base - (page << 8) + x;
            holes[(page - 32) * 256 + offset] = 0; /* screen.py:62 */
        }
}

void orc_xy_tables(uint8_t x_y_to_page[192 * 40], uint8_t x_y_to_offset[192 * 40])
{
    for (int y = 0; y < 192; y++)
        for (int x = 0; x < 40; x++) {
            int y_base = orc_y_to_base_addr(y, 0);
            int page = y_base >> 8;
            int offset = y_base - (page << 8) + x;
            x_y_to_page[y * 40 + x] = (uint8_t)(page - 32);   /* screen.py:58 */
            x_y_to_offset[y * 40 + x] = (uint8_t)offset;      /* screen.py:59 */
        }
}

/* ------------------------------------------------------------------------- */
/* packed representation                                                      */
/* ------------------------------------------------------------------------- */

uint64_t orc_make_header(int mode, uint64_t col)
{
    if (mode == ORC_DHGR)
        return (col & ((uint64_t)7 << 28)) >> 28; /* screen.py:924 */
    /* screen.py:658-661 */
    return ((col & ((uint64_t)1 << 11)) >> 9) ^ ((col & ((uint64_t)3 << 17)) >> 17);
}

uint64_t orc_make_footer(int mode, uint64_t col)
{
    if (mode == ORC_DHGR)
        return (col & ((uint64_t)7 << 3)) << 28; /* screen.py:952 */
    /* screen.py:687-690 */
    return (((col & ((uint64_t)1 << 10)) >> 10) ^ ((col & ((uint64_t)3 << 3)) >> 2)) << 19;
}

static uint64_t body_of(int mode, const uint8_t *main_mem, const uint8_t *aux_mem, int page, int col)
{
    const uint8_t *m = main_mem + page * 256;
    if (mode == ORC_DHGR) {
        /* screen.py:939-947 */
        const uint8_t *a = aux_mem + page * 256;
        uint64_t a0 = a[2 * col] & 0x7f, m0 = m[2 * col] & 0x7f;
        uint64_t a1 = a[2 * col + 1] & 0x7f, m1 = m[2 * col + 1] & 0x7f;
        return (a0 << 3) + (m0 << 10) + (a1 << 17) + (m1 << 24);
    }
    /* screen.py:672-677 */
    uint64_t even = m[2 * col], odd = m[2 * col + 1];
    return (even << 3) + ((odd & 0x7f) << 12) + ((odd & 0x80) << 4);
}

/* Bitmap._pack (screen.py:207-226): header from column c-1 (np.roll +1),
 * footer from column c+1 (np.roll -1); only column 0's header and column 127's
 * footer are forced to 0. */
void orc_pack(int mode, const uint8_t *main_mem, const uint8_t *aux_mem, uint64_t packed[4096])
{
    for (int p = 0; p < 32; p++) {
        uint64_t body[128];
        for (int c = 0; c < 128; c++)
            body[c] = body_of(mode, main_mem, aux_mem, p, c);
        for (int c = 0; c < 128; c++) {
            uint64_t header = c == 0 ? 0 : orc_make_header(mode, body[c - 1]);
            uint64_t footer = c == 127 ? 0 : orc_make_footer(mode, body[c + 1]);
            packed[p * 128 + c] = header ^ body[c] ^ footer;
        }
    }
}

static uint64_t byte_mask(int mode, int o)
{
    if (mode == ORC_DHGR) /* screen.py:894-907 */
        return (uint64_t)0x1fff << (7 * o);
    return o == 0 ? 0x3fffull : 0x3fff00ull; /* screen.py:632-635 */
}

static int byte_shift(int mode, int o)
{
    if (mode == ORC_DHGR)
        return 7 * o;      /* screen.py:910 */
    return o == 0 ? 0 : 8; /* screen.py:636 */
}

uint64_t orc_mask_and_shift(int mode, uint64_t packed, int byte_offset)
{
    return (packed & byte_mask(mode, byte_offset)) >> byte_shift(mode, byte_offset); /* screen.py:375 */
}

uint64_t orc_masked_update(int mode, int byte_offset, uint64_t old_value, uint8_t new_value)
{
    if (mode == ORC_DHGR) {
        /* screen.py:1001-1007 */
        int sh = 7 * byte_offset + 3;
        uint64_t masked = old_value & ~((uint64_t)0x7f << sh);
        return masked ^ ((uint64_t)(new_value & 0x7f) << sh);
    }
    if (byte_offset == 0) {
        /* screen.py:801-805 */
        uint64_t masked = old_value & ~((uint64_t)0xff << 3);
        return masked ^ ((uint64_t)new_value << 3);
    }
    /* screen.py:807-816 */
    uint64_t masked = old_value & ~((uint64_t)0xff << 11);
    uint64_t shifted = (uint64_t)(((new_value & 0x7f) << 1) ^ ((new_value & 0x80) >> 7));
    return masked ^ (shifted << 11);
}

int orc_byte_offset(int mode, int page_offset, int is_aux)
{
    int is_odd = page_offset % 2 == 1;
    if (mode == ORC_DHGR) /* screen.py:956-969 */
        return is_aux ? (is_odd ? 2 : 0) : (is_odd ? 3 : 1);
    return is_odd ? 1 : 0; /* screen.py:694-700 */
}

static void byte_offsets(int mode, int is_aux, int out[2])
{
    if (mode == ORC_DHGR) { /* screen.py:973-980 */
        out[0] = is_aux ? 0 : 1;
        out[1] = is_aux ? 2 : 3;
    } else { /* screen.py:704-708 */
        out[0] = 0;
        out[1] = 1;
    }
}

static int header_bits(int mode) { (void)mode; return 3; }
static int body_bits(int mode) { return mode == ORC_DHGR ? 28 : 16; }
static int footer_bits(int mode) { (void)mode; return 3; }

/* screen.py:295-307 */
static uint64_t fix_column_left(int mode, uint64_t column_left, uint64_t column)
{
    column_left &= (((uint64_t)1 << (header_bits(mode) + body_bits(mode))) - 1);
    column_left ^= orc_make_footer(mode, column);
    return column_left;
}

/* screen.py:309-320 */
static uint64_t fix_column_right(int mode, uint64_t column_right, uint64_t column)
{
    column_right &= ((((uint64_t)1 << (body_bits(mode) + footer_bits(mode))) - 1) << header_bits(mode));
    column_right ^= orc_make_header(mode, column);
    return column_right;
}

/* Bitmap.apply + _fix_scalar_neighbours (screen.py:256-293). MemoryMap.write is
 * called with page in 0..31 and relies on negative-index wraparound
 * (screen.py:125), i.e. it lands in row `page`. */
void orc_apply(int mode, uint64_t packed[4096], uint8_t *main_mem, uint8_t *aux_mem,
               int page, int offset, int is_aux, uint8_t value)
{
    int bo = orc_byte_offset(mode, offset, is_aux);
    int po = offset / 2;
    int screen_bytes = orc_num_offsets(mode);
    uint64_t *row = packed + page * 128;
    row[po] = orc_masked_update(mode, bo, row[po], value);
    if (bo == 0 && po > 0)
        row[po - 1] = fix_column_left(mode, row[po - 1], row[po]);
    else if (bo == screen_bytes - 1 && po < 127)
        row[po + 1] = fix_column_right(mode, row[po + 1], row[po]);
    if (is_aux)
        aux_mem[page * 256 + offset] = value;
    else
        main_mem[page * 256 + offset] = value;
}

/* ------------------------------------------------------------------------- */
/* colour model                                                               */
/* ------------------------------------------------------------------------- */

/* HGRBitmap._double_pixels (screen.py:712-739) */
uint32_t orc_double_pixels(uint32_t v)
{
    return ((v & 0x40) << 8) + ((v & 0x40) << 7) + ((v & 0x40) << 6) +
           ((v & 0x20) << 6) + ((v & 0x20) << 5) +
           ((v & 0x10) << 5) + ((v & 0x10) << 4) +
           ((v & 0x08) << 4) + ((v & 0x08) << 3) +
           ((v & 0x04) << 3) + ((v & 0x04) << 2) +
           ((v & 0x02) << 2) + ((v & 0x02) << 1) +
           ((v & 0x01) << 1) + (v & 0x01);
}

/* HGRBitmap.to_dots (screen.py:743-789); DHGRBitmap.to_dots is the identity
 * (screen.py:983-990). */
uint32_t orc_to_dots(int mode, uint32_t m, int byte_offset)
{
    if (mode == ORC_DHGR)
        return m;
    uint32_t h = (m & 7) << 5;
    uint32_t hp = (h & 0x80) >> 7;
    uint32_t res = orc_double_pixels(h & 0x7f) >> (11 - hp);
    uint32_t b, bp;
    if (byte_offset == 0) {
        b = (m >> 3) & 0xff;
        bp = (b & 0x80) >> 7;
    } else {
        bp = (m >> 3) & 1;
        b = ((m >> 4) & 0x7f) ^ (bp << 7);
    }
    res &= ~((uint32_t)0x3fff << (3 + bp));
    res ^= orc_double_pixels(b & 0x7f) << (3 + bp);
    uint32_t f = ((m >> 12) & 3) ^ (((m >> 11) & 1) << 7);
    uint32_t fp = (f & 0x80) >> 7;
    res &= ~((uint32_t)0xf << (17 + fp));
    res ^= orc_double_pixels(f & 0x7f) << (17 + fp);
    return res & ((1u << 21) - 1);
}

static inline uint32_t rol4(uint32_t v, int r) /* colours.py:87-97 */
{
    r &= 3;
    return ((v << r) | (v >> (4 - r))) & 0xf;
}

/* colours.dots_to_nominal_colour_pixel_values (colours.py:100-148).  The enum
 * round trip colours(colour).value is the identity on 0..15. */
void orc_dots_to_pixel_values(int num_bits, uint32_t dots, int init_phase, uint8_t *out)
{
    uint32_t shifted = dots;
    int phase = init_phase;
    for (int i = 0; i < num_bits; i++) {
        out[i] = (uint8_t)rol4(shifted & 0xf, phase);
        shifted >>= 1;
        phase += 1;
        if (phase == 4)
            phase = 0;
    }
}

/* make_data_tables.py:147-152 */
void orc_pixel_values(int mode, uint32_t masked_val, int byte_offset, uint8_t *out)
{
    uint32_t dots = orc_to_dots(mode, masked_val, byte_offset);
    orc_dots_to_pixel_values(orc_masked_dots(mode), dots, orc_phase(mode, byte_offset), out);
}

/* ------------------------------------------------------------------------- */
/* CIE2000 (restating colormath 3.0.0 -- PARITY UNPINNED)                      */
/* ------------------------------------------------------------------------- */

/* colormath.color_conversions.RGB_to_XYZ + XYZ_to_Lab for an upscaled
 * sRGBColor, D65/2deg, no chromatic adaptation.  Matrix constants are
 * colormath's low-precision sRGB "rgb_to_xyz" matrix. */
void orc_rgb_to_lab(const uint8_t rgb[3], double lab[3])
{
    double lin[3];
    for (int i = 0; i < 3; i++) {
        double v = rgb[i] / 255.0;
        lin[i] = v <= 0.04045 ? v / 12.92 : pow((v + 0.055) / 1.055, 2.4);
    }
    static const double M[3][3] = {{0.412424, 0.357579, 0.180464},
                                   {0.212656, 0.715158, 0.0721856},
                                   {0.0193324, 0.119193, 0.950444}};
    double xyz[3];
    for (int r = 0; r < 3; r++) {
        double s = M[r][0] * lin[0] + M[r][1] * lin[1] + M[r][2] * lin[2];
        xyz[r] = s > 0.0 ? s : 0.0;
    }
    static const double illum[3] = {0.95047, 1.00000, 1.08883};
    const double CIE_E = 216.0 / 24389.0;
    double t[3];
    for (int i = 0; i < 3; i++) {
        double v = xyz[i] / illum[i];
        t[i] = v > CIE_E ? pow(v, 1.0 / 3.0) : (7.787 * v) + (16.0 / 116.0);
    }
    lab[0] = (116.0 * t[1]) - 16.0;
    lab[1] = 500.0 * (t[0] - t[1]);
    lab[2] = 200.0 * (t[1] - t[2]);
}

static inline double deg(double r) { return r * (180.0 / M_PI); }
static inline double rad(double d) { return d * (M_PI / 180.0); }

/* colormath.color_diff_matrix.delta_e_cie2000, Kl=Kc=Kh=1, scalar form. */
double orc_delta_e_cie2000(const double c1[3], const double c2[3])
{
    double L = c1[0], a = c1[1], b = c1[2];
    double L2 = c2[0], a2 = c2[1], b2 = c2[2];
    double avg_Lp = (L + L2) / 2.0;
    double C1 = sqrt(a * a + b * b);
    double C2 = sqrt(a2 * a2 + b2 * b2);
    double avg_C1_C2 = (C1 + C2) / 2.0;
    double G = 0.5 * (1 - sqrt(pow(avg_C1_C2, 7.0) / (pow(avg_C1_C2, 7.0) + pow(25.0, 7.0))));
    double a1p = (1.0 + G) * a;
    double a2p = (1.0 + G) * a2;
    double C1p = sqrt(a1p * a1p + b * b);
    double C2p = sqrt(a2p * a2p + b2 * b2);
    double avg_C1p_C2p = (C1p + C2p) / 2.0;
    double h1p = deg(atan2(b, a1p));
    h1p += (h1p < 0) * 360;
    double h2p = deg(atan2(b2, a2p));
    h2p += (h2p < 0) * 360;
    double avg_Hp = (((fabs(h1p - h2p) > 180) * 360) + h1p + h2p) / 2.0;
    double T = 1 - 0.17 * cos(rad(avg_Hp - 30)) + 0.24 * cos(rad(2 * avg_Hp)) +
               0.32 * cos(rad(3 * avg_Hp + 6)) - 0.2 * cos(rad(4 * avg_Hp - 63));
    double diff_h2p_h1p = h2p - h1p;
    double delta_hp = diff_h2p_h1p + (fabs(diff_h2p_h1p) > 180) * 360;
    delta_hp -= (h2p > h1p) * 720;
    double delta_Lp = L2 - L;
    double delta_Cp = C2p - C1p;
    double delta_Hp = 2 * sqrt(C2p * C1p) * sin(rad(delta_hp) / 2.0);
    double S_L = 1 + ((0.015 * pow(avg_Lp - 50, 2)) / sqrt(20 + pow(avg_Lp - 50, 2.0)));
    double S_C = 1 + 0.045 * avg_C1p_C2p;
    double S_H = 1 + 0.015 * avg_C1p_C2p * T;
    double delta_ro = 30 * exp(-(pow(((avg_Hp - 275) / 25), 2.0)));
    double R_C = sqrt((pow(avg_C1p_C2p, 7.0)) / (pow(avg_C1p_C2p, 7.0) + pow(25.0, 7.0)));
    double R_T = -2 * R_C * sin(2 * rad(delta_ro));
    return sqrt(pow(delta_Lp / S_L, 2) + pow(delta_Cp / S_C, 2) + pow(delta_Hp / S_H, 2) +
                R_T * (delta_Cp / S_C) * (delta_Hp / S_H));
}

/* make_data_tables.compute_diff_matrix (make_data_tables.py:55-70); rgb row i
 * is the palette entry whose HGRColours value is i (palette.py:37-78). */
void orc_cie2000_matrix(const uint8_t rgb[48], double out_f[256], int32_t out_i[256])
{
    double lab[16][3];
    for (int i = 0; i < 16; i++)
        orc_rgb_to_lab(rgb + 3 * i, lab[i]);
    for (int i = 0; i < 16; i++)
        for (int j = 0; j < 16; j++) {
            double d = orc_delta_e_cie2000(lab[i], lab[j]);
            if (out_f)
                out_f[i * 16 + j] = d;
            if (out_i)
                out_i[i * 16 + j] = (int32_t)d; /* int(): truncation */
        }
}

/* compute_substitute_costs (make_data_tables.py:73-89): the loop writes (c,d)
 * and (d,c) each iteration, so the final matrix is dm's lower triangle mirrored:
 * sub[u][v] = dm[max(u,v)][min(u,v)]. */
void orc_substitute_costs(const int32_t dm[256], int32_t sub[256])
{
    for (int i = 0; i < 16; i++)
        for (int j = 0; j < 16; j++) {
            int32_t cost = dm[i * 16 + j];
            sub[i * 16 + j] = cost;
            sub[j * 16 + i] = cost;
        }
}

/* ------------------------------------------------------------------------- */
/* edit distance                                                              */
/* ------------------------------------------------------------------------- */

/* dam_lev(a, b, ins=1e5, del=1e5, substitute=sub, transpose=1) for equal-length
 * strings whose substitution-only cost is << 2e5 reduces to this 1-D DP
 * (make_data_tables.py:92-108; reduction checked against orc_dam_lev_full). */
uint32_t orc_edit_distance(const int32_t sub[256], const uint8_t *a, const uint8_t *b, int n)
{
    uint32_t e2 = 0, e1 = 0; /* E[k-2], E[k-1] */
    for (int k = 0; k < n; k++) {
        uint32_t s = a[k] == b[k] ? 0u : (uint32_t)sub[a[k] * 16 + b[k]];
        uint32_t e = e1 + s;
        if (k >= 1 && a[k - 1] == b[k] && a[k] == b[k - 1]) {
            uint32_t t = e2 + 1;
            if (t < e)
                e = t;
        }
        e2 = e1;
        e1 = e;
    }
    return e1;
}

/* weighted-levenshtein 0.2.2 c_damerau_levenshtein (Lowrance-Wagner with
 * per-character costs), restated from its published algorithm. */
double orc_dam_lev_full(const int32_t sub[256], const uint8_t *s1, int len1, const uint8_t *s2, int len2)
{
    const double INS = 100000.0, DEL = 100000.0, TRANS = 1.0, BIG = 1e300;
    int W = len2 + 2;
    double *d = (double *)malloc(sizeof(double) * (size_t)(len1 + 2) * (size_t)W);
    int da[16];
#define D(i, j) d[((i) + 1) * W + ((j) + 1)]
    for (int i = 0; i < 16; i++)
        da[i] = 0;
    for (int i = -1; i <= len1; i++)
        D(i, -1) = BIG;
    for (int j = -1; j <= len2; j++)
        D(-1, j) = BIG;
    D(0, 0) = 0;
    for (int i = 1; i <= len1; i++)
        D(i, 0) = D(i - 1, 0) + DEL;
    for (int j = 1; j <= len2; j++)
        D(0, j) = D(0, j - 1) + INS;
    for (int i = 1; i <= len1; i++) {
        int ci = s1[i - 1];
        int db = 0;
        for (int j = 1; j <= len2; j++) {
            int cj = s2[j - 1];
            int k = da[cj];
            int l = db;
            double cost;
            if (ci == cj) {
                cost = 0;
                db = j;
            } else {
                cost = sub[ci * 16 + cj];
            }
            double best = D(i - 1, j - 1) + cost;
            double v = D(i, j - 1) + INS;
            if (v < best)
                best = v;
            v = D(i - 1, j) + DEL;
            if (v < best)
                best = v;
            if (k > 0 && l > 0) {
                /* delete s1[k+1..i-1], transpose, insert s2[l+1..j-1] */
                v = D(k - 1, l - 1) + (D(i - 1, 0) - D(k, 0)) + TRANS + (D(0, j - 1) - D(0, l));
                if (v < best)
                    best = v;
            }
            D(i, j) = best;
        }
        da[ci] = i;
    }
    double r = D(len1, len2);
#undef D
    free(d);
    return r;
}

/* compute_edit_distance (make_data_tables.py:111-174) */
void orc_build_table(int mode, const int32_t dm[256], uint16_t *out, int symmetric)
{
    int bits = orc_masked_bits(mode);
    int n = orc_masked_dots(mode);
    int noff = orc_num_offsets(mode);
    size_t range = (size_t)1 << bits;
    int32_t sub[256];
    orc_substitute_costs(dm, sub);
    memset(out, 0, orc_table_entries(mode) * sizeof(uint16_t));
    for (int o = 0; o < noff; o++) {
        uint8_t *pix = (uint8_t *)malloc(range * (size_t)n);
        for (size_t v = 0; v < range; v++)
            orc_pixel_values(mode, (uint32_t)v, o, pix + v * n);
        uint16_t *t = out + ((size_t)o << (2 * bits));
#pragma omp parallel for schedule(dynamic, 64)
        for (long i = 0; i < (long)range; i++) {
            const uint8_t *a = pix + (size_t)i * n;
            for (long j = 0; j < i; j++) {
                uint16_t e = (uint16_t)orc_edit_distance(sub, a, pix + (size_t)j * n, n);
                t[((size_t)i << bits) + (size_t)j] = e; /* make_data_tables.py:163-172 */
                if (symmetric)
                    t[((size_t)j << bits) + (size_t)i] = e; /* screen.py:358-365 */
            }
        }
        free(pix);
    }
}

/* ------------------------------------------------------------------------- */
/* table lookups                                                              */
/* ------------------------------------------------------------------------- */

/* Bitmap.byte_pair_difference (screen.py:383-398) */
uint16_t orc_byte_pair_difference(int mode, const uint16_t *table, int bo, uint64_t old_packed,
                                  uint8_t content)
{
    int bits = orc_masked_bits(mode);
    uint64_t old_pixels = orc_mask_and_shift(mode, old_packed, bo);
    uint64_t new_pixels = orc_mask_and_shift(mode, orc_masked_update(mode, bo, old_packed, content), bo);
    uint64_t pair = (old_pixels << bits) + new_pixels;
    return table[((size_t)bo << (2 * bits)) + pair];
}

/* Bitmap._diff_weights with content=None (screen.py:409-449) */
void orc_diff_weights(int mode, const uint16_t *table, const uint64_t src[4096],
                      const uint64_t tgt[4096], int is_aux, int32_t out[8192])
{
    int bits = orc_masked_bits(mode);
    int offs[2];
    byte_offsets(mode, is_aux, offs);
    for (int k = 0; k < 2; k++) {
        int o = offs[k];
        const uint16_t *t = table + ((size_t)o << (2 * bits));
        for (int p = 0; p < 32; p++)
            for (int c = 0; c < 128; c++) {
                uint64_t sp = orc_mask_and_shift(mode, src[p * 128 + c], o);
                uint64_t tp = orc_mask_and_shift(mode, tgt[p * 128 + c], o);
                uint64_t pair = (sp << bits) + tp;
                out[p * 256 + 2 * c + k] = t[pair]; /* screen.py:446-447 */
            }
    }
}

/* Bitmap.compute_delta_page -> _diff_weights_page(packed_page, packed_page,
 * is_aux, content) (screen.py:453-494, 525-547), including the
 * _fix_array_neighbours call (screen.py:322-341) exactly as written (it only
 * touches bits outside the mask that is then looked up). */
void orc_compute_delta_page(int mode, const uint16_t *table, const uint64_t tgt_packed[4096],
                            int page, uint8_t content, const int32_t dw_row[256], int is_aux,
                            int32_t out[256])
{
    int bits = orc_masked_bits(mode);
    int screen_bytes = orc_num_offsets(mode);
    const uint64_t *row = tgt_packed + page * 128;
    int offs[2];
    byte_offsets(mode, is_aux, offs);
    for (int k = 0; k < 2; k++) {
        int o = offs[k];
        const uint16_t *t = table + ((size_t)o << (2 * bits));
        uint64_t cmp[128];
        for (int c = 0; c < 128; c++)
            cmp[c] = orc_masked_update(mode, o, row[c], content);
        if (o == 0) {
            /* shifted_left = np.roll(ary, -1): element c sees c+1 (wrapping) */
            uint64_t fixed[128];
            for (int c = 0; c < 128; c++)
                fixed[c] = fix_column_left(mode, cmp[c], cmp[(c + 1) & 127]);
            memcpy(cmp, fixed, sizeof(cmp));
        } else if (o == screen_bytes - 1) {
            uint64_t fixed[128];
            for (int c = 0; c < 128; c++)
                fixed[c] = fix_column_right(mode, cmp[c], cmp[(c + 127) & 127]);
            memcpy(cmp, fixed, sizeof(cmp));
        }
        for (int c = 0; c < 128; c++) {
            uint64_t sp = orc_mask_and_shift(mode, cmp[c], o);
            uint64_t tp = orc_mask_and_shift(mode, row[c], o);
            uint64_t pair = (sp << bits) + tp;
            int y = 2 * c + k;
            out[y] = (int32_t)t[pair] - dw_row[y]; /* screen.py:547 */
        }
    }
}

/* ------------------------------------------------------------------------- */
/* MT19937                                                                    */
/* ------------------------------------------------------------------------- */

void orc_mt_init_genrand(orc_mt *m, uint32_t s)
{
    m->mt[0] = s;
    for (int i = 1; i < 624; i++)
        m->mt[i] = 1812433253u * (m->mt[i - 1] ^ (m->mt[i - 1] >> 30)) + (uint32_t)i;
    m->idx = 624;
}

void orc_mt_init_by_array(orc_mt *m, const uint32_t *key, int n)
{
    orc_mt_init_genrand(m, 19650218u);
    uint32_t *mt = m->mt;
    int i = 1, j = 0;
    int k = 624 > n ? 624 : n;
    for (; k; k--) {
        mt[i] = (mt[i] ^ ((mt[i - 1] ^ (mt[i - 1] >> 30)) * 1664525u)) + key[j] + (uint32_t)j;
        i++;
        j++;
        if (i >= 624) {
            mt[0] = mt[623];
            i = 1;
        }
        if (j >= n)
            j = 0;
    }
    for (k = 623; k; k--) {
        mt[i] = (mt[i] ^ ((mt[i - 1] ^ (mt[i - 1] >> 30)) * 1566083941u)) - (uint32_t)i;
        i++;
        if (i >= 624) {
            mt[0] = mt[623];
            i = 1;
        }
    }
    mt[0] = 0x80000000u;
    m->idx = 624;
}

/* random.seed(int): key = abs(seed) in 32-bit little-endian words, >= 1 word */
void orc_mt_seed_py(orc_mt *m, uint64_t seed)
{
    uint32_t key[2] = {(uint32_t)(seed & 0xffffffffu), (uint32_t)(seed >> 32)};
    orc_mt_init_by_array(m, key, key[1] ? 2 : 1);
}

/* np.random.seed(int) (legacy RandomState): init_genrand(seed) */
void orc_mt_seed_np(orc_mt *m, uint32_t seed) { orc_mt_init_genrand(m, seed); }

uint32_t orc_mt_next(orc_mt *m)
{
    uint32_t *mt = m->mt;
    if (m->idx >= 624) {
        int kk;
        uint32_t y;
        for (kk = 0; kk < 624 - 397; kk++) {
            y = (mt[kk] & 0x80000000u) | (mt[kk + 1] & 0x7fffffffu);
            mt[kk] = mt[kk + 397] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
        }
        for (; kk < 623; kk++) {
            y = (mt[kk] & 0x80000000u) | (mt[kk + 1] & 0x7fffffffu);
            mt[kk] = mt[kk + (397 - 624)] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
        }
        y = (mt[623] & 0x80000000u) | (mt[0] & 0x7fffffffu);
        mt[623] = mt[396] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
        m->idx = 0;
    }
    uint32_t y = mt[m->idx++];
    y ^= (y >> 11);
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= (y >> 18);
    return y;
}

uint32_t orc_py_getrandbits8(orc_mt *m) { return orc_mt_next(m) >> 24; }  /* video.py:178,291 */
uint32_t orc_np_randint256(orc_mt *m) { return orc_mt_next(m) & 0xffu; }  /* video.py:265    */

/* ------------------------------------------------------------------------- */
/* Video                                                                      */
/* ------------------------------------------------------------------------- */

typedef struct {
    int64_t key;
    int32_t nonce, page, off;
} hent;

static int hent_lt(const hent *a, const hent *b)
{
    if (a->key != b->key)
        return a->key < b->key;
    if (a->nonce != b->nonce)
        return a->nonce < b->nonce;
    if (a->page != b->page)
        return a->page < b->page;
    return a->off < b->off;
}

enum { FORM_NONE = 0, FORM_HEAP = 1, FORM_STRUCT = 2 };

struct orc_video {
    int mode;
    const uint16_t *table;
    uint8_t holes[8192];
    uint8_t mem[2][8192];    /* [is_aux]  video.py:38-42 */
    uint64_t packed[4096];   /* self.pixelmap.packed */
    int32_t up[2][8192];     /* [is_aux]  video.py:56-58 */
    int out_of_work[2];      /* video.py:62 */
    orc_mt rng_py, rng_np;
    uint64_t draws_py, draws_np;
    int joint;               /* f4: joint choice of the content byte (not the reference's behaviour) */
    int fourth;              /* f4: a real fourth offset per opcode (video.py:181 with 4 for 3; not the reference's behaviour) */
    /* diagnostic (tools/joint_prune_rate.py): what a pruning of the joint step's bytes would skip; see joint_prune_stats */
    int joint_stats_on;
    uint64_t joint_stats[6];
    /* generator */
    int gen_active, gen_started, gen_is_aux, gen_exhausted, gen_form;
    uint8_t tgt[2][8192];
    uint64_t tgt_packed[4096];
    int32_t dw[8192];
    hent *heap;
    int heap_n, heap_cap;
    uint64_t *sorted;
    int n_sorted, head;
    uint32_t *pushed;
    int n_pushed, pushed_cap;
};

orc_video *orc_video_create(int mode, const uint16_t *table)
{
    orc_video *v = (orc_video *)calloc(1, sizeof(orc_video));
    v->mode = mode;
    v->table = table;
    orc_screen_holes(v->holes);
    /* empty screen: packed of all-zero memory is all-zero */
    orc_mt_seed_py(&v->rng_py, 0);
    orc_mt_seed_np(&v->rng_np, 0);
    v->heap_cap = 8192 * 3 + 16;
    v->heap = (hent *)malloc(sizeof(hent) * (size_t)v->heap_cap);
    v->sorted = (uint64_t *)malloc(sizeof(uint64_t) * 8192);
    v->pushed_cap = 8192 * 3 + 16;
    v->pushed = (uint32_t *)malloc(sizeof(uint32_t) * (size_t)v->pushed_cap);
    return v;
}

void orc_video_destroy(orc_video *v)
{
    if (!v)
        return;
    free(v->heap);
    free(v->sorted);
    free(v->pushed);
    free(v);
}

void orc_video_set_joint(orc_video *v, int joint) { v->joint = joint ? 1 : 0; }
/* diagnostic: switch joint_prune_stats on / read its six counters (steps, eligible bytes, of those with dw == 0, bytes a
 * descending-dw walk looks at before its bound stops it, bytes behind the 16 largest, of those prunable by the bound reached there) */
void orc_video_joint_stats(orc_video *v, int on, uint64_t out[6])
{
    v->joint_stats_on = on ? 1 : 0;
    if (out)
        memcpy(out, v->joint_stats, sizeof(v->joint_stats));
}
/* f4 -- the opcode's fourth offset.  The player stores every opcode's content byte at FOUR offsets
 * (opcodes.py: tick opcodes), and video.py:146 says "Need to find 3 more offsets to fill this opcode", but the
 * loop's exit test `if len(offsets) == 3: break` (video.py:180-181) counts the primary: it stops after TWO more, and
 * :184-186 pad the fourth slot with a copy of the first.  With the flag set the test reads 4: up to three extra
 * offsets, each handled by the loop body exactly as the reference's two are (candidate order, one nonce per
 * candidate, priority 0 skipped, re-queued with a nonce if the store leaves an error).  NOT the reference's opcode
 * stream -- a quarter of every opcode's stores is no longer wasted.  Pinned against the reference itself with that
 * one literal changed (tests/golden/make_golden.py --fourth-only -> g8_fourth_offset.npz). */
void orc_video_set_fourth_offset(orc_video *v, int fourth) { v->fourth = fourth ? 1 : 0; }
orc_mt *orc_video_rng_py(orc_video *v) { return &v->rng_py; }
orc_mt *orc_video_rng_np(orc_video *v) { return &v->rng_np; }
uint8_t *orc_video_memory(orc_video *v, int is_aux) { return v->mem[is_aux ? 1 : 0]; }
int32_t *orc_video_update_priority(orc_video *v, int is_aux) { return v->up[is_aux ? 1 : 0]; }
uint64_t *orc_video_packed(orc_video *v) { return v->packed; }
int orc_video_out_of_work(orc_video *v, int is_aux) { return v->out_of_work[is_aux ? 1 : 0]; }
void orc_video_reset_out_of_work(orc_video *v) { v->out_of_work[0] = v->out_of_work[1] = 0; }
uint64_t orc_video_draws_py(orc_video *v) { return v->draws_py; }
uint64_t orc_video_draws_np(orc_video *v) { return v->draws_np; }

static uint32_t draw_py(orc_video *v)
{
    v->draws_py++;
    return orc_py_getrandbits8(&v->rng_py);
}

static uint32_t draw_np(orc_video *v)
{
    v->draws_np++;
    return orc_np_randint256(&v->rng_np);
}

/* encode_frame (video.py:72-93): lazy -- nothing runs until the first next() */
void orc_video_encode_frame(orc_video *v, const uint8_t *tgt_main, const uint8_t *tgt_aux, int is_aux)
{
    memcpy(v->tgt[0], tgt_main, 8192);
    if (v->mode == ORC_DHGR)
        memcpy(v->tgt[1], tgt_aux, 8192);
    else
        memset(v->tgt[1], 0, 8192);
    orc_pack(v->mode, v->tgt[0], v->tgt[1], v->tgt_packed);
    v->gen_active = 1;
    v->gen_started = 0;
    v->gen_is_aux = is_aux ? 1 : 0;
    v->gen_exhausted = 0;
    v->gen_form = FORM_NONE;
}

/* ---- heap helpers (heapq semantics; tuples are totally ordered) ---- */
static void heap_sift_up(hent *h, int i)
{
    while (i > 0) {
        int parent = (i - 1) / 2;
        if (!hent_lt(&h[i], &h[parent]))
            break;
        hent t = h[i];
        h[i] = h[parent];
        h[parent] = t;
        i = parent;
    }
}

static void heap_sift_down(hent *h, int n, int i)
{
    for (;;) {
        int l = 2 * i + 1, r = l + 1, m = i;
        if (l < n && hent_lt(&h[l], &h[m]))
            m = l;
        if (r < n && hent_lt(&h[r], &h[m]))
            m = r;
        if (m == i)
            break;
        hent t = h[i];
        h[i] = h[m];
        h[m] = t;
        i = m;
    }
}

static void heap_push(orc_video *v, hent e)
{
    if (v->heap_n >= v->heap_cap) {
        v->heap_cap *= 2;
        v->heap = (hent *)realloc(v->heap, sizeof(hent) * (size_t)v->heap_cap);
    }
    v->heap[v->heap_n] = e;
    heap_sift_up(v->heap, v->heap_n);
    v->heap_n++;
}

static hent heap_pop(orc_video *v)
{
    hent top = v->heap[0];
    v->heap_n--;
    if (v->heap_n > 0) {
        v->heap[0] = v->heap[v->heap_n];
        heap_sift_down(v->heap, v->heap_n, 0);
    }
    return top;
}

static int cmp_u64(const void *a, const void *b)
{
    uint64_t x = *(const uint64_t *)a, y = *(const uint64_t *)b;
    return x < y ? -1 : (x > y ? 1 : 0);
}

/* _index_changes prologue (video.py:104-119) + _heapify_priorities (:254-271) */
static int gen_prologue(orc_video *v, int form)
{
    int ia = v->gen_is_aux;
    int32_t *up = v->up[ia];
    /* video.py:87: nothing may have leaked into the screen holes */
    for (int i = 0; i < 8192; i++)
        if (v->holes[i] && v->mem[ia][i] != 0)
            return -1;
    orc_diff_weights(v->mode, v->table, v->packed, v->tgt_packed, ia, v->dw); /* :109 */
    for (int i = 0; i < 8192; i++)
        if (v->holes[i])
            v->dw[i] = 0; /* :111 */
    for (int i = 0; i < 8192; i++) {
        if (v->dw[i] == 0)
            up[i] = 0;      /* :115 */
        up[i] += v->dw[i];  /* :116 */
        if (up[i] < 0)
            return -2;      /* :117 */
    }
    int n = 0;
    for (int i = 0; i < 8192; i++)
        if (up[i] != 0)
            n++;
    v->gen_form = form;
    if (form == FORM_HEAP) {
        v->heap_n = 0;
        /* row-major nonzero(); one np.random.randint(0,256,size=n) */
        for (int i = 0; i < 8192; i++)
            if (up[i] != 0) {
                hent e;
                e.key = -(int64_t)up[i];
                e.nonce = (int32_t)draw_np(v);
                e.page = i >> 8;
                e.off = i & 255;
                heap_push(v, e); /* heapify == any valid heap; pop order is total */
            }
    } else {
        int k = 0;
        for (int i = 0; i < 8192; i++)
            if (up[i] != 0) {
                uint64_t nonce = draw_np(v);
                v->sorted[k++] = ((uint64_t)(0x7fffffff - up[i]) << 21) | (nonce << 13) | (uint64_t)i;
            }
        qsort(v->sorted, (size_t)n, sizeof(uint64_t), cmp_u64);
        v->n_sorted = n;
        v->head = 0;
        v->n_pushed = 0;
    }
    return 0;
}

typedef struct {
    int32_t delta, nonce, off;
} dent;

static int dent_cmp(const void *a, const void *b)
{
    const dent *x = (const dent *)a, *y = (const dent *)b;
    if (x->delta != y->delta)
        return x->delta < y->delta ? -1 : 1;
    if (x->nonce != y->nonce)
        return x->nonce < y->nonce ? -1 : 1;
    return x->off < y->off ? -1 : (x->off > y->off ? 1 : 0);
}

static void emit_pad(orc_video *v, uint8_t *out)
{
    /* video.py:249-251 */
    out[0] = 32;
    out[1] = v->tgt[v->gen_is_aux][0];
    out[2] = out[3] = out[4] = out[5] = 0;
}

/* f4 -- README.md:212-215 ("Global optimization": "the best value to store to minimize the total
 * error of 4 offsets may not even be any one of those target content bytes").  NOT reference
 * behaviour: the reference has no such mode, so this is the definition the kernels are tested
 * against, not a restatement.  With the flag set a step differs from video.py:121-187 in two
 * places only:
 *   :134  content = the byte value c that maximises
 *             R(c) = (dw[primary] - nd_c[primary]) - (d1 + d2),
 *         nd_c[y] = error of byte y after storing c (compute_delta_page's new_diff),
 *         d1, d2 = the two smallest negative deltas nd_c[y] - dw[y] over the other bytes of the page
 *         whose priority is non-zero (exactly the offsets _compute_error would hand out; 0 if absent);
 *         ties: the primary's target byte first, then the smallest c.  R(target byte) is what
 *         the reference's step removes by its own accounting (it scores a store against the
 *         target's neighbours, screen.py:542-545), so a joint step never removes less by that
 *         accounting; on the screen itself a single step may, a frame of them does not.
 *   :140  update_priority[page, offset] = nd_c[primary], the error the chosen byte leaves
 *         (0 when c is the target byte); it is not re-queued -- the next generator sees it.
 * Together with the fourth-offset flag (orc_video_set_fourth_offset) an opcode has three extra offsets, so R(c) takes
 * the THREE smallest negative deltas: R(c) = (dw[primary] - nd_c[primary]) - (d1 + d2 + d3).
 * Everything else (candidate order, nonce draws, re-queueing of the extra offsets) is the
 * reference's, applied to the chosen byte. */
/* Diagnostic, not part of any parity path: how many of a joint step's bytes could be PRUNED.  The joint score of a byte
 * value c takes its two (three) smallest negative deltas over the page's eligible bytes; delta(y, c) = nd - dw[y] >= -dw[y],
 * so a byte y cannot enter any value's top two (three) once every value's second (third) smallest delta is already <= -dw[y].
 * Walking the eligible bytes in DESCENDING dw, the walk can stop at the first byte for which that holds: everything behind it
 * is pruned.  Counted per step: eligible bytes (priority != 0, not the primary), of those with dw == 0 (no delta of theirs is
 * negative: prunable without any bound), bytes the walk looks at before it stops, and -- a weaker kernel-friendly form --
 * the bytes a walk in plain offset order could skip given the bound reached after a first pass over the 16 largest. */
static void joint_prune_stats(orc_video *v, int page, int offset, int ia)
{
    const int bits = orc_masked_bits(v->mode);
    const int32_t *up = v->up[ia];
    const int ncontent = v->mode == ORC_DHGR ? 128 : 256;
    const uint64_t *row = v->tgt_packed + page * 128;
    const int k_th = v->fourth ? 3 : 2;
    int order[256], n = 0, n_zero = 0;
    for (int y = 0; y < 256; y++) {
        if (y == offset || up[page * 256 + y] == 0)
            continue;
        if (v->dw[page * 256 + y] == 0)
            n_zero++;
        order[n++] = y;
    }
    for (int i = 1; i < n; i++) { /* insertion sort, dw descending */
        int y = order[i], j = i;
        while (j > 0 && v->dw[page * 256 + order[j - 1]] < v->dw[page * 256 + y]) {
            order[j] = order[j - 1];
            j--;
        }
        order[j] = y;
    }
    int32_t m[256][3];
    memset(m, 0, sizeof(m));
    int looked = 0, after16 = -1;
    for (int i = 0; i < n; i++) {
        int y = order[i];
        int32_t bound = INT32_MIN; /* the weakest k-th smallest delta over all byte values */
        for (int c = 0; c < ncontent; c++)
            if (m[c][k_th - 1] > bound)
                bound = m[c][k_th - 1];
        if (i == 16) { /* bytes (of all eligible) whose -dw is already >= the bound reached after the 16 largest */
            after16 = 0;
            for (int j = 16; j < n; j++)
                if (-v->dw[page * 256 + order[j]] >= bound)
                    after16++;
        }
        if (-v->dw[page * 256 + y] >= bound)
            break;
        looked++;
        int bo = orc_byte_offset(v->mode, y, ia);
        uint64_t t = orc_mask_and_shift(v->mode, row[y / 2], bo);
        for (int c = 0; c < ncontent; c++) {
            uint64_t sv = orc_mask_and_shift(v->mode, orc_masked_update(v->mode, bo, row[y / 2], (uint8_t)c), bo);
            int32_t d = (int32_t)v->table[((size_t)bo << (2 * bits)) + ((sv << bits) + t)] - v->dw[page * 256 + y];
            if (d >= 0)
                continue;
            if (d < m[c][0]) {
                m[c][2] = m[c][1];
                m[c][1] = m[c][0];
                m[c][0] = d;
            } else if (d < m[c][1]) {
                m[c][2] = m[c][1];
                m[c][1] = d;
            } else if (d < m[c][2]) {
                m[c][2] = d;
            }
        }
    }
    v->joint_stats[0] += 1;
    v->joint_stats[1] += (uint64_t)n;
    v->joint_stats[2] += (uint64_t)n_zero;
    v->joint_stats[3] += (uint64_t)looked;
    v->joint_stats[4] += (uint64_t)(n > 16 ? n - 16 : 0);
    v->joint_stats[5] += (uint64_t)(after16 > 0 ? after16 : 0);
}

static uint8_t choose_content_joint(orc_video *v, int page, int offset, int ia, int32_t *residual)
{
    if (v->joint_stats_on)
        joint_prune_stats(v, page, offset, ia);
    const int bits = orc_masked_bits(v->mode);
    const int32_t *up = v->up[ia];
    const int ncontent = v->mode == ORC_DHGR ? 128 : 256;
    const uint8_t tc = v->tgt[ia][page * 256 + offset];
    const uint64_t *row = v->tgt_packed + page * 128;
    int64_t best_key = INT64_MIN;
    int best_c = tc;
    int32_t best_res = 0;
    for (int c = 0; c < ncontent; c++) {
        int32_t m1 = 0, m2 = 0, m3 = 0, nd_primary = 0; /* m1 <= m2 <= m3 <= 0: the smallest negative deltas */
        for (int y = 0; y < 256; y++) {
            int bo = orc_byte_offset(v->mode, y, ia);
            uint64_t t = orc_mask_and_shift(v->mode, row[y / 2], bo);
            uint64_t s = orc_mask_and_shift(v->mode, orc_masked_update(v->mode, bo, row[y / 2], (uint8_t)c), bo);
            int32_t nd = v->table[((size_t)bo << (2 * bits)) + ((s << bits) + t)];
            if (y == offset) {
                nd_primary = nd;
                continue;
            }
            int32_t d = nd - v->dw[page * 256 + y];
            if (d >= 0 || up[page * 256 + y] == 0)
                continue;
            if (d < m1) {
                m3 = m2;
                m2 = m1;
                m1 = d;
            } else if (d < m2) {
                m3 = m2;
                m2 = d;
            } else if (d < m3) {
                m3 = d;
            }
        }
        int64_t r = (int64_t)v->dw[page * 256 + offset] - nd_primary - m1 - m2 - (v->fourth ? m3 : 0);
        int64_t key = r * 512 + (c == tc ? 256 : 0) + (255 - c);
        if (key > best_key) {
            best_key = key;
            best_c = c;
            best_res = nd_primary;
        }
    }
    *residual = best_res;
    return (uint8_t)best_c;
}

/* One greedy step, heap form: follows video.py:121-187 and :275-301 literally. */
static int step_heap(orc_video *v, uint8_t *out)
{
    int ia = v->gen_is_aux;
    int32_t *up = v->up[ia];
    const uint8_t *target = v->tgt[ia]; /* video.py:104-107 */
    while (v->heap_n > 0) {
        hent e = heap_pop(v); /* :122 */
        int page = e.page, offset = e.off;
        if (v->holes[page * 256 + offset])
            return -3; /* :124 */
        if (up[page * 256 + offset] == 0)
            continue; /* :130 */
        int offsets[4];
        int noffs = 0;
        offsets[noffs++] = offset;
        uint8_t content = target[page * 256 + offset]; /* :134 */
        if (v->mode == ORC_DHGR && content >= 0x80)
            return -4; /* :137 */
        int32_t residual = 0;
        if (v->joint)
            content = choose_content_joint(v, page, offset, ia, &residual);
        up[page * 256 + offset] = residual; /* :140 (0 unless joint) */
        v->dw[page * 256 + offset] = 0;     /* :141 */
        orc_apply(v->mode, v->packed, v->mem[0], v->mem[1], page, offset, ia, content); /* :144 */

        /* _compute_error (:275-301) */
        int32_t delta[256];
        orc_compute_delta_page(v->mode, v->table, v->tgt_packed, page, content, v->dw + page * 256, ia, delta);
        dent cand[256];
        int nc = 0;
        for (int o = 0; o < 256; o++)
            if (delta[o] < 0) {
                cand[nc].delta = delta[o];
                cand[nc].nonce = (int32_t)draw_py(v); /* :291, ascending offset */
                cand[nc].off = o;
                nc++;
            }
        qsort(cand, (size_t)nc, sizeof(dent), dent_cmp); /* heapify + pop ascending */
        for (int i = 0; i < nc; i++) {
            int o = cand[i].off;
            if (o == offset)
                return -5; /* :154 */
            if (v->holes[page * 256 + o])
                return -6; /* :155 */
            if (up[page * 256 + o] == 0)
                continue; /* :159 */
            int bo = orc_byte_offset(v->mode, o, ia);
            uint64_t old_packed = v->tgt_packed[page * 128 + o / 2];
            uint16_t p = orc_byte_pair_difference(v->mode, v->table, bo, old_packed, content); /* :166 */
            up[page * 256 + o] = p; /* :170 */
            orc_apply(v->mode, v->packed, v->mem[0], v->mem[1], page, o, ia, content); /* :172 */
            if (p) {
                hent ne;
                ne.key = (int64_t)(uint16_t)(0u - (uint32_t)p); /* -np.uint16(p) wraps: 65536-p */
                ne.nonce = (int32_t)draw_py(v);                  /* :178 */
                ne.page = page;
                ne.off = o;
                heap_push(v, ne);
            }
            offsets[noffs++] = o;
            if (noffs == (v->fourth ? 4 : 3))
                break; /* :181 */
        }
        for (; noffs < 4;)
            offsets[noffs++] = offsets[0]; /* :185 */
        out[0] = (uint8_t)(page + 32);
        out[1] = content;
        out[2] = (uint8_t)offsets[0];
        out[3] = (uint8_t)offsets[1];
        out[4] = (uint8_t)offsets[2];
        out[5] = (uint8_t)offsets[3];
        return 0;
    }
    v->out_of_work[ia] = 1; /* :189 */
    v->gen_exhausted = 1;
    emit_pad(v, out);
    return 0;
}

/* One greedy step, restructured form (what the HIP kernels implement):
 *  - initial entries come from a sorted list (all their keys are negative, so
 *    they precede every pushed entry, whose keys are 65536-p > 0);
 *  - pushed entries are an unsorted bag popped by arg-min;
 *  - a location whose priority is 0 can never become non-zero again inside one
 *    generator (secondaries require priority != 0), so lazy deletion is permanent;
 *  - the two extra offsets (three with the fourth-offset flag) are the smallest (delta, nonce, offset)
 *    among candidates whose priority is non-zero; every candidate draws a nonce. */
static int step_struct(orc_video *v, uint8_t *out)
{
    int ia = v->gen_is_aux;
    int bits = orc_masked_bits(v->mode);
    int32_t *up = v->up[ia];
    const uint8_t *target = v->tgt[ia];
    for (;;) {
        int page, offset;
        if (v->head < v->n_sorted) {
            uint64_t k = v->sorted[v->head++];
            page = (int)((k >> 8) & 31);
            offset = (int)(k & 255);
        } else {
            int best = -1;
            uint32_t bk = 0xffffffffu;
            for (int i = 0; i < v->n_pushed; i++)
                if (v->pushed[i] < bk) {
                    bk = v->pushed[i];
                    best = i;
                }
            if (best < 0)
                break;
            v->pushed[best] = 0xffffffffu;
            page = (int)((bk >> 8) & 31);
            offset = (int)(bk & 255);
        }
        if (up[page * 256 + offset] == 0)
            continue;
        uint8_t content = target[page * 256 + offset];
        if (v->mode == ORC_DHGR && content >= 0x80)
            return -4;
        int32_t residual = 0;
        if (v->joint)
            content = choose_content_joint(v, page, offset, ia, &residual);
        up[page * 256 + offset] = residual;
        v->dw[page * 256 + offset] = 0;
        orc_apply(v->mode, v->packed, v->mem[0], v->mem[1], page, offset, ia, content);

        /* nd[y] from the 3-byte target window, delta = nd - dw */
        uint32_t nd[256];
        const uint64_t *row = v->tgt_packed + page * 128;
        for (int y = 0; y < 256; y++) {
            int bo = orc_byte_offset(v->mode, y, ia);
            uint64_t t = orc_mask_and_shift(v->mode, row[y / 2], bo);
            uint64_t s = orc_mask_and_shift(v->mode, orc_masked_update(v->mode, bo, row[y / 2], content), bo);
            nd[y] = v->table[((size_t)bo << (2 * bits)) + ((s << bits) + t)];
        }
        uint32_t k1 = 0xffffffffu, k2 = 0xffffffffu, k3 = 0xffffffffu;
        for (int y = 0; y < 256; y++) {
            int32_t d = (int32_t)nd[y] - v->dw[page * 256 + y];
            if (d >= 0)
                continue;
            uint32_t nonce = draw_py(v);
            if (up[page * 256 + y] == 0)
                continue;
            uint32_t key = ((uint32_t)(d + 2048) << 16) | (nonce << 8) | (uint32_t)y;
            if (key < k1) {
                k3 = k2;
                k2 = k1;
                k1 = key;
            } else if (key < k2) {
                k3 = k2;
                k2 = key;
            } else if (key < k3) {
                k3 = key;
            }
        }
        int offs[4] = {offset, offset, offset, offset};
        uint32_t ks[3] = {k1, k2, k3};
        for (int i = 0; i < (v->fourth ? 3 : 2); i++) {
            if (ks[i] == 0xffffffffu)
                break;
            int y = (int)(ks[i] & 255);
            uint32_t q = nd[y];
            up[page * 256 + y] = (int32_t)q;
            orc_apply(v->mode, v->packed, v->mem[0], v->mem[1], page, y, ia, content);
            if (q) {
                uint32_t nonce = draw_py(v);
                v->pushed[v->n_pushed++] = ((2047u - q) << 21) | (nonce << 13) | ((uint32_t)page << 8) | (uint32_t)y;
            }
            offs[1 + i] = y;
        }
        out[0] = (uint8_t)(page + 32);
        out[1] = content;
        out[2] = (uint8_t)offs[0];
        out[3] = (uint8_t)offs[1];
        out[4] = (uint8_t)offs[2];
        out[5] = (uint8_t)offs[3];
        return 0;
    }
    v->out_of_work[ia] = 1;
    v->gen_exhausted = 1;
    emit_pad(v, out);
    return 0;
}

static int video_next(orc_video *v, int k, uint8_t *ops_out, int form)
{
    if (!v->gen_active)
        return -10;
    for (int i = 0; i < k; i++) {
        uint8_t *out = ops_out + 6 * i;
        if (!v->gen_started) {
            int rc = gen_prologue(v, form);
            if (rc)
                return rc;
            v->gen_started = 1;
        }
        if (v->gen_form != form)
            return -11;
        if (v->gen_exhausted) {
            emit_pad(v, out);
            continue;
        }
        int rc = form == FORM_HEAP ? step_heap(v, out) : step_struct(v, out);
        if (rc)
            return rc;
    }
    return 0;
}

int orc_video_next(orc_video *v, int k, uint8_t *ops_out) { return video_next(v, k, ops_out, FORM_HEAP); }

int orc_video_next_structured(orc_video *v, int k, uint8_t *ops_out)
{
    return video_next(v, k, ops_out, FORM_STRUCT);
}

/* ------------------------------------------------------------------------- */
/* byte emission (f2): movie.Movie.emit_stream / done (movie.py:113-161),      */
/* opcodes.Header / BaseTick / Ack / Terminate (opcodes.py:64-139),            */
/* machine.Machine.emit (machine.py:11-25)                                     */
/* ------------------------------------------------------------------------- */

/* ops: n x 6 (page+32, content, 4 offsets); ticks: n values 4..66 (even);
 * tick_addr[(tick-4)/2 * 32 + page-32], ack / terminate = opcode start addresses
 * (from the player's symbol table).  Returns the number of bytes written. */
size_t orc_emit_stream(int mode, int n_ops, const uint8_t *ops, const uint8_t *ticks,
                       const uint16_t tick_addr[1024], uint16_t ack_addr, uint16_t terminate_addr,
                       long max_bytes_out, uint8_t *out)
{
    size_t pos = 0;
    int aux = 0;
    /* Header: no command bytes, 6 x 0xff + mode (opcodes.py:64-90) */
    int stop = (max_bytes_out > 0 && (long)pos >= max_bytes_out);
    if (!stop) {
        for (int i = 0; i < 6; i++) out[pos++] = 0xff;
        out[pos++] = (uint8_t)mode;
        for (int k = 0; k < n_ops; k++) {
            if (max_bytes_out > 0 && (long)pos >= max_bytes_out) break; /* movie.py:132-134 */
            const uint8_t *o = ops + 6 * k;
            uint16_t a = tick_addr[((ticks[k] - 4) / 2) * 32 + (o[0] - 32)];
            out[pos++] = (uint8_t)(a >> 8);  /* emit_command, opcodes.py:49-53 */
            out[pos++] = (uint8_t)(a & 0xff);
            for (int i = 1; i < 6; i++) out[pos++] = o[i]; /* content + 4 offsets */
            if (pos % 2048 >= 2044) {      /* movie.py:139-148 */
                if (mode == ORC_DHGR) aux = !aux;
                out[pos++] = (uint8_t)(ack_addr >> 8);
                out[pos++] = (uint8_t)(ack_addr & 0xff);
                out[pos++] = aux ? 0x55 : 0x54;
                out[pos++] = 0xff;
            }
        }
    }
    /* done(): Terminate + zero padding to the 2 KiB boundary (movie.py:152-161) */
    out[pos++] = (uint8_t)(terminate_addr >> 8);
    out[pos++] = (uint8_t)(terminate_addr & 0xff);
    size_t pad = 2048 - (pos % 2048);
    for (size_t i = 0; i < pad; i++) out[pos++] = 0;
    return pos;
}


/* ---- frame ingest (f3): see iiv_oracle.h -- this is the definition, not a restatement ---- */

static const int kBayer4[4][4] = {{0, 8, 2, 10}, {12, 4, 14, 6}, {3, 11, 1, 9}, {15, 7, 13, 5}};

/* colour pixel k of row y: mean of source pixels 2k, 2k+1, plus the ordered-dither offset */
static void ingest_pixel(const uint8_t *rgb, int y, int k, int dither, int out[3])
{
    const uint8_t *p = rgb + ((size_t)y * 280 + 2 * k) * 3;
    const int d = ((2 * kBayer4[y & 3][k & 3] - 15) * dither + 16 * 256) / 16 - 256;  /* floor((2b-15) * dither / 16) */
    for (int c = 0; c < 3; c++) {
        int v = ((int)p[c] + (int)p[3 + c] + 1) / 2 + d;
        out[c] = v < 0 ? 0 : v > 255 ? 255 : v;
    }
}

static int ingest_err(const uint8_t *pal, int colour, const int px[3])
{
    const int dr = px[0] - pal[3 * colour], dg = px[1] - pal[3 * colour + 1], db = px[2] - pal[3 * colour + 2];
    return 2 * dr * dr + 4 * dg * dg + 3 * db * db;
}

/* HGR: the four colours a pixel can take under a palette bit, as (colour value, 2-dot pattern):
 * pattern bit 0 = the even dot column, bit 1 = the odd one (colours.py:18-44) */
static const int kHgrColour[2][4] = {{0, 3, 12, 15}, {0, 6, 9, 15}};  /* black, violet|blue, green|orange, white */
static const int kHgrPattern[4] = {0, 1, 2, 3};

/* dither == ORC_DITHER_DIFFUSION (256): Floyd-Steinberg error diffusion over the 140 x 192 colour pixels, rows top
 * to bottom, left to right, all integer (include/iivision.h): value = clamp(mean of the two source pixels +
 * floor(acc / 16)); e = value - chosen colour; acc[right] += 7e, acc[below left] += 3e, acc[below] += 5e,
 * acc[below right] += e.  HGR: the palette bit of screen byte b is fixed just before the first pixel whose first
 * dot lies in b is quantised, by comparing, for both palette bits, the summed nearest-colour error of the pixels
 * whose first dot lies in b (each weighted by how many of its dots lie in b), values taken with the error
 * accumulated so far; ties to palette bit 0.  A pixel is quantised under the palette bit of the byte holding its
 * first dot; its pattern's bit 0 / 1 go to its first / second dot. */
static void frame_to_memory_map_diffusion(int mode, const uint8_t *pal, const uint8_t *rgb, uint8_t *main_mem, uint8_t *aux_mem)
{
    int acc[2][142][3];   /* [row parity][pixel + 1][channel], sixteenths */
    memset(acc, 0, sizeof acc);
    for (int y = 0; y < 192; y++) {
        const int base = orc_y_to_base_addr(y, 0) - 0x2000;
        int (*cur)[3] = acc[y & 1], (*nxt)[3] = acc[(y + 1) & 1];
        memset(nxt, 0, sizeof acc[0]);
        int pattern[140];   /* DHGR: colour value = dot quad; HGR: 2-dot pattern */
        int pb = 0;
        for (int k = 0; k < 140; k++) {
            int v[3];
#define ORC_VALUE(K, OUT)                                                                        \
    do {                                                                                         \
        const uint8_t *p_ = rgb + ((size_t)y * 280 + 2 * (K)) * 3;                               \
        for (int c_ = 0; c_ < 3; c_++) {                                                         \
            const int a_ = cur[(K) + 1][c_];                                                     \
            int t_ = ((int)p_[c_] + (int)p_[3 + c_] + 1) / 2 + (a_ >= 0 ? a_ / 16 : -((-a_ + 15) / 16)); \
            (OUT)[c_] = t_ < 0 ? 0 : t_ > 255 ? 255 : t_;                                        \
        }                                                                                        \
    } while (0)
            int colour;
            if (mode == ORC_DHGR) {
                ORC_VALUE(k, v);
                int best = 0, be = 0x7fffffff;
                for (int c = 0; c < 16; c++) {
                    const int e = ingest_err(pal, c, v);
                    if (e < be) { be = e; best = c; }
                }
                pattern[k] = colour = best;
            } else {
                if ((2 * k) / 7 != (2 * k - 2) / 7 || k == 0) {   /* the first dot of this pixel opens screen byte b */
                    const int b = (2 * k) / 7;
                    long err[2] = {0, 0};
                    for (int kk = k; kk < 140 && (2 * kk) / 7 == b; kk++) {
                        int u[3];
                        ORC_VALUE(kk, u);
                        const int w = (2 * kk + 1) / 7 == b ? 2 : 1;
                        for (int q = 0; q < 2; q++) {
                            int be = 0x7fffffff;
                            for (int i = 0; i < 4; i++) {
                                const int e = ingest_err(pal, kHgrColour[q][i], u);
                                if (e < be) be = e;
                            }
                            err[q] += (long)w * be;
                        }
                    }
                    pb = err[1] < err[0] ? 1 : 0;
                    main_mem[base + b] = (uint8_t)(pb << 7);
                }
                ORC_VALUE(k, v);
                int best = 0, be = 0x7fffffff;
                for (int i = 0; i < 4; i++) {
                    const int e = ingest_err(pal, kHgrColour[pb][i], v);
                    if (e < be) { be = e; best = i; }
                }
                pattern[k] = best;
                colour = kHgrColour[pb][best];
            }
#undef ORC_VALUE
            for (int c = 0; c < 3; c++) {
                const int e = v[c] - pal[3 * colour + c];
                cur[k + 2][c] += 7 * e;
                nxt[k][c] += 3 * e;
                nxt[k + 1][c] += 5 * e;
                nxt[k + 2][c] += e;
            }
        }
        if (mode == ORC_DHGR) {
            for (int j = 0; j < 80; j++) {
                int val = 0;
                for (int i = 0; i < 7; i++) {
                    const int X = 7 * j + i;
                    val |= ((pattern[X >> 2] >> (X & 3)) & 1) << i;
                }
                ((j & 1) ? main_mem : aux_mem)[base + (j >> 1)] = (uint8_t)val;
            }
        } else {
            for (int b = 0; b < 40; b++) {
                int val = main_mem[base + b];
                for (int i = 0; i < 7; i++) {
                    const int X = 7 * b + i;
                    val |= ((pattern[X >> 1] >> (X & 1)) & 1) << i;
                }
                main_mem[base + b] = (uint8_t)val;
            }
        }
    }
}

void orc_frame_to_memory_map(int mode, const uint8_t palette_rgb[48], const uint8_t *rgb, int dither,
                             uint8_t *main_mem, uint8_t *aux_mem)
{
    memset(main_mem, 0, 8192);
    if (mode == ORC_DHGR) memset(aux_mem, 0, 8192);
    if (dither == ORC_DITHER_DIFFUSION) {
        frame_to_memory_map_diffusion(mode, palette_rgb, rgb, main_mem, aux_mem);
        return;
    }
    for (int y = 0; y < 192; y++) {
        const int base = orc_y_to_base_addr(y, 0) - 0x2000;
        if (mode == ORC_DHGR) {
            int quad[140];
            for (int k = 0; k < 140; k++) {
                int px[3], best = 0, be = 0x7fffffff;
                ingest_pixel(rgb, y, k, dither, px);
                for (int c = 0; c < 16; c++) {
                    const int e = ingest_err(palette_rgb, c, px);
                    if (e < be) { be = e; best = c; }
                }
                quad[k] = best;
            }
            for (int j = 0; j < 80; j++) {      /* bytes in dot order: aux, main, aux, main ... (screen.py:822-826) */
                int v = 0;
                for (int i = 0; i < 7; i++) {
                    const int X = 7 * j + i;
                    v |= ((quad[X >> 2] >> (X & 3)) & 1) << i;
                }
                ((j & 1) ? main_mem : aux_mem)[base + (j >> 1)] = (uint8_t)v;
            }
        } else {
            int pat[2][140], err[2][140];
            for (int k = 0; k < 140; k++) {
                int px[3];
                ingest_pixel(rgb, y, k, dither, px);
                for (int pb = 0; pb < 2; pb++) {
                    int best = 0, be = 0x7fffffff;
                    for (int q = 0; q < 4; q++) {
                        const int e = ingest_err(palette_rgb, kHgrColour[pb][q], px);
                        if (e < be) { be = e; best = q; }
                    }
                    pat[pb][k] = kHgrPattern[best];
                    err[pb][k] = be;
                }
            }
            for (int b = 0; b < 40; b++) {
                long e0 = 0, e1 = 0;
                for (int i = 0; i < 7; i++) {
                    e0 += err[0][(7 * b + i) >> 1];
                    e1 += err[1][(7 * b + i) >> 1];
                }
                const int pb = e1 < e0 ? 1 : 0;
                int v = pb << 7;
                for (int i = 0; i < 7; i++) {
                    const int X = 7 * b + i;
                    v |= ((pat[pb][X >> 1] >> (X & 1)) & 1) << i;
                }
                main_mem[base + b] = (uint8_t)v;
            }
        }
    }
}
