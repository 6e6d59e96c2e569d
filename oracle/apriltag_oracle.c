/*
 * oracle/apriltag_oracle.c -- TEST INFRASTRUCTURE ONLY (see apriltag_oracle.h).
 *
 * Plain scalar C restatement of the detector + PnP algorithms that the reference
 * reaches via tag_detector.py:25 (cv2.cvtColor), :26 (apriltag detect), :41
 * (cv2.solvePnP) and :47 (cv2.Rodrigues).  Upstream sources are not available in
 * this container, so every stage follows the published algorithm; constants that
 * upstream leaves tunable are fixed here and listed in DESIGN.md ("build-defined").
 * Where upstream's result depends on container iteration order (hash-map order of
 * clusters, unstable sort ties) a deterministic order is defined instead:
 *   - component representative = smallest raster index of the component
 *   - clusters are visited in ascending cluster id
 *   - boundary points are sorted by (slope, y, x)
 * Compile with -ffp-contract=off: the HIP kernels are built the same way so that
 * float/double results can be compared bit for bit.
 */
#include "apriltag_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------ S0 gray */
/* OpenCV >= 4 BGR2GRAY for 8-bit: 15-bit fixed point, round half up. */
void aso_bgr2gray(const uint8_t *bgr, int w, int h, int stride, uint8_t *gray)
{
    for (int y = 0; y < h; y++) {
        const uint8_t *row = bgr + (size_t)y * stride;
        for (int x = 0; x < w; x++) {
            int b = row[3 * x], g = row[3 * x + 1], r = row[3 * x + 2];
            gray[(size_t)y * w + x] = (uint8_t)((b * 3735 + g * 19235 + r * 9798 + 16384) >> 15);
        }
    }
}

/* -------------------------------------------------------------- S1 decimate */
void aso_decimate(const uint8_t *gray, int w, int h, int stride, int f, uint8_t *out, int *sw, int *sh)
{
    int swidth = 1 + (w - 1) / f, sheight = 1 + (h - 1) / f;
    int sy = 0;
    for (int y = 0; y < h; y += f, sy++) {
        int sx = 0;
        for (int x = 0; x < w; x += f, sx++)
            out[(size_t)sy * swidth + sx] = gray[(size_t)y * stride + x];
    }
    *sw = swidth;
    *sh = sheight;
}

/* ------------------------------------------------------------- S2 threshold */
#define TILESZ 4
#define MIN_WHITE_BLACK_DIFF 5

void aso_threshold(const uint8_t *im, int w, int h, uint8_t *out)
{
    int tw = w / TILESZ, th = h / TILESZ;
    if (tw == 0 || th == 0) {
        memset(out, 127, (size_t)w * h);
        return;
    }
    uint8_t *tmax = malloc((size_t)tw * th), *tmin = malloc((size_t)tw * th);
    uint8_t *dmax = malloc((size_t)tw * th), *dmin = malloc((size_t)tw * th);
    for (int ty = 0; ty < th; ty++)
        for (int tx = 0; tx < tw; tx++) {
            uint8_t mx = 0, mn = 255;
            for (int dy = 0; dy < TILESZ; dy++)
                for (int dx = 0; dx < TILESZ; dx++) {
                    uint8_t v = im[(size_t)(ty * TILESZ + dy) * w + tx * TILESZ + dx];
                    if (v < mn) mn = v;
                    if (v > mx) mx = v;
                }
            tmax[ty * tw + tx] = mx;
            tmin[ty * tw + tx] = mn;
        }
    /* 3x3 tile neighbourhood: max of max, min of min */
    for (int ty = 0; ty < th; ty++)
        for (int tx = 0; tx < tw; tx++) {
            uint8_t mx = 0, mn = 255;
            for (int dy = -1; dy <= 1; dy++) {
                if (ty + dy < 0 || ty + dy >= th) continue;
                for (int dx = -1; dx <= 1; dx++) {
                    if (tx + dx < 0 || tx + dx >= tw) continue;
                    uint8_t a = tmax[(ty + dy) * tw + tx + dx], b = tmin[(ty + dy) * tw + tx + dx];
                    if (a > mx) mx = a;
                    if (b < mn) mn = b;
                }
            }
            dmax[ty * tw + tx] = mx;
            dmin[ty * tw + tx] = mn;
        }
    /* every pixel uses the (clamped) tile it falls in: identical to upstream's
       "full tiles, then right/bottom leftovers with the nearest tile" */
    for (int y = 0; y < h; y++) {
        int ty = y / TILESZ;
        if (ty >= th) ty = th - 1;
        for (int x = 0; x < w; x++) {
            int tx = x / TILESZ;
            if (tx >= tw) tx = tw - 1;
            int mn = dmin[ty * tw + tx], mx = dmax[ty * tw + tx];
            uint8_t o;
            if (mx - mn < MIN_WHITE_BLACK_DIFF)
                o = 127;
            else {
                int thresh = mn + (mx - mn) / 2;
                o = im[(size_t)y * w + x] > thresh ? 255 : 0;
            }
            out[(size_t)y * w + x] = o;
        }
    }
    free(tmax); free(tmin); free(dmax); free(dmin);
}

/* ------------------------------------------------- S3 connected components */
static uint32_t uf_find(uint32_t *parent, uint32_t a)
{
    uint32_t r = a;
    while (parent[r] != r) r = parent[r];
    while (parent[a] != r) { uint32_t n = parent[a]; parent[a] = r; a = n; }
    return r;
}
/* union keeps the smaller index as root, so the final root IS the canonical label */
static void uf_union(uint32_t *parent, uint32_t a, uint32_t b)
{
    a = uf_find(parent, a);
    b = uf_find(parent, b);
    if (a == b) return;
    if (a < b) parent[b] = a; else parent[a] = b;
}

/* Pixels with 1 <= x <= w-2 initiate unions: left always; up for y > 0; white pixels
   additionally up-left and up-right (white 8-connected, black 4-connected, 127 never
   joined).  Columns 0 and w-1 only ever receive unions -- upstream's row loops run
   x = 1 .. w-2 -- and that quirk is kept. */
void aso_connected_components(const uint8_t *th, int w, int h, uint32_t *labels, uint32_t *sizes)
{
    size_t n = (size_t)w * h;
    for (size_t i = 0; i < n; i++) labels[i] = (uint32_t)i;
    for (int y = 0; y < h; y++)
        for (int x = 1; x < w - 1; x++) {
            uint8_t v = th[(size_t)y * w + x];
            if (v == 127) continue;
            uint32_t p = (uint32_t)(y * w + x);
            if (th[p - 1] == v) uf_union(labels, p, p - 1);
            if (y > 0) {
                if (th[p - w] == v) uf_union(labels, p, p - w);
                if (v == 255) {
                    if (th[p - w - 1] == v) uf_union(labels, p, p - w - 1);
                    if (th[p - w + 1] == v) uf_union(labels, p, p - w + 1);
                }
            }
        }
    memset(sizes, 0, n * sizeof(uint32_t));
    for (size_t i = 0; i < n; i++) {
        labels[i] = uf_find(labels, (uint32_t)i);
        sizes[labels[i]]++;
    }
}

/* --------------------------------------------------- S4 gradient clusters */
#define MIN_COMPONENT 25 /* both components of a boundary must have >= 25 pixels */

static int pt_cmp_full(const void *pa, const void *pb)
{
    const aso_point *a = pa, *b = pb;
    if (a->cluster != b->cluster) return a->cluster < b->cluster ? -1 : 1;
    if (a->y != b->y) return a->y < b->y ? -1 : 1;
    if (a->x != b->x) return a->x < b->x ? -1 : 1;
    if (a->gx != b->gx) return a->gx < b->gx ? -1 : 1;
    if (a->gy != b->gy) return a->gy < b->gy ? -1 : 1;
    return 0;
}

static int conn(const uint8_t *th, int w, const uint32_t *labels, const uint32_t *sizes,
                int x, int y, int dx, int dy, uint8_t v0, uint32_t rep0, aso_point *out)
{
    uint8_t v1 = th[(size_t)(y + dy) * w + x + dx];
    if (v0 + v1 != 255) return 0;
    uint32_t rep1 = labels[(size_t)(y + dy) * w + x + dx];
    if (sizes[rep1] < MIN_COMPONENT) return 0;
    if (out) {
        uint64_t a = rep0, b = rep1;
        out->cluster = a < b ? (b << 32) + a : (a << 32) + b;
        out->x = (uint16_t)(2 * x + dx);
        out->y = (uint16_t)(2 * y + dy);
        out->gx = (int16_t)(dx * ((int)v1 - (int)v0));
        out->gy = (int16_t)(dy * ((int)v1 - (int)v0));
    }
    return 1;
}

long aso_gradient_clusters(const uint8_t *th, int w, int h, const uint32_t *labels, const uint32_t *sizes,
                           aso_point *out, long cap)
{
    long n = 0;
    aso_point tmp;
    for (int y = 1; y < h - 1; y++) {
        int connected_last = 0;
        for (int x = 1; x < w - 1; x++) {
            uint8_t v0 = th[(size_t)y * w + x];
            if (v0 == 127) { connected_last = 0; continue; }
            uint32_t rep0 = labels[(size_t)y * w + x];
            if (sizes[rep0] < MIN_COMPONENT) { connected_last = 0; continue; }
#define DO_CONN(dx, dy) (conn(th, w, labels, sizes, x, y, dx, dy, v0, rep0, n < cap ? &out[n] : &tmp) ? (n++, 1) : 0)
            DO_CONN(1, 0);
            DO_CONN(0, 1);
            /* (x-1,y)+(1,1) and (x,y)+(-1,1) name the same half-pixel point */
            if (!connected_last) DO_CONN(-1, 1);
            connected_last = DO_CONN(1, 1);
#undef DO_CONN
        }
    }
    if (n > cap) return -n;
    qsort(out, (size_t)n, sizeof(aso_point), pt_cmp_full);
    return n;
}

/* ------------------------------------------------------------ S5 quad fit */
#define MAX_NMAXIMA 10
#define MAX_LINE_FIT_MSE 10.0
#define COS_CRITICAL_RAD 0.984807753012208 /* cos(10 deg) */

typedef struct { double Mx, My, Mxx, Mxy, Myy, W; } lfp_t;

typedef struct { float slope; uint16_t x, y; } spt_t;

static int spt_cmp(const void *pa, const void *pb)
{
    const spt_t *a = pa, *b = pb;
    if (a->slope != b->slope) return a->slope < b->slope ? -1 : 1;
    if (a->y != b->y) return a->y < b->y ? -1 : 1;
    if (a->x != b->x) return a->x < b->x ? -1 : 1;
    return 0;
}

static void fit_line(const lfp_t *lfps, int sz, int i0, int i1, double *lineparm, double *err, double *mse)
{
    double Mx, My, Mxx, Myy, Mxy, W;
    int N;
    if (i0 < i1) {
        N = i1 - i0 + 1;
        Mx = lfps[i1].Mx; My = lfps[i1].My; Mxx = lfps[i1].Mxx; Mxy = lfps[i1].Mxy; Myy = lfps[i1].Myy; W = lfps[i1].W;
        if (i0 > 0) {
            Mx -= lfps[i0 - 1].Mx; My -= lfps[i0 - 1].My; Mxx -= lfps[i0 - 1].Mxx;
            Mxy -= lfps[i0 - 1].Mxy; Myy -= lfps[i0 - 1].Myy; W -= lfps[i0 - 1].W;
        }
    } else {
        Mx = lfps[sz - 1].Mx - lfps[i0 - 1].Mx; My = lfps[sz - 1].My - lfps[i0 - 1].My;
        Mxx = lfps[sz - 1].Mxx - lfps[i0 - 1].Mxx; Mxy = lfps[sz - 1].Mxy - lfps[i0 - 1].Mxy;
        Myy = lfps[sz - 1].Myy - lfps[i0 - 1].Myy; W = lfps[sz - 1].W - lfps[i0 - 1].W;
        Mx += lfps[i1].Mx; My += lfps[i1].My; Mxx += lfps[i1].Mxx; Mxy += lfps[i1].Mxy; Myy += lfps[i1].Myy; W += lfps[i1].W;
        N = sz - i0 + i1 + 1;
    }
    double Ex = Mx / W, Ey = My / W;
    double Cxx = Mxx / W - Ex * Ex, Cxy = Mxy / W - Ex * Ey, Cyy = Myy / W - Ey * Ey;
    double disc = (double)sqrtf((float)((Cxx - Cyy) * (Cxx - Cyy) + 4 * Cxy * Cxy));
    double eig_small = 0.5 * (Cxx + Cyy - disc);
    if (lineparm) {
        lineparm[0] = Ex;
        lineparm[1] = Ey;
        double eig = 0.5 * (Cxx + Cyy + disc);
        double nx1 = Cxx - eig, ny1 = Cxy, M1 = nx1 * nx1 + ny1 * ny1;
        double nx2 = Cxy, ny2 = Cyy - eig, M2 = nx2 * nx2 + ny2 * ny2;
        double nx, ny, M;
        if (M1 > M2) { nx = nx1; ny = ny1; M = M1; } else { nx = nx2; ny = ny2; M = M2; }
        double length = (double)sqrtf((float)M);
        if (fabs(length) < 1e-12) { lineparm[2] = lineparm[3] = 0; }
        else { lineparm[2] = nx / length; lineparm[3] = ny / length; }
    }
    if (err) *err = N * eig_small;
    if (mse) *mse = eig_small;
}

/* Gaussian low-pass taps exp(-j*j/2), j=-3..3, rounded to float (sigma=1, cutoff=0.05 -> 7 taps) */
static const float LPF[7] = {0.011108996538242306f, 0.1353352832366127f, 0.6065306597126334f, 1.0f,
                             0.6065306597126334f, 0.1353352832366127f, 0.011108996538242306f};

static int quad_segment_maxima(const lfp_t *lfps, int sz, int indices[4])
{
    int ksz = sz / 12 < 20 ? sz / 12 : 20;
    if (ksz < 2) return 0;
    double *errs = malloc(sizeof(double) * sz), *sm = malloc(sizeof(double) * sz);
    for (int i = 0; i < sz; i++)
        fit_line(lfps, sz, (i + sz - ksz) % sz, (i + ksz) % sz, NULL, &errs[i], NULL);
    for (int iy = 0; iy < sz; iy++) {
        double acc = 0;
        for (int i = 0; i < 7; i++) acc += errs[(iy + i - 3 + sz) % sz] * LPF[i];
        sm[iy] = acc;
    }
    int *maxima = malloc(sizeof(int) * sz);
    double *maxima_errs = malloc(sizeof(double) * sz);
    int nmaxima = 0;
    for (int i = 0; i < sz; i++)
        if (sm[i] > sm[(i + 1) % sz] && sm[i] > sm[(i + sz - 1) % sz]) {
            maxima[nmaxima] = i;
            maxima_errs[nmaxima] = sm[i];
            nmaxima++;
        }
    free(errs); free(sm);
    int ok = 0;
    if (nmaxima < 4) goto done;
    if (nmaxima > MAX_NMAXIMA) {
        /* keep maxima strictly above the (MAX_NMAXIMA+1)-th largest error */
        double *cp = malloc(sizeof(double) * nmaxima);
        memcpy(cp, maxima_errs, sizeof(double) * nmaxima);
        for (int i = 0; i <= MAX_NMAXIMA; i++) { /* partial selection sort, descending */
            int best = i;
            for (int j = i + 1; j < nmaxima; j++) if (cp[j] > cp[best]) best = j;
            double t = cp[i]; cp[i] = cp[best]; cp[best] = t;
        }
        double thresh = cp[MAX_NMAXIMA];
        free(cp);
        int out = 0;
        for (int in = 0; in < nmaxima; in++) {
            if (maxima_errs[in] <= thresh) continue;
            maxima[out++] = maxima[in];
        }
        nmaxima = out;
    }
    {
        int best_indices[4] = {0, 0, 0, 0};
        double best_error = HUGE_VALF;
        double err01, err12, err23, err30, mse01, mse12, mse23, mse30;
        double params01[4], params12[4];
        for (int m0 = 0; m0 < nmaxima - 3; m0++) {
            int i0 = maxima[m0];
            for (int m1 = m0 + 1; m1 < nmaxima - 2; m1++) {
                int i1 = maxima[m1];
                fit_line(lfps, sz, i0, i1, params01, &err01, &mse01);
                if (mse01 > MAX_LINE_FIT_MSE) continue;
                for (int m2 = m1 + 1; m2 < nmaxima - 1; m2++) {
                    int i2 = maxima[m2];
                    fit_line(lfps, sz, i1, i2, params12, &err12, &mse12);
                    if (mse12 > MAX_LINE_FIT_MSE) continue;
                    double dot = params01[2] * params12[2] + params01[3] * params12[3];
                    if (fabs(dot) > COS_CRITICAL_RAD) continue;
                    for (int m3 = m2 + 1; m3 < nmaxima; m3++) {
                        int i3 = maxima[m3];
                        fit_line(lfps, sz, i2, i3, NULL, &err23, &mse23);
                        if (mse23 > MAX_LINE_FIT_MSE) continue;
                        fit_line(lfps, sz, i3, i0, NULL, &err30, &mse30);
                        if (mse30 > MAX_LINE_FIT_MSE) continue;
                        double err = err01 + err12 + err23 + err30;
                        if (err < best_error) {
                            best_error = err;
                            best_indices[0] = i0; best_indices[1] = i1; best_indices[2] = i2; best_indices[3] = i3;
                        }
                    }
                }
            }
        }
        if (best_error != HUGE_VALF) {
            for (int i = 0; i < 4; i++) indices[i] = best_indices[i];
            if (best_error / sz < MAX_LINE_FIT_MSE) ok = 1;
        }
    }
done:
    free(maxima); free(maxima_errs);
    return ok;
}

static double sq(double x) { return x * x; }

/* one cluster -> quad (corners in decimated-image pixel coordinates) */
static int fit_quad(const uint8_t *im, int w, int h, const aso_point *pts, int sz, int tag_width,
                    int normal_border, int reversed_border, aso_quad *quad)
{
    if (sz < 24) return 0;
    int xmax = pts[0].x, xmin = pts[0].x, ymax = pts[0].y, ymin = pts[0].y;
    for (int i = 1; i < sz; i++) {
        if (pts[i].x > xmax) xmax = pts[i].x; else if (pts[i].x < xmin) xmin = pts[i].x;
        if (pts[i].y > ymax) ymax = pts[i].y; else if (pts[i].y < ymin) ymin = pts[i].y;
    }
    if ((xmax - xmin) * (ymax - ymin) < tag_width) return 0;

    float cx = (float)((xmin + xmax) * 0.5 + 0.05118);
    float cy = (float)((ymin + ymax) * 0.5 + -0.028581);
    float dot = 0;
    static const float quadrants[2][2] = {{-1 * (2 << 15), 0}, {2 * (2 << 15), 2 << 15}};
    spt_t *sp = malloc(sizeof(spt_t) * sz);
    for (int i = 0; i < sz; i++) {
        float dx = pts[i].x - cx, dy = pts[i].y - cy;
        dot += dx * pts[i].gx + dy * pts[i].gy;
        float quadrant = quadrants[dy > 0][dx > 0];
        if (dy < 0) { dy = -dy; dx = -dx; }
        if (dx < 0) { float t = dx; dx = dy; dy = -t; }
        sp[i].slope = quadrant + dy / dx;
        sp[i].x = pts[i].x;
        sp[i].y = pts[i].y;
    }
    quad->reversed_border = dot < 0;
    if ((!reversed_border && quad->reversed_border) || (!normal_border && !quad->reversed_border)) { free(sp); return 0; }

    qsort(sp, (size_t)sz, sizeof(spt_t), spt_cmp);
    { /* drop duplicate points */
        int outpos = 1;
        for (int i = 1; i < sz; i++)
            if (sp[i].x != sp[i - 1].x || sp[i].y != sp[i - 1].y) sp[outpos++] = sp[i];
        /* sp[i-1] is still the previous INPUT element here: writes only reach indices < i-1 or i itself */
        sz = outpos;
    }
    if (sz < 24) { free(sp); return 0; }

    lfp_t *lfps = calloc((size_t)sz, sizeof(lfp_t));
    for (int i = 0; i < sz; i++) {
        if (i > 0) lfps[i] = lfps[i - 1];
        double x = sp[i].x * .5 + 0.5, y = sp[i].y * .5 + 0.5;
        int ix = (int)x, iy = (int)y;
        double W = 1;
        if (ix > 0 && ix + 1 < w && iy > 0 && iy + 1 < h) {
            int grad_x = im[iy * w + ix + 1] - im[iy * w + ix - 1];
            int grad_y = im[(iy + 1) * w + ix] - im[(iy - 1) * w + ix];
            W = sqrt((double)(grad_x * grad_x + grad_y * grad_y)) + 1;
        }
        lfps[i].Mx += W * x; lfps[i].My += W * y;
        lfps[i].Mxx += W * x * x; lfps[i].Mxy += W * x * y; lfps[i].Myy += W * y * y;
        lfps[i].W += W;
    }
    free(sp);

    int res = 0, indices[4];
    double lines[4][4];
    if (!quad_segment_maxima(lfps, sz, indices)) goto finish;
    for (int i = 0; i < 4; i++) {
        /* The points next to a corner are the least reliable ones of a side (a missing corner pixel or
           unequal gradient weights on the two sides move the error maximum by a point or two), so a sixth
           of the side is left out at either end -- the block upstream carries for this.  The reference's
           committed run only reproduces with it on: see tests/golden/README.md. */
        int i0 = indices[i], i1 = indices[(i + 1) & 3];
        int len = i1 - i0;
        if (len < 0) len += sz;
        if (len > 8) {
            int t = len / 6;
            i0 = (i0 + t) % sz;
            i1 = (i1 + sz - t) % sz;
        }
        double mse;
        fit_line(lfps, sz, i0, i1, lines[i], NULL, &mse);
        if (mse > MAX_LINE_FIT_MSE) goto finish;
    }
    for (int i = 0; i < 4; i++) {
        double A00 = lines[i][3], A01 = -lines[(i + 1) & 3][3];
        double A10 = -lines[i][2], A11 = lines[(i + 1) & 3][2];
        double B0 = -lines[i][0] + lines[(i + 1) & 3][0];
        double B1 = -lines[i][1] + lines[(i + 1) & 3][1];
        double det = A00 * A11 - A10 * A01;
        if (fabs(det) < 0.001) goto finish;
        double W00 = A11 / det, W01 = -A01 / det;
        double L0 = W00 * B0 + W01 * B1;
        quad->p[i][0] = lines[i][0] + L0 * A00;
        quad->p[i][1] = lines[i][1] + L0 * A10;
    }
    { /* area */
        double area = 0, length[3], p;
        for (int i = 0; i < 3; i++) {
            int a = i, b = (i + 1) % 3;
            length[i] = sqrt(sq(quad->p[b][0] - quad->p[a][0]) + sq(quad->p[b][1] - quad->p[a][1]));
        }
        p = (length[0] + length[1] + length[2]) / 2;
        area += sqrt(p * (p - length[0]) * (p - length[1]) * (p - length[2]));
        static const int idxs[4] = {2, 3, 0, 2};
        for (int i = 0; i < 3; i++) {
            int a = idxs[i], b = idxs[i + 1];
            length[i] = sqrt(sq(quad->p[b][0] - quad->p[a][0]) + sq(quad->p[b][1] - quad->p[a][1]));
        }
        p = (length[0] + length[1] + length[2]) / 2;
        area += sqrt(p * (p - length[0]) * (p - length[1]) * (p - length[2]));
        if (area < 0.95 * tag_width * tag_width) goto finish;
    }
    for (int i = 0; i < 4; i++) { /* convexity, winding and minimum corner angle */
        int i0 = i, i1 = (i + 1) & 3, i2 = (i + 2) & 3;
        double dx1 = quad->p[i1][0] - quad->p[i0][0], dy1 = quad->p[i1][1] - quad->p[i0][1];
        double dx2 = quad->p[i2][0] - quad->p[i1][0], dy2 = quad->p[i2][1] - quad->p[i1][1];
        double cos_dtheta = (dx1 * dx2 + dy1 * dy2) / sqrt((dx1 * dx1 + dy1 * dy1) * (dx2 * dx2 + dy2 * dy2));
        if ((cos_dtheta > COS_CRITICAL_RAD || cos_dtheta < -COS_CRITICAL_RAD) || dx1 * dy2 < dy1 * dx2) goto finish;
    }
    res = 1;
finish:
    free(lfps);
    return res;
}

int aso_fit_quads(const uint8_t *dec, int w, int h, const aso_point *pts, long npts,
                  const aso_family *fam, int decimate, aso_quad *out, int cap)
{
    int min_tag_width = fam->width_at_border / decimate;
    if (min_tag_width < 3) min_tag_width = 3;
    int normal_border = !fam->reversed_border, reversed_border = fam->reversed_border;
    int nq = 0;
    long i = 0;
    while (i < npts) {
        long j = i;
        while (j < npts && pts[j].cluster == pts[i].cluster) j++;
        long sz = j - i;
        if (sz >= 24 && sz <= 3L * (2 * w + 2 * h) && nq < cap) {
            aso_quad q;
            memset(&q, 0, sizeof q);
            if (fit_quad(dec, w, h, pts + i, (int)sz, min_tag_width, normal_border, reversed_border, &q)) {
                q.cluster = pts[i].cluster;
                out[nq++] = q;
            }
        }
        i = j;
    }
    return nq;
}

/* ------------------------------------------------------- S6 edge refinement */
/* unit normal (cos t, sin t), t = 0.5*atan2(-2Cxy, Cyy-Cxx), written with square
   roots only so that CPU and GPU agree bit for bit (no libm trig). */
static void half_angle_normal(double a /*Cyy-Cxx*/, double b /*-2Cxy*/, double *nx, double *ny)
{
    double r = sqrt(a * a + b * b);
    if (r == 0) { *nx = 1; *ny = 0; return; }
    double c2 = a / r;
    if (c2 >= 0) {
        double c = sqrt((1 + c2) / 2);
        *nx = c;
        *ny = b / (2 * r * c);
    } else {
        double s = sqrt((1 - c2) / 2);
        if (b < 0) s = -s;
        *ny = s;
        *nx = b / (2 * r * s);
    }
}

#define PIX_EPS 9.5367431640625e-07 /* 2^-20 px */

void aso_refine_edges(const uint8_t *gray, int w, int h, int stride, int decimate, aso_quad *quad)
{
    double lines[4][4];
    for (int edge = 0; edge < 4; edge++) {
        int a = edge, b = (edge + 1) & 3;
        double nx = quad->p[b][1] - quad->p[a][1];
        double ny = -quad->p[b][0] + quad->p[a][0];
        double mag = sqrt(nx * nx + ny * ny);
        nx /= mag; ny /= mag;
        if (quad->reversed_border) { nx = -nx; ny = -ny; }
        int nsamples = (int)(mag / 8);
        if (nsamples < 16) nsamples = 16;
        double Mx = 0, My = 0, Mxx = 0, Mxy = 0, Myy = 0, N = 0;
        for (int s = 0; s < nsamples; s++) {
            double alpha = (1.0 + s) / (nsamples + 1);
            double x0 = alpha * quad->p[a][0] + (1 - alpha) * quad->p[b][0];
            double y0 = alpha * quad->p[a][1] + (1 - alpha) * quad->p[b][1];
            double Mn = 0, Mcount = 0;
            double range = decimate + 1;
            int steps = (int)(2 * range * 4) + 1; /* n = -range .. range step 0.25 (exact in binary) */
            for (int k = 0; k < steps; k++) {
                double n = -range + 0.25 * k;
                double grange = 1;
                /* PIX_EPS: a probe that lands on a pixel boundary to within rounding error reads the
                   pixel above it.  Edges that are exactly pixel-aligned put every fourth probe there. */
                int x1 = (int)(x0 + (n + grange) * nx + PIX_EPS), y1 = (int)(y0 + (n + grange) * ny + PIX_EPS);
                if (x1 < 0 || x1 >= w || y1 < 0 || y1 >= h) continue;
                int x2 = (int)(x0 + (n - grange) * nx + PIX_EPS), y2 = (int)(y0 + (n - grange) * ny + PIX_EPS);
                if (x2 < 0 || x2 >= w || y2 < 0 || y2 >= h) continue;
                int g1 = gray[(size_t)y1 * stride + x1], g2 = gray[(size_t)y2 * stride + x2];
                if (g1 < g2) continue;
                double weight = (double)((g2 - g1) * (g2 - g1));
                Mn += weight * n;
                Mcount += weight;
            }
            if (Mcount == 0) continue;
            double n0 = Mn / Mcount;
            double bestx = x0 + n0 * nx, besty = y0 + n0 * ny;
            Mx += bestx; My += besty; Mxx += bestx * bestx; Mxy += bestx * besty; Myy += besty * besty; N++;
        }
        double Ex = Mx / N, Ey = My / N;
        double Cxx = Mxx / N - Ex * Ex, Cxy = Mxy / N - Ex * Ey, Cyy = Myy / N - Ey * Ey;
        half_angle_normal(Cyy - Cxx, -2 * Cxy, &nx, &ny);
        lines[edge][0] = Ex; lines[edge][1] = Ey; lines[edge][2] = nx; lines[edge][3] = ny;
    }
    for (int i = 0; i < 4; i++) {
        double A00 = lines[i][3], A01 = -lines[(i + 1) & 3][3];
        double A10 = -lines[i][2], A11 = lines[(i + 1) & 3][2];
        double B0 = -lines[i][0] + lines[(i + 1) & 3][0];
        double B1 = -lines[i][1] + lines[(i + 1) & 3][1];
        double det = A00 * A11 - A10 * A01;
        if (fabs(det) > 0.001) {
            double W00 = A11 / det, W01 = -A01 / det;
            double L0 = W00 * B0 + W01 * B1;
            quad->p[(i + 1) & 3][0] = lines[i][0] + L0 * A00;
            quad->p[(i + 1) & 3][1] = lines[i][1] + L0 * A10;
        }
    }
}

/* ------------------------------------------------ S7 homography + decode */
/* tag frame corners (-1,-1),(1,-1),(1,1),(-1,1) -> quad p[0..3]; 8x9 elimination, partial pivoting */
static int homography_compute(const double p[4][2], double H[9])
{
    static const double cx[4] = {-1, 1, 1, -1}, cy[4] = {-1, -1, 1, 1};
    double A[72];
    for (int i = 0; i < 4; i++) {
        double x = cx[i], y = cy[i], u = p[i][0], v = p[i][1];
        double *r0 = A + 18 * i, *r1 = r0 + 9;
        r0[0] = x; r0[1] = y; r0[2] = 1; r0[3] = 0; r0[4] = 0; r0[5] = 0; r0[6] = -x * u; r0[7] = -y * u; r0[8] = u;
        r1[0] = 0; r1[1] = 0; r1[2] = 0; r1[3] = x; r1[4] = y; r1[5] = 1; r1[6] = -x * v; r1[7] = -y * v; r1[8] = v;
    }
    for (int col = 0; col < 8; col++) {
        double max_val = 0;
        int max_idx = -1;
        for (int row = col; row < 8; row++) {
            double val = fabs(A[row * 9 + col]);
            if (val > max_val) { max_val = val; max_idx = row; }
        }
        if (max_val < 1e-10) return 0;
        if (max_idx != col)
            for (int i = col; i < 9; i++) { double t = A[col * 9 + i]; A[col * 9 + i] = A[max_idx * 9 + i]; A[max_idx * 9 + i] = t; }
        for (int i = col + 1; i < 8; i++) {
            double f = A[i * 9 + col] / A[col * 9 + col];
            A[i * 9 + col] = 0;
            for (int j = col + 1; j < 9; j++) A[i * 9 + j] -= f * A[col * 9 + j];
        }
    }
    for (int col = 7; col >= 0; col--) {
        double sum = 0;
        for (int i = col + 1; i < 8; i++) sum += A[col * 9 + i] * A[i * 9 + 8];
        A[col * 9 + 8] = (A[col * 9 + 8] - sum) / A[col * 9 + col];
    }
    for (int i = 0; i < 8; i++) H[i] = A[i * 9 + 8];
    H[8] = 1;
    return 1;
}

static void hproject(const double H[9], double x, double y, double *ox, double *oy)
{
    double xx = H[0] * x + H[1] * y + H[2];
    double yy = H[3] * x + H[4] * y + H[5];
    double zz = H[6] * x + H[7] * y + H[8];
    *ox = xx / zz;
    *oy = yy / zz;
}

typedef struct { double A[9], B[3], C[3]; } graymodel;

static void gm_add(graymodel *gm, double x, double y, double gray)
{
    gm->A[0] += x * x; gm->A[1] += x * y; gm->A[2] += x; gm->A[4] += y * y; gm->A[5] += y; gm->A[8] += 1;
    gm->B[0] += x * gray; gm->B[1] += y * gray; gm->B[2] += gray;
}

static void gm_solve(graymodel *gm) /* 3x3 SPD solve via Cholesky */
{
    const double *A = gm->A, *B = gm->B;
    double L[9], M[9], t[3];
    L[0] = sqrt(A[0]); L[3] = A[1] / L[0]; L[6] = A[2] / L[0];
    L[4] = sqrt(A[4] - L[3] * L[3]); L[7] = (A[5] - L[3] * L[6]) / L[4];
    L[8] = sqrt(A[8] - L[6] * L[6] - L[7] * L[7]);
    M[0] = 1 / L[0]; M[3] = -L[3] * M[0] / L[4]; M[4] = 1 / L[4];
    M[6] = (-L[6] * M[0] - L[7] * M[3]) / L[8]; M[7] = -L[7] * M[4] / L[8]; M[8] = 1 / L[8];
    t[0] = M[0] * B[0]; t[1] = M[3] * B[0] + M[4] * B[1]; t[2] = M[6] * B[0] + M[7] * B[1] + M[8] * B[2];
    gm->C[0] = M[0] * t[0] + M[3] * t[1] + M[6] * t[2];
    gm->C[1] = M[4] * t[1] + M[7] * t[2];
    gm->C[2] = M[8] * t[2];
}

static double gm_interp(const graymodel *gm, double x, double y) { return gm->C[0] * x + gm->C[1] * y + gm->C[2]; }

static double value_for_pixel(const uint8_t *im, int w, int h, int stride, double px, double py)
{
    int x1 = (int)floor(px - 0.5), x2 = (int)ceil(px - 0.5);
    double x = px - 0.5 - x1;
    int y1 = (int)floor(py - 0.5), y2 = (int)ceil(py - 0.5);
    double y = py - 0.5 - y1;
    if (x1 < 0 || x2 >= w || y1 < 0 || y2 >= h) return -1;
    return im[(size_t)y1 * stride + x1] * (1 - x) * (1 - y) + im[(size_t)y1 * stride + x2] * x * (1 - y) +
           im[(size_t)y2 * stride + x1] * (1 - x) * y + im[(size_t)y2 * stride + x2] * x * y;
}

static uint64_t rotate90(uint64_t w, int nbits)
{
    int p = nbits;
    uint64_t l = 0;
    if (nbits % 4 == 1) { p = nbits - 1; l = 1; }
    w = ((w >> l) << (p / 4 + l)) | (w >> (3 * p / 4 + l) << l) | (w & l);
    w &= (((uint64_t)1 << nbits) - 1);
    return w;
}

static int popcount64(uint64_t v) { int c = 0; while (v) { v &= v - 1; c++; } return c; }

/* first rotation (0..3) of rcode that lies within maxhamming of a code book entry */
static void decode_codeword(const aso_family *fam, uint64_t rcode, int maxhamming, int *id, int *hamming, int *rotation)
{
    for (int ridx = 0; ridx < 4; ridx++) {
        int best = 255, bid = -1;
        for (int i = 0; i < fam->ncodes; i++) {
            int d = popcount64(rcode ^ fam->codes[i]);
            if (d < best) { best = d; bid = i; }
        }
        if (best <= maxhamming) { *id = bid; *hamming = best; *rotation = ridx; return; }
        rcode = rotate90(rcode, fam->nbits);
    }
    *id = -1; *hamming = 255; *rotation = 0;
}

int aso_decode_quad(const uint8_t *gray, int w, int h, int stride, const aso_family *fam, int maxhamming,
                    const aso_quad *q, aso_detection *det)
{
    double H[9];
    if (!homography_compute(q->p, H)) return 0;
    if (fam->reversed_border != q->reversed_border) return 0;
    int wb = fam->width_at_border, tw = fam->total_width;
    const double patterns[8][5] = {
        {-0.5, 0.5, 0, 1, 1}, {0.5, 0.5, 0, 1, 0}, {wb + 0.5, .5, 0, 1, 1}, {wb - 0.5, .5, 0, 1, 0},
        {0.5, -0.5, 1, 0, 1}, {0.5, 0.5, 1, 0, 0}, {0.5, wb + 0.5, 1, 0, 1}, {0.5, wb - 0.5, 1, 0, 0}};
    graymodel white, black;
    memset(&white, 0, sizeof white);
    memset(&black, 0, sizeof black);
    for (int pi = 0; pi < 8; pi++) {
        const double *pat = patterns[pi];
        int is_white = (int)pat[4];
        for (int i = 0; i < wb; i++) {
            double tagx01 = (pat[0] + i * pat[2]) / wb, tagy01 = (pat[1] + i * pat[3]) / wb;
            double tagx = 2 * (tagx01 - 0.5), tagy = 2 * (tagy01 - 0.5);
            double px, py;
            hproject(H, tagx, tagy, &px, &py);
            int ix = (int)px, iy = (int)py;
            if (ix < 0 || iy < 0 || ix >= w || iy >= h) continue;
            int v = gray[(size_t)iy * stride + ix];
            if (is_white) gm_add(&white, tagx, tagy, v); else gm_add(&black, tagx, tagy, v);
        }
    }
    gm_solve(&white);
    gm_solve(&black);
    if ((gm_interp(&white, 0, 0) - gm_interp(&black, 0, 0) < 0) != fam->reversed_border) return 0;

    double values[16 * 16];
    memset(values, 0, sizeof values);
    int min_coord = (wb - tw) / 2;
    for (int i = 0; i < fam->nbits; i++) {
        int bitx = fam->bit_x[i], bity = fam->bit_y[i];
        double tagx01 = (bitx + 0.5) / wb, tagy01 = (bity + 0.5) / wb;
        double tagx = 2 * (tagx01 - 0.5), tagy = 2 * (tagy01 - 0.5);
        double px, py;
        hproject(H, tagx, tagy, &px, &py);
        double v = value_for_pixel(gray, w, h, stride, px, py);
        if (v == -1) continue;
        double thresh = (gm_interp(&black, tagx, tagy) + gm_interp(&white, tagx, tagy)) / 2.0;
        values[tw * (bity - min_coord) + bitx - min_coord] = v - thresh;
    }
    { /* sharpen: values += 0.25 * laplacian(values) over the tw x tw grid */
        double sh[16 * 16];
        for (int y = 0; y < tw; y++)
            for (int x = 0; x < tw; x++) {
                double s = 0;
                /* kernel rows: (0,-1,0) (-1,4,-1) (0,-1,0), visited in row-major order */
                if (y - 1 >= 0) s += values[(y - 1) * tw + x] * -1.0;
                if (x - 1 >= 0) s += values[y * tw + x - 1] * -1.0;
                s += values[y * tw + x] * 4.0;
                if (x + 1 <= tw - 1) s += values[y * tw + x + 1] * -1.0;
                if (y + 1 <= tw - 1) s += values[(y + 1) * tw + x] * -1.0;
                sh[y * tw + x] = s;
            }
        for (int i = 0; i < tw * tw; i++) values[i] = values[i] + 0.25 * sh[i];
    }
    float black_score = 0, white_score = 0, black_count = 1, white_count = 1;
    uint64_t rcode = 0;
    for (int i = 0; i < fam->nbits; i++) {
        int bitx = fam->bit_x[i], bity = fam->bit_y[i];
        rcode <<= 1;
        double v = values[(bity - min_coord) * tw + bitx - min_coord];
        if (v > 0) { white_score += v; white_count++; rcode |= 1; }
        else { black_score -= v; black_count++; }
    }
    int id, hamming, rotation;
    decode_codeword(fam, rcode, maxhamming, &id, &hamming, &rotation);
    float margin = fminf(white_score / white_count, black_score / black_count);
    if (!(margin >= 0) || hamming >= 255) return 0;

    /* orient: H <- H * Rz(rotation * 90deg), exact entries */
    static const double C[4] = {1, 0, -1, 0}, S[4] = {0, 1, 0, -1};
    double c = C[rotation], s = S[rotation], Hr[9];
    for (int r = 0; r < 3; r++) {
        Hr[3 * r + 0] = H[3 * r + 0] * c + H[3 * r + 1] * s;
        Hr[3 * r + 1] = H[3 * r + 0] * -s + H[3 * r + 1] * c;
        Hr[3 * r + 2] = H[3 * r + 2];
    }
    det->id = id;
    det->hamming = hamming;
    det->margin = margin;
    det->reserved = 0;
    hproject(Hr, 0, 0, &det->center[0], &det->center[1]);
    for (int i = 0; i < 4; i++) {
        int tcx = (i == 1 || i == 2) ? 1 : -1, tcy = (i < 2) ? 1 : -1;
        hproject(Hr, tcx, tcy, &det->corners[i][0], &det->corners[i][1]);
    }
    return 1;
}

/* ------------------------------------------------------ S8 dedup and sort */
static int seg_intersect(const double *a, const double *b, const double *c, const double *d)
{
    double d1 = (b[0] - a[0]) * (c[1] - a[1]) - (b[1] - a[1]) * (c[0] - a[0]);
    double d2 = (b[0] - a[0]) * (d[1] - a[1]) - (b[1] - a[1]) * (d[0] - a[0]);
    double d3 = (d[0] - c[0]) * (a[1] - c[1]) - (d[1] - c[1]) * (a[0] - c[0]);
    double d4 = (d[0] - c[0]) * (b[1] - c[1]) - (d[1] - c[1]) * (b[0] - c[0]);
    return ((d1 > 0) != (d2 > 0)) && ((d3 > 0) != (d4 > 0));
}
static int point_in_quad(const double q[4][2], const double *p)
{
    int pos = 0, neg = 0;
    for (int i = 0; i < 4; i++) {
        const double *a = q[i], *b = q[(i + 1) & 3];
        double c = (b[0] - a[0]) * (p[1] - a[1]) - (b[1] - a[1]) * (p[0] - a[0]);
        if (c > 0) pos++; else if (c < 0) neg++;
    }
    return pos == 0 || neg == 0;
}
static int quads_overlap(const double a[4][2], const double b[4][2])
{
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++)
            if (seg_intersect(a[i], a[(i + 1) & 3], b[j], b[(j + 1) & 3])) return 1;
    return point_in_quad(a, b[0]) || point_in_quad(b, a[0]);
}
static int prefer_smaller(int pref, double q0, double q1)
{
    if (pref) return pref;
    if (q0 < q1) return -1;
    if (q1 < q0) return 1;
    return 0;
}
static int det_cmp(const void *pa, const void *pb)
{
    const aso_detection *a = pa, *b = pb;
    if (a->id != b->id) return a->id < b->id ? -1 : 1;
    if (a->hamming != b->hamming) return a->hamming < b->hamming ? -1 : 1;
    for (int i = 0; i < 4; i++)
        for (int k = 0; k < 2; k++)
            if (a->corners[i][k] != b->corners[i][k]) return a->corners[i][k] < b->corners[i][k] ? -1 : 1;
    return 0;
}
static int dedup_and_sort(aso_detection *d, int n)
{
    for (int i0 = 0; i0 < n; i0++) {
        for (int i1 = i0 + 1; i1 < n; i1++) {
            if (d[i0].id != d[i1].id) continue;
            if (!quads_overlap(d[i0].corners, d[i1].corners)) continue;
            int pref = 0;
            pref = prefer_smaller(pref, d[i0].hamming, d[i1].hamming);
            pref = prefer_smaller(pref, -d[i0].margin, -d[i1].margin);
            for (int i = 0; i < 4; i++) {
                pref = prefer_smaller(pref, d[i0].corners[i][0], d[i1].corners[i][0]);
                pref = prefer_smaller(pref, d[i0].corners[i][1], d[i1].corners[i][1]);
            }
            if (pref < 0) { /* keep i0, drop i1 */
                memmove(&d[i1], &d[i1 + 1], sizeof(*d) * (n - i1 - 1));
                n--; i1--;
            } else { /* keep i1, drop i0 */
                memmove(&d[i0], &d[i0 + 1], sizeof(*d) * (n - i0 - 1));
                n--; i0--;
                break;
            }
        }
    }
    qsort(d, (size_t)n, sizeof(*d), det_cmp);
    return n;
}

/* ---------------------------------------------------------- full detector */
int aso_detect_gray(const uint8_t *gray, int w, int h, int stride, const aso_family *fam,
                    const aso_params *prm, aso_detection *out, int cap)
{
    int f = prm->decimate < 1 ? 1 : prm->decimate;
    int sw, sh;
    uint8_t *dec = malloc((size_t)w * h);
    aso_decimate(gray, w, h, stride, f, dec, &sw, &sh);
    size_t n = (size_t)sw * sh;
    uint8_t *th = malloc(n);
    aso_threshold(dec, sw, sh, th);
    uint32_t *labels = malloc(n * 4), *sizes = malloc(n * 4);
    aso_connected_components(th, sw, sh, labels, sizes);
    long capp = (long)n * 4;
    aso_point *pts = malloc(sizeof(aso_point) * (size_t)capp);
    long npts = aso_gradient_clusters(th, sw, sh, labels, sizes, pts, capp);
    int qcap = 4096;
    aso_quad *quads = malloc(sizeof(aso_quad) * qcap);
    int nq = aso_fit_quads(dec, sw, sh, pts, npts, fam, f, quads, qcap);
    int nd = 0;
    for (int i = 0; i < nq && nd < cap; i++) {
        aso_quad *q = &quads[i];
        if (f > 1)
            for (int j = 0; j < 4; j++) {
                q->p[j][0] = (q->p[j][0] - 0.5) * f + 0.5;
                q->p[j][1] = (q->p[j][1] - 0.5) * f + 0.5;
            }
        if (prm->refine_edges) aso_refine_edges(gray, w, h, stride, f, q);
        if (aso_decode_quad(gray, w, h, stride, fam, prm->maxhamming, q, &out[nd])) nd++;
    }
    nd = dedup_and_sort(out, nd);
    free(dec); free(th); free(labels); free(sizes); free(pts); free(quads);
    return nd;
}

int aso_detect_bgr(const uint8_t *bgr, int w, int h, int stride, const aso_family *fam,
                   const aso_params *prm, aso_detection *out, int cap)
{
    uint8_t *gray = malloc((size_t)w * h);
    aso_bgr2gray(bgr, w, h, stride, gray);
    int n = aso_detect_gray(gray, w, h, w, fam, prm, out, cap);
    free(gray);
    return n;
}
