/*
 * oracle/apriltag_oracle.h -- TEST INFRASTRUCTURE ONLY.
 *
 * Scalar CPU restatement of the per-frame hot path that the reference reaches
 * through `TagDetector.detect` / `TagDetector.get_pose`
 * (/root/reference/src/detection/tag_detector.py:23-52).  The arithmetic of that
 * path lives in two third-party native packages that are NOT vendored in the
 * reference and NOT present in this container:
 *   - AprilRobotics/apriltag (C, version unpinned: README.md:46 "git clone" HEAD,
 *     requirements.txt:2) -- call sites tag_detector.py:11,18,26
 *   - opencv-python >= 4.5.0 (requirements.txt:5) -- call sites tag_detector.py:25,41,47
 * so this file restates their PUBLISHED algorithms (Olson ICRA'11, Wang & Olson
 * IROS'16, Krogius et al. IROS'19; OpenCV solvePnP ITERATIVE on a planar target).
 *
 * PARITY STATUS: the tag family bit layout and codes 0..4 are pinned by the
 * reference's own assets (assets/tags/tag{0..4}.png -> tests/golden/tag_grids.json);
 * end-to-end poses are pinned at tolerance level by the reference's committed run
 * (data/csv/slam_clustered_data.csv:2, data/logs/simulation_runner.log:26-27).
 * Sub-pixel corners, hamming/margin and rvec/tvec have no reference fixture:
 * "parity unpinned" for those.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use
 * anything in oracle/.  The product (aprilslam_amd/) never links or imports it.
 */
#ifndef APRILTAG_ORACLE_H
#define APRILTAG_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct {
    int32_t id;
    int32_t hamming;
    float margin;
    int32_t reserved;
    double center[2];
    double corners[4][2]; /* lb, rb, rt, lt  (tag_detector.py:32) */
} aso_detection;

typedef struct {
    int nbits;            /* 41 */
    int width_at_border;  /* 5 */
    int total_width;      /* 9 */
    int reversed_border;  /* 1 */
    int ncodes;
    const uint64_t *codes;
    const int *bit_x;
    const int *bit_y;
} aso_family;

typedef struct {
    int decimate;      /* integer factor >= 1 (wrapper default 2) */
    int maxhamming;    /* wrapper default 1 */
    int refine_edges;  /* wrapper default 1 */
} aso_params;

/* boundary point between a black and a white component (half-pixel fixed point) */
typedef struct {
    uint64_t cluster; /* (max(rep0,rep1) << 32) + min(rep0,rep1), reps = min raster index */
    uint16_t x, y;
    int16_t gx, gy;
} aso_point;

typedef struct {
    double p[4][2];
    int reversed_border;
    uint64_t cluster;
} aso_quad;

/* S0: cv2.cvtColor(BGR2GRAY) fixed point, tag_detector.py:25 */
void aso_bgr2gray(const uint8_t *bgr, int w, int h, int stride, uint8_t *gray);
/* S1: integer decimation (top-left pixel of each f x f cell) */
void aso_decimate(const uint8_t *gray, int w, int h, int stride, int f, uint8_t *out, int *sw, int *sh);
/* S2: 4x4-tile min/max adaptive threshold -> {0,127,255} */
void aso_threshold(const uint8_t *im, int w, int h, uint8_t *out);
/* S3: connected components.  labels[p] = smallest raster index in p's component; sizes indexed by label */
void aso_connected_components(const uint8_t *th, int w, int h, uint32_t *labels, uint32_t *sizes);
/* S4: boundary points, sorted by (cluster, y, x, gx, gy).  Returns count (<= cap), or -needed if cap too small */
long aso_gradient_clusters(const uint8_t *th, int w, int h, const uint32_t *labels, const uint32_t *sizes,
                           aso_point *out, long cap);
/* S5: fit quads to all clusters of the decimated image; corners in decimated pixel coords */
int aso_fit_quads(const uint8_t *dec, int w, int h, const aso_point *pts, long npts,
                  const aso_family *fam, int decimate, aso_quad *out, int cap);
/* S6: refine quad edges on the full-resolution gray image (in place) */
void aso_refine_edges(const uint8_t *gray, int w, int h, int stride, int decimate, aso_quad *q);
/* S7: homography + decode one quad; returns 1 and fills det on success */
int aso_decode_quad(const uint8_t *gray, int w, int h, int stride, const aso_family *fam, int maxhamming,
                    const aso_quad *q, aso_detection *det);
/* full detector (S1..S8) on a gray image; detections sorted by id */
int aso_detect_gray(const uint8_t *gray, int w, int h, int stride, const aso_family *fam,
                    const aso_params *prm, aso_detection *out, int cap);
/* S0 + full detector on a BGR image */
int aso_detect_bgr(const uint8_t *bgr, int w, int h, int stride, const aso_family *fam,
                   const aso_params *prm, aso_detection *out, int cap);

/* S9: planar PnP (tag_detector.py:30-52).  corners: n x 4 x 2 doubles (already rounded to f32 by caller),
 * K row-major 3x3, dist: k1,k2,p1,p2[,k3] (ndist in {0,4,5}).  Outputs per tag: rvec[3], tvec[3], T[16], ok. */
void aso_solve_pnp(const double *corners, int n, const double *K, const double *dist, int ndist,
                   double tag_size, double *rvec, double *tvec, double *T, uint8_t *ok);

/* Rodrigues vector -> rotation matrix (row-major 3x3) */
void aso_rodrigues(const double r[3], double R[9]);

#ifdef __cplusplus
}
#endif
#endif
