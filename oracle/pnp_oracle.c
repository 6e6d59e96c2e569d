/*
 * oracle/pnp_oracle.c -- TEST INFRASTRUCTURE ONLY (see apriltag_oracle.h).
 *
 * CPU restatement of what the reference obtains from
 *   cv2.solvePnP(obj_points, corners, K, dist)   tag_detector.py:41  (flag ITERATIVE)
 *   cv2.Rodrigues(rvec)                          tag_detector.py:47
 * for the 4 coplanar tag corners of tag_detector.py:35-38.  OpenCV is not in this
 * container; the published ITERATIVE algorithm for a planar target is followed:
 *   1. undistort the image points to normalised coordinates (5 fixed-point sweeps),
 *   2. exact 4-point homography  plane(X,Y) -> normalised image,
 *   3. pose from the homography (unit first two columns, t = 2*h3/(|h1|+|h2|),
 *      third column = cross product, nearest rotation by SVD),
 *   4. Levenberg-Marquardt on the 8 pixel reprojection residuals over 6 dof.
 * Step 4 here runs to convergence in float64 within OpenCV's cap of 20 outer iterations, i.e. to the
 * local minimum that OpenCV's LM approaches from the same start; parity with cv2 is
 * therefore tolerance-level and has no fixture ("parity unpinned").
 */
#include "apriltag_oracle.h"

#include <float.h>
#include <math.h>
#include <string.h>

static void mat3_mul(const double *A, const double *B, double *C)
{
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++)
            C[3 * i + j] = A[3 * i] * B[j] + A[3 * i + 1] * B[3 + j] + A[3 * i + 2] * B[6 + j];
}

void aso_rodrigues(const double r[3], double R[9])
{
    double theta = sqrt(r[0] * r[0] + r[1] * r[1] + r[2] * r[2]);
    if (theta < DBL_EPSILON) {
        R[0] = 1; R[1] = 0; R[2] = 0; R[3] = 0; R[4] = 1; R[5] = 0; R[6] = 0; R[7] = 0; R[8] = 1;
        return;
    }
    double c = cos(theta), s = sin(theta), c1 = 1 - c;
    double x = r[0] / theta, y = r[1] / theta, z = r[2] / theta;
    R[0] = c + c1 * x * x;     R[1] = c1 * x * y - s * z; R[2] = c1 * x * z + s * y;
    R[3] = c1 * x * y + s * z; R[4] = c + c1 * y * y;     R[5] = c1 * y * z - s * x;
    R[6] = c1 * x * z - s * y; R[7] = c1 * y * z + s * x; R[8] = c + c1 * z * z;
}

/* rotation matrix -> Rodrigues vector, robust at theta ~ pi (the usual case here:
   a tag facing the camera is a half-turn about x, cf. Est_Roll = 3.14151 in
   data/csv/slam_clustered_data.csv:2) */
static void rot_to_rvec(const double *R, double r[3])
{
    double rx = R[7] - R[5], ry = R[2] - R[6], rz = R[3] - R[1];
    double s = sqrt((rx * rx + ry * ry + rz * rz) * 0.25);
    double c = (R[0] + R[4] + R[8] - 1) * 0.5;
    c = c > 1 ? 1 : (c < -1 ? -1 : c);
    double theta = atan2(s, c);
    if (s < 1e-5) {
        if (c > 0) { r[0] = r[1] = r[2] = 0; return; }
        double t;
        t = (R[0] + 1) * 0.5; rx = sqrt(t > 0 ? t : 0);
        t = (R[4] + 1) * 0.5; ry = sqrt(t > 0 ? t : 0) * (R[1] < 0 ? -1.0 : 1.0);
        t = (R[8] + 1) * 0.5; rz = sqrt(t > 0 ? t : 0) * (R[2] < 0 ? -1.0 : 1.0);
        if (fabs(rx) < fabs(ry) && fabs(rx) < fabs(rz) && (R[5] > 0) != (ry * rz > 0)) rz = -rz;
        theta /= sqrt(rx * rx + ry * ry + rz * rz);
        r[0] = theta * rx; r[1] = theta * ry; r[2] = theta * rz;
        return;
    }
    double vth = 1 / (2 * s) * theta;
    r[0] = rx * vth; r[1] = ry * vth; r[2] = rz * vth;
}

/* symmetric 3x3 Jacobi eigen-decomposition: A = V diag(d) V^T */
static void jacobi3(double A[9], double V[9], double d[3])
{
    for (int i = 0; i < 9; i++) V[i] = (i % 4 == 0);
    for (int sweep = 0; sweep < 60; sweep++) {
        double off = fabs(A[1]) + fabs(A[2]) + fabs(A[5]);
        /* relative stop: below this the remaining rotations are identity in float64 */
        if (off < 1e-15 * (fabs(A[0]) + fabs(A[4]) + fabs(A[8]))) break;
        for (int p = 0; p < 2; p++)
            for (int q = p + 1; q < 3; q++) {
                double apq = A[3 * p + q];
                if (fabs(apq) < 1e-300) continue;
                double theta = (A[3 * q + q] - A[3 * p + p]) / (2 * apq);
                double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1));
                double c = 1 / sqrt(t * t + 1), s = t * c;
                for (int k = 0; k < 3; k++) { /* A <- A J */
                    double akp = A[3 * k + p], akq = A[3 * k + q];
                    A[3 * k + p] = c * akp - s * akq;
                    A[3 * k + q] = s * akp + c * akq;
                }
                for (int k = 0; k < 3; k++) { /* A <- J^T A */
                    double apk = A[3 * p + k], aqk = A[3 * q + k];
                    A[3 * p + k] = c * apk - s * aqk;
                    A[3 * q + k] = s * apk + c * aqk;
                }
                for (int k = 0; k < 3; k++) {
                    double vkp = V[3 * k + p], vkq = V[3 * k + q];
                    V[3 * k + p] = c * vkp - s * vkq;
                    V[3 * k + q] = s * vkp + c * vkq;
                }
            }
    }
    d[0] = A[0]; d[1] = A[4]; d[2] = A[8];
}

/* nearest rotation to M (det > 0 assumed): R = M (M^T M)^(-1/2) */
static void nearest_rotation(const double *M, double *R)
{
    double S[9], V[9], d[3], Mt[9], W[9], T[9];
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) Mt[3 * i + j] = M[3 * j + i];
    mat3_mul(Mt, M, S);
    jacobi3(S, V, d);
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) {
            double acc = 0;
            for (int k = 0; k < 3; k++) acc += V[3 * i + k] * (1 / sqrt(d[k] > 1e-300 ? d[k] : 1e-300)) * V[3 * j + k];
            W[3 * i + j] = acc;
        }
    mat3_mul(M, W, T);
    memcpy(R, T, sizeof T);
}

/* n x n linear solve, partial pivoting; returns 0 if singular */
static int solve_n(double *A, double *b, int n)
{
    for (int col = 0; col < n; col++) {
        int piv = col;
        double mx = fabs(A[col * n + col]);
        for (int r = col + 1; r < n; r++) if (fabs(A[r * n + col]) > mx) { mx = fabs(A[r * n + col]); piv = r; }
        if (mx < 1e-300) return 0;
        if (piv != col) {
            for (int j = 0; j < n; j++) { double t = A[col * n + j]; A[col * n + j] = A[piv * n + j]; A[piv * n + j] = t; }
            double t = b[col]; b[col] = b[piv]; b[piv] = t;
        }
        for (int r = col + 1; r < n; r++) {
            double f = A[r * n + col] / A[col * n + col];
            for (int j = col; j < n; j++) A[r * n + j] -= f * A[col * n + j];
            b[r] -= f * b[col];
        }
    }
    for (int col = n - 1; col >= 0; col--) {
        double s = b[col];
        for (int j = col + 1; j < n; j++) s -= A[col * n + j] * b[j];
        b[col] = s / A[col * n + col];
    }
    return 1;
}

typedef struct { double fx, fy, cx, cy, k1, k2, p1, p2, k3; } cam_t;

static void undistort_point(const cam_t *c, double u, double v, double *xo, double *yo)
{
    double x0 = (u - c->cx) / c->fx, y0 = (v - c->cy) / c->fy, x = x0, y = y0;
    if (c->k1 != 0 || c->k2 != 0 || c->p1 != 0 || c->p2 != 0 || c->k3 != 0)
        for (int it = 0; it < 5; it++) {
            double r2 = x * x + y * y;
            double icdist = 1 / (1 + ((c->k3 * r2 + c->k2) * r2 + c->k1) * r2);
            double dx = 2 * c->p1 * x * y + c->p2 * (r2 + 2 * x * x);
            double dy = c->p1 * (r2 + 2 * y * y) + 2 * c->p2 * x * y;
            x = (x0 - dx) * icdist;
            y = (y0 - dy) * icdist;
        }
    *xo = x; *yo = y;
}

/* pixel projection of a camera-frame point and its 2x3 Jacobian w.r.t. the point */
static void project(const cam_t *c, const double P[3], double uv[2], double J[6])
{
    double iz = 1 / P[2], x = P[0] * iz, y = P[1] * iz;
    double r2 = x * x + y * y;
    double cd = 1 + ((c->k3 * r2 + c->k2) * r2 + c->k1) * r2;
    double dcd = c->k1 + r2 * (2 * c->k2 + 3 * c->k3 * r2); /* d cd / d r2 */
    double xd = x * cd + 2 * c->p1 * x * y + c->p2 * (r2 + 2 * x * x);
    double yd = y * cd + c->p1 * (r2 + 2 * y * y) + 2 * c->p2 * x * y;
    uv[0] = c->fx * xd + c->cx;
    uv[1] = c->fy * yd + c->cy;
    if (!J) return;
    double dxd_dx = cd + x * dcd * 2 * x + 2 * c->p1 * y + c->p2 * (2 * x + 4 * x);
    double dxd_dy = x * dcd * 2 * y + 2 * c->p1 * x + c->p2 * 2 * y;
    double dyd_dx = y * dcd * 2 * x + c->p1 * 2 * x + 2 * c->p2 * y;
    double dyd_dy = cd + y * dcd * 2 * y + c->p1 * (2 * y + 4 * y) + 2 * c->p2 * x;
    /* d(x,y)/dP */
    double dx_dP[3] = {iz, 0, -x * iz}, dy_dP[3] = {0, iz, -y * iz};
    for (int k = 0; k < 3; k++) {
        J[k] = c->fx * (dxd_dx * dx_dP[k] + dxd_dy * dy_dP[k]);
        J[3 + k] = c->fy * (dyd_dx * dx_dP[k] + dyd_dy * dy_dP[k]);
    }
}

static double residuals(const cam_t *c, const double *R, const double *t, const double obj[4][3],
                        const double *img, double *res, double *Jac /* 8x6 or NULL */)
{
    double cost = 0;
    for (int i = 0; i < 4; i++) {
        double RX[3], P[3], uv[2], Jp[6];
        for (int r = 0; r < 3; r++) {
            RX[r] = R[3 * r] * obj[i][0] + R[3 * r + 1] * obj[i][1] + R[3 * r + 2] * obj[i][2];
            P[r] = RX[r] + t[r];
        }
        project(c, P, uv, Jac ? Jp : NULL);
        res[2 * i] = uv[0] - img[2 * i];
        res[2 * i + 1] = uv[1] - img[2 * i + 1];
        cost += res[2 * i] * res[2 * i] + res[2 * i + 1] * res[2 * i + 1];
        if (Jac) {
            /* P(dw) = exp([dw]x) R X + t  ->  dP/dw = -[RX]x ,  dP/dt = I */
            double dPdw[9] = {0, RX[2], -RX[1], -RX[2], 0, RX[0], RX[1], -RX[0], 0};
            for (int row = 0; row < 2; row++) {
                double *Jr = Jac + 6 * (2 * i + row);
                const double *jp = Jp + 3 * row;
                for (int k = 0; k < 3; k++) {
                    Jr[k] = jp[0] * dPdw[k] + jp[1] * dPdw[3 + k] + jp[2] * dPdw[6 + k];
                    Jr[3 + k] = jp[k];
                }
            }
        }
    }
    return cost;
}

static int pnp_one(const double *img /*4x2*/, const cam_t *c, double tag_size, double *rvec, double *tvec, double *T)
{
    double s = tag_size / 2;
    /* object corners are stored as float32 by the reference (tag_detector.py:35-38) */
    float sf = (float)s;
    double sd = (double)sf;
    const double obj[4][3] = {{-sd, -sd, 0}, {sd, -sd, 0}, {sd, sd, 0}, {-sd, sd, 0}};

    /* 1-2: homography plane -> normalised image */
    double A[64], b[8];
    for (int i = 0; i < 4; i++) {
        double xn, yn;
        undistort_point(c, img[2 * i], img[2 * i + 1], &xn, &yn);
        double X = obj[i][0], Y = obj[i][1];
        double *r0 = A + 16 * i, *r1 = r0 + 8;
        r0[0] = X; r0[1] = Y; r0[2] = 1; r0[3] = 0; r0[4] = 0; r0[5] = 0; r0[6] = -X * xn; r0[7] = -Y * xn; b[2 * i] = xn;
        r1[0] = 0; r1[1] = 0; r1[2] = 0; r1[3] = X; r1[4] = Y; r1[5] = 1; r1[6] = -X * yn; r1[7] = -Y * yn; b[2 * i + 1] = yn;
    }
    if (!solve_n(A, b, 8)) return 0;
    double H[9] = {b[0], b[1], b[2], b[3], b[4], b[5], b[6], b[7], 1};

    /* 3: pose from homography */
    double h1n = sqrt(H[0] * H[0] + H[3] * H[3] + H[6] * H[6]);
    double h2n = sqrt(H[1] * H[1] + H[4] * H[4] + H[7] * H[7]);
    double i1 = 1 / (h1n > DBL_EPSILON ? h1n : DBL_EPSILON), i2 = 1 / (h2n > DBL_EPSILON ? h2n : DBL_EPSILON);
    double ts = 2 / ((h1n + h2n) > DBL_EPSILON ? (h1n + h2n) : DBL_EPSILON);
    double M[9], R[9], t[3];
    for (int r = 0; r < 3; r++) { M[3 * r] = H[3 * r] * i1; M[3 * r + 1] = H[3 * r + 1] * i2; t[r] = H[3 * r + 2] * ts; }
    M[2] = M[3] * M[7] - M[6] * M[4];
    M[5] = M[6] * M[1] - M[0] * M[7];
    M[8] = M[0] * M[4] - M[3] * M[1];
    nearest_rotation(M, R);

    /* 4: Levenberg-Marquardt on pixel reprojection error */
    double res[8], J[48], cost = residuals(c, R, t, obj, img, res, J);
    double lambda = 1e-3;
    for (int it = 0; it < 20; it++) { /* OpenCV caps its LM at 20 iterations as well */
        double JtJ[36], g[6];
        for (int a = 0; a < 6; a++) {
            g[a] = 0;
            for (int k = 0; k < 8; k++) g[a] += J[6 * k + a] * res[k];
            for (int bb = 0; bb < 6; bb++) {
                double acc = 0;
                for (int k = 0; k < 8; k++) acc += J[6 * k + a] * J[6 * k + bb];
                JtJ[6 * a + bb] = acc;
            }
        }
        int improved = 0;
        double step_norm = 0;
        for (int tries = 0; tries < 12 && !improved; tries++) {
            double Aa[36], d[6];
            memcpy(Aa, JtJ, sizeof Aa);
            for (int a = 0; a < 6; a++) { Aa[6 * a + a] += lambda * (JtJ[6 * a + a] > 1e-12 ? JtJ[6 * a + a] : 1e-12); d[a] = -g[a]; }
            if (!solve_n(Aa, d, 6)) { lambda *= 10; continue; }
            double dR[9], Rn[9], tn[3] = {t[0] + d[3], t[1] + d[4], t[2] + d[5]}, resn[8];
            aso_rodrigues(d, dR);
            mat3_mul(dR, R, Rn);
            double costn = residuals(c, Rn, tn, obj, img, resn, NULL);
            if (costn < cost) {
                memcpy(R, Rn, sizeof Rn); memcpy(t, tn, sizeof tn);
                step_norm = sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2] + d[3] * d[3] + d[4] * d[4] + d[5] * d[5]);
                improved = 1;
                lambda *= 0.1;
                if (lambda < 1e-12) lambda = 1e-12;
            } else {
                /* a rejected step this small means the minimum is reached: more damping cannot help */
                double dn = sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2] + d[3] * d[3] + d[4] * d[4] + d[5] * d[5]);
                if (dn < 1e-10 * (sqrt(t[0] * t[0] + t[1] * t[1] + t[2] * t[2]) + 1)) break;
                lambda *= 10;
            }
        }
        if (!improved) break;
        double prev = cost;
        cost = residuals(c, R, t, obj, img, res, J);
        double scale = sqrt(t[0] * t[0] + t[1] * t[1] + t[2] * t[2]) + 1;
        if (step_norm < 1e-10 * scale || prev - cost < 1e-15 * (1 + prev)) break;
    }
    nearest_rotation(R, R);
    rot_to_rvec(R, rvec);
    tvec[0] = t[0]; tvec[1] = t[1]; tvec[2] = t[2];
    double Rr[9];
    aso_rodrigues(rvec, Rr); /* tag_detector.py:45-52: T is rebuilt from rvec */
    for (int r = 0; r < 3; r++) {
        T[4 * r] = Rr[3 * r]; T[4 * r + 1] = Rr[3 * r + 1]; T[4 * r + 2] = Rr[3 * r + 2]; T[4 * r + 3] = t[r];
    }
    T[12] = 0; T[13] = 0; T[14] = 0; T[15] = 1;
    for (int k = 0; k < 3; k++) if (!isfinite(rvec[k]) || !isfinite(tvec[k])) return 0;
    return 1;
}

void aso_solve_pnp(const double *corners, int n, const double *K, const double *dist, int ndist,
                   double tag_size, double *rvec, double *tvec, double *T, uint8_t *ok)
{
    cam_t c = {K[0], K[4], K[2], K[5], 0, 0, 0, 0, 0};
    if (ndist >= 4) { c.k1 = dist[0]; c.k2 = dist[1]; c.p1 = dist[2]; c.p2 = dist[3]; }
    if (ndist >= 5) c.k3 = dist[4];
    for (int i = 0; i < n; i++)
        ok[i] = (uint8_t)pnp_one(corners + 8 * i, &c, tag_size, rvec + 3 * i, tvec + 3 * i, T + 16 * i);
}
