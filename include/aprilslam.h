/*
 * include/aprilslam.h -- C ABI of libaprilslam.so, the MI355X (gfx950) AprilTag-SLAM hot path.
 *
 * Drop-in boundary.  The reference reaches its hot path through two native packages:
 *   apriltag(tag_type).detect(gray)                 reference src/detection/tag_detector.py:11,18,26
 *   cv2.cvtColor(image, cv2.COLOR_BGR2GRAY)         reference src/detection/tag_detector.py:25
 *   cv2.solvePnP(obj_points, corners, K, dist)      reference src/detection/tag_detector.py:41
 *   cv2.Rodrigues(rvec)                             reference src/detection/tag_detector.py:47
 * Upstream's `apriltag` module is a CPython extension, so there is no existing FFI
 * signature to copy; the entry points below are what a ctypes binding for that module
 *  needs.  Plain pointers and sizes only; no torch / HIP types.
 *
 * Conventions
 *   - every function returns 0 on success and a negative ASL_E* code on failure;
 *     asl_last_error() returns a thread-local message for the last failure.
 *   - "host" pointers are ordinary process memory, "device" pointers are HIP device
 *     memory on the detector's GPU (e.g. torch.Tensor.data_ptr()).
 *   - `stream` is a hipStream_t passed as void* (NULL = the default stream).
 *   - images are row-major uint8, `stride` = bytes between rows, 1 channel (gray) or
 *     3 channels (BGR, as cv2 hands them to TagDetector.detect).
 *   - pixel convention: pixel (ix, iy) covers [ix, ix+1) x [iy, iy+1), centre at +0.5.
 *   - one detector per (host thread, stream); a detector is not re-entrant
 *     (same rule as upstream's detector object).
 *   - there is NO CPU fallback: without a usable gfx950 device every call fails.
 */
#ifndef APRILSLAM_H
#define APRILSLAM_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ASL_OK 0
#define ASL_EINVAL (-1)      /* bad argument (unknown family, unsupported decimate/blur, NULL pointer ...) */
#define ASL_EDEVICE (-2)     /* HIP runtime / no GPU */
#define ASL_ECAPACITY (-3)   /* an internal work buffer overflowed even after growing */
#define ASL_ENOMEM (-4)

typedef struct asl_detector asl_detector;

/* One detection: the fields of the dict upstream's wrapper returns
   ('id', 'hamming', 'margin', 'center', 'lb-rb-rt-lt'), plus the frame index inside a batch. */
typedef struct {
    int32_t id;
    int32_t hamming;
    float margin;
    int32_t frame;
    double center[2];
    double corners[4][2]; /* lb, rb, rt, lt in pixels (reference tag_detector.py:32) */
} asl_detection;

/* Pose of one detection: what TagDetector.get_pose returns (reference tag_detector.py:30-43). */
typedef struct {
    double rvec[3];
    double tvec[3];
    double T[16]; /* row-major camera<-tag 4x4 (reference tag_detector.py:45-52) */
    int32_t ok;   /* solvePnP's retval */
    int32_t reserved;
} asl_pose;

/* Replaces `apriltag(family, threads=1, maxhamming=1, decimate=2.0, blur=0.0, refine_edges=True)`
   (reference tag_detector.py:18 passes the family only; the wrapper's defaults apply).
   decimate must be an integer value >= 1; blur must be 0 (the reference never sets either).
   device = HIP device ordinal. */
int asl_detector_create(const char *family, int nthreads, int maxhamming, float decimate, float blur,
                        int refine_edges, int device, asl_detector **out);
void asl_detector_destroy(asl_detector *det);
/* Which ids the decoder may return.  Only ids 0..4 of tagStandard41h12 are pinned by the reference (its
   assets/tags/tag{0..4}.png); upstream's 2115-entry code table is not in the reference tree, so the other
   entries of this library's table are build-defined and would mislabel a physical tag with id >= 5.  A new
   detector therefore decodes ids 0..4 only.  n_ids <= 0 opens the whole table (synthetic scenes rendered from
   the same table), n_ids > 0 keeps ids 0..n_ids-1. */
int asl_detector_set_id_limit(asl_detector *det, int n_ids);
/* A planar tag has two poses that reproject almost equally well.  cv2.solvePnP(ITERATIVE), which the reference calls
   (tag_detector.py:41), returns the one its homography start leads to, and so does this library by default (enabled = 0).
   enabled = 1 refines the mirrored pose as well and keeps the one with the lower reprojection error (IPPE's
   two-solution test): better orientation for small, near-frontal tags, but no longer what the reference computes. */
int asl_detector_set_pnp_both_minima(asl_detector *det, int enabled);
const char *asl_last_error(void);
/* "aprilslam <version> gfx950 ..." */
const char *asl_version(void);

/* Replaces detector.detect(gray) (reference tag_detector.py:26).  Host image in, detections
   (sorted by id) out.  *n_out = number found; at most max_out are written. */
int asl_detect_gray_u8(asl_detector *det, const uint8_t *gray, int w, int h, int stride,
                       asl_detection *out, int max_out, int *n_out);
/* Replaces cv2.cvtColor(BGR2GRAY) + detector.detect (reference tag_detector.py:25-27): the
   gray conversion is fused into the first kernel. */
int asl_detect_bgr_u8(asl_detector *det, const uint8_t *bgr, int w, int h, int stride,
                      asl_detection *out, int max_out, int *n_out);

/* Batched form of the two calls above: n_frames host images of identical geometry.
   channels = 1 (gray) or 3 (BGR).  out holds up to max_out detections in total, ordered by
   (frame, id); n_per_frame[n_frames] receives the count for each frame. */
int asl_detect_batch_u8(asl_detector *det, const uint8_t *const *frames, int n_frames, int channels,
                        int w, int h, int stride, asl_detection *out, int max_out, int *n_per_frame,
                        int *n_out);

/* asl_detect_batch_u8 with the per-tag PnP (asl_solve_pnp_batch) fused into the same submission: what the reference
   does per frame with detector.detect + one cv2.solvePnP per tag (tag_detector.py:26,41) in a single call.
   poses[i] belongs to out[i]; K is 9 doubles row-major; dist holds n_dist = 0, 4 or 5 coefficients. */
int asl_detect_batch_pose_u8(asl_detector *det, const uint8_t *const *frames, int n_frames, int channels,
                             int w, int h, int stride, const double *K, const double *dist, int n_dist,
                             double tag_size, asl_detection *out, asl_pose *poses, int max_out,
                             int *n_per_frame, int *n_out);

/* Same, frames already resident in HBM: frame i starts at d_frames + i*frame_pitch.
   If K is non-NULL the per-tag PnP (asl_solve_pnp_batch) runs on the device in the same
   submission and poses[i] belongs to out[i].  Results are written to HOST memory; the call
   returns after the stream has drained. */
int asl_detect_batch_device(asl_detector *det, const void *d_frames, int n_frames, int channels,
                            int w, int h, int stride, size_t frame_pitch, void *stream,
                            const double *K /*9, row-major, or NULL*/, const double *dist, int n_dist,
                            double tag_size, asl_detection *out, asl_pose *poses, int max_out,
                            int *n_per_frame, int *n_out);

/* The same call split in two so that batches can be pipelined: submit enqueues the whole batch (kernels and
   the asynchronous read-back) on `stream` and returns immediately; collect waits for it, de-duplicates and
   writes the results.  One batch may be in flight per detector; use two detectors (two workspaces) and
   alternate them to keep the GPU busy while the host post-processes the previous batch. */
int asl_submit_batch_device(asl_detector *det, const void *d_frames, int n_frames, int channels, int w, int h,
                            int stride, size_t frame_pitch, void *stream, const double *K, const double *dist,
                            int n_dist, double tag_size);
int asl_collect_batch(asl_detector *det, asl_detection *out, asl_pose *poses, int max_out, int *n_per_frame,
                      int *n_out);
/* The same without the last copy: *out / *poses / *n_per_frame point into the detector's own page-locked result buffers
   (*n_out detections in (frame, id) order, *poses = NULL for a batch submitted without a camera); they stay valid until
   the next submit on this detector. */
int asl_collect_batch_view(asl_detector *det, const asl_detection **out, const asl_pose **poses, const uint32_t **n_per_frame,
                           int *n_out);

/* Replaces cv2.solvePnP(ITERATIVE) + cv2.Rodrigues for N tags at once (reference
   tag_detector.py:30-52).  corners: N x 4 x 2 float32 (lb,rb,rt,lt), K row-major 3x3,
   dist: n_dist in {0,4,5} coefficients (k1,k2,p1,p2[,k3]).  All pointers are host memory. */
int asl_solve_pnp_batch(asl_detector *det, const float *corners, const double *K, const double *dist,
                        int n_dist, double tag_size, double *rvec /*N x 3*/, double *tvec /*N x 3*/,
                        double *T /*N x 16*/, uint8_t *ok /*N*/, int N);

/* Pose-graph Gauss-Newton back-end (NOT in the reference: slam_graph.py:72-76 is a stub).
   Unknowns: n_cams camera poses and n_tags tag poses (world<-x 4x4, row-major, updated in
   place); tag `fixed_tag` is held at its input value and defines the world frame.
   Observation k: camera obs_cam[k] saw tag obs_tag[k] with pixel corners obs_corners[k] (4x2).
   Residual: the 8 pixel reprojection errors of the tag's 4 corners.  Runs `iters`
   Levenberg-Marquardt steps on the device; stats[0] = initial cost, stats[1] = final cost,
   stats[2] = iterations accepted. */
int asl_gn_solve(asl_detector *det, int n_cams, int n_tags, int n_obs, const int32_t *obs_cam,
                 const int32_t *obs_tag, const double *obs_corners, const double *K, double tag_size,
                 int fixed_tag, double *cam_T /*n_cams x 16*/, double *tag_T /*n_tags x 16*/, int iters,
                 double *stats /*3*/);

/* ---- after the detector: what the multi-GPU path exchanges and how the graph update consumes it.
   One observation = one detected tag of one frame: what SLAM.get_pose hands to SLAMGraph.add_or_update_node
   (reference slam.py:27-32) plus the corners a later bundle adjustment needs. */
typedef struct {
    int32_t id;       /* -1 = empty slot */
    int32_t flags;    /* bit 0: slot used; bit 1: solvePnP succeeded (the reference updates the graph only then) */
    float corners[8]; /* lb, rb, rt, lt in pixels, float32 as the reference passes them to solvePnP */
    double T[12];     /* rows 0..2 of the camera<-tag 4x4 (row 3 is 0 0 0 1) */
} asl_obs;            /* 136 bytes */

/* Packs the de-duplicated results of the batch last submitted on `det` into d_obs (device memory,
   n_frames x max_tags records, slots ordered by id, empty slots id = -1) on `stream` -- no host round trip, so the
   block can go straight into an all-gather.  Call between asl_submit_batch_device and the next submit. */
int asl_pack_observations_device(asl_detector *det, void *d_obs, int max_tags, void *stream);

/* The data-parallel part of the graph update over gathered records d_obs[world][n_frames][max_tags]:
   d_pose[world*n_frames][16] = SLAM.my_pose() of every frame that (a) sees the world tag `coordinate_id` as its lowest id
   and (b) has no failed PnP -- such a frame's update depends on nothing but the frame (branches A / C1 of
   slam_graph.py:33-49); d_status = 0 for those, 1 for frames that need the sequential update, 2 for frames without
   detections; d_last[id] (caller zeroes it) = 1 + ((frame*world + stream)*max_tags + slot) of the last status-0 frame,
   in (frame, stream) order, that saw tag `id`; d_picks (optional, 2*n_ids asl_obs) receives for every such tag its
   record in that frame and the frame's world-tag record -- all the host needs to finish the update without touching
   the block again. */
int asl_graph_frames_device(asl_detector *det, const void *d_obs, int world, int n_frames, int max_tags, int coordinate_id,
                            double *d_pose, uint8_t *d_status, uint32_t *d_last, int n_ids, void *d_picks, void *stream);

/* Last sightings and picks as above, restricted to the status-0 frames whose position frame*world + stream lies in
   [order_lo, order_hi) (d_status from asl_graph_frames_device; d_last is zeroed here).  The host applies the
   self-contained stretches of a block between two frames that need the sequential update with it
   (aprilslam_amd/dist.py: apply_block). */
int asl_graph_picks_device(asl_detector *det, const void *d_obs, int world, int n_frames, int max_tags, const uint8_t *d_status,
                           unsigned int order_lo, unsigned int order_hi, uint32_t *d_last, int n_ids, void *d_picks, void *stream);

/* ---- before the detector: the image-formation step on the device (reference src/simulation/renderer.py:197-274:
   purple clear colour, one GL_LINEAR-textured quad per tag, BGR read-back).  One plane per visible tag and frame, in
   painter's order (far to near); a plane with tex < 0 ends a frame's list. */
typedef struct {
    double Hi[9];     /* row-major 3x3: pixel centre (x+0.5, y+0.5, 1) -> tag plane (X, Y, w); with lens coefficients it maps
                         UNDISTORTED NORMALISED image coordinates instead */
    int32_t bbox[4];  /* x0, x1, y0, y1: pixels outside [x0,x1) x [y0,y1) cannot hit the tag */
    int32_t tex;      /* index into d_textures */
    int32_t pad;
} asl_render_plane;   /* 96 bytes */

/* Renders n_frames BGR frames of w x h pixels into device memory.  d_planes: n_frames x max_planes asl_render_plane
   (device), far to near, a record with tex < 0 ends a frame's list;
   d_textures: n_tex gray textures of tw x th bytes, rows top to bottom (device); half = half the side of the textured quad
   in scene units.  K (9 doubles) and dist (n_dist = 4 or 5) are host pointers and only needed for a camera with lens
   distortion (the reference's webcam caller, src/detection/video_detection.py:209-296); pass K = NULL for the
   simulator's pinhole. */
int asl_render_frames_device(asl_detector *det, void *d_frames, int n_frames, int w, int h, int stride, size_t frame_pitch,
                             const void *d_planes, int max_planes, const void *d_textures, int tw, int th, double half,
                             const double *K, const double *dist, int n_dist, void *stream);

/* Introspection for the parity tests: copy an intermediate buffer of the LAST batch to host.
   what: 0 = decimated gray (u8, B*sh*sw)     1 = threshold image (u8, B*sh*sw)
         2 = component labels (u32, B*sh*sw)  3 = component sizes by label (u32, B*sh*sw)
         4 = candidate quads (asl_debug_quad, count via *n_items)
         5 = stage counters (int64[18]: frames, sw, sh, clusters, points, quads, detections, ..., tiles of the two dense launches)
         7 = diagnostic: re-run the quad fit of the last batch `bytes` times on the device buffers it left behind; dst receives
             int64[7]: repetitions, quads that differ from the first repetition, and those by size class (any is a race); clusters of
             more than 1024 points are left out: their fit sorts in place, a second run would not see the same input
         8 = diagnostic: the shared-reciprocal division of the line fits (asl_common.h) against the compiler's on 2^29 random
             operand pairs with exponents within +-`bytes` (default 100); dst receives int64[2]: pairs, mismatches
         6 = clusters handed to the quad fit (uint64[3] each: key, points, hash of the sorted point records), ordered by key
   bytes = capacity of dst; *n_items = number of elements written. */
typedef struct {
    double p[4][2]; /* decimated-image pixel coordinates, before the full-resolution rescale */
    uint64_t cluster;
    int32_t frame;
    int32_t reversed_border;
} asl_debug_quad;
int asl_debug_fetch(asl_detector *det, int what, void *dst, size_t bytes, size_t *n_items);

/* Time of each stage of the last batch in milliseconds (HIP events on the detector's stream):
   names[i] / ms[i], i < *n. */
int asl_stage_times(asl_detector *det, const char **names, float *ms, int max_n, int *n);
/* Enable/disable per-stage event timing (adds a few events per batch; off by default). */
int asl_set_profiling(asl_detector *det, int enabled);

/* Diagnostic builds only (-DASL_PHASE_TIMING): summed shader-clock cycles per kernel phase
   (64 counters); all zeros in the shipped library. */
int asl_debug_phase_cycles(asl_detector *det, unsigned long long *out64, int reset);

#ifdef __cplusplus
}
#endif
#endif
