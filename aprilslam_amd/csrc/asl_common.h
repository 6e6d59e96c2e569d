// Shared definitions for the gfx950 kernels of libaprilslam.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define ASL_WAVE 64
#define TILESZ 4
#define MIN_WHITE_BLACK_DIFF 5
#define MIN_COMPONENT 25u
#define MAX_NMAXIMA 10
#define MAX_LINE_FIT_MSE 10.0
#define COS_CRITICAL_RAD 0.984807753012208 /* cos(10 deg) */
#define HASH_EMPTY 0xFFFFFFFFFFFFFFFFull
#define HASH_MAX_PROBE 512
#define NCLASSES 5
// size classes of the quad fit: points <= 128, <= 256, <= 512, <= 1024 (all-LDS), larger (global slab).  The LDS slab of a
// workgroup is 64 bytes x cap, so every halving of the cap doubles the workgroups a CU can hold.
#define CLASS0_CAP 128
#define CLASS1_CAP 256
#define CLASS2_CAP 512
#define CLASS3_CAP 1024

// growable device buffer
template <typename T>
struct DevBuf {
    T *p = nullptr;
    size_t n = 0;
    int ensure(size_t want)
    {
        if (want <= n) return 0;
        if (p) { (void)hipFree(p); p = nullptr; n = 0; }
        hipError_t e = hipMalloc((void **)&p, want * sizeof(T));
        if (e != hipSuccess) { p = nullptr; return -1; }
        n = want;
        return 0;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; n = 0; }
};

// Geometry of one batch, passed by value to every kernel.
struct Geom {
    int w, h;          // full-resolution frame
    int stride;        // bytes between rows of a frame
    int channels;      // 1 = gray, 3 = BGR
    int f;             // integer decimation factor
    int sw, sh;        // decimated size
    int tw, th;        // full 4x4 tiles of the decimated image
    int nframes;
    size_t frame_pitch;  // bytes between frames of the input
    size_t npix;         // sw*sh
};

// A cluster that survived the size pre-filter and will be fitted.
struct ClusterRec {
    uint64_t key;        // (frame << 48) | (hi << 24) | lo
    unsigned int offset; // first point in the point pool
    unsigned int count;
    unsigned int fill;   // scatter cursor
    unsigned int pad;
};

struct QuadRec {
    double p[4][2];  // decimated-image coordinates
    uint64_t key;
    int valid;
    int reversed_border;
};

struct DetRec {
    int32_t id, hamming;
    float margin;
    int32_t frame;
    double center[2];
    double corners[4][2];
    uint64_t key;  // cluster key, for a deterministic order
    double rvec[3], tvec[3], T[16];
    int32_t pose_ok, pad;
};

// Code-book index (k_decode): a word within maxhamming of a code agrees with it on at least one of maxhamming + 1 chunks
// of the bits, so the codes are filed per chunk under the chunk's value (hashed to IDX_BUCKETS when the chunk is wider than
// 12 bits) and a lookup compares the word with the handful of codes in its buckets instead of the whole book.
#define IDX_BUCKET_BITS 12
#define IDX_BUCKETS (1 << IDX_BUCKET_BITS)
#define IDX_MAX_CHUNKS 4
struct FamilyDev {
    int nbits, width_at_border, total_width, reversed_border, ncodes;
    int bit_x[64], bit_y[64];
    const unsigned long long *codes;  // device pointer
    int idx_nch;                                  // chunks (0: no index, the book is searched as a whole)
    int idx_lo[IDX_MAX_CHUNKS], idx_w[IDX_MAX_CHUNKS];  // first bit and width of every chunk
    const unsigned short *idx_start;              // [chunk][IDX_BUCKETS + 1]: first entry of every bucket
    const unsigned long long *idx_entries;        // [chunk][ncodes]: id << 48 | code, by bucket, ids ascending
};
__host__ __device__ __forceinline__ unsigned int idx_bucket(unsigned long long chunk_value, int width)
{
    return width <= IDX_BUCKET_BITS ? (unsigned int)chunk_value : (unsigned int)((chunk_value * 0x9E3779B97F4A7C15ull) >> (64 - IDX_BUCKET_BITS));
}

struct CamDev {
    double fx, fy, cx, cy, k1, k2, p1, p2, k3;
    double half;  // tag_size/2 rounded through float32 as the reference does
    int both_minima;  // 0 (default): the minimum the homography start leads to, like cv2.solvePnP(ITERATIVE); 1: the better of the two planar poses
    int pad;
};

// counters living in device memory (zeroed per batch).  The ones that whole launches add to sit 128 bytes apart: atomics on
// one cache line are served one at a time (~10 ns each, measured in k_graph_frames), and k_cluster_filter's seven block
// totals per workgroup used to share two lines -- most of that kernel's time.
enum {
    CNT_NCLUSTERS = 0,  // surviving clusters
    CNT_OVERFLOW_HASH = 1,
    CNT_OVERFLOW_POINTS = 2,
    CNT_OVERFLOW_CLUSTERS = 3,
    CNT_OVERFLOW_DETS = 4,
    CNT_UF_GUARD = 5,     // a union-find loop ran into its iteration guard (never seen; fails the batch loudly)
    CNT_DEDUP_LIMIT = 6,  // frames with more detections than the de-duplication sorts (fails the batch loudly)
    CNT_NPOINTS = 16,     // points reserved for surviving clusters
    CNT_CLASS0 = 32,      // clusters per size class (lists consumed by the fit kernels): CNT_CLASS(c)
    CNT_NDETS = 112,      // detections appended
    CNT_NQUADS = 128,
    CNT_NKEEP = 144,      // detections that survive the de-duplication (what the caller receives)
    CNT_DENSE_TILES = 160,  // tiles of the cluster pass with more points than a workgroup parks in its small LDS buffer
    CNT_DENSE_SEG = 161,    // tiles of the labelling pass with more runs or links than its LDS tables hold (k_seg_tile_dense takes them)
    CNT__N = 176
};
#define CNT_CLASS(c) (CNT_CLASS0 + 16 * (c))

// IEEE double division with the denominator's part done once.  The compiler expands a / d to v_div_scale x2, v_rcp_f64,
// two Newton steps on the reciprocal (4 fma), q = a*r, rem = fma(-d, q, a), v_div_fmas (= fma(rem, r, q) unless
// v_div_scale rescaled an operand), v_div_fixup (passes q through unless an operand or the quotient is zero, subnormal,
// infinite or NaN): 11 instructions, 6 of which depend on d alone.  Recip keeps those; div_by() is the other three, the
// same operations on the same values, so the quotient is the correctly rounded one, bit for bit -- PROVIDED neither
// rescaling nor fix-up would have acted: d normal and finite with |d| in [2^-1000, 2^1000], a zero or of like magnitude,
// |a/d| in [2^-1000, 2^1000] (v_div_scale acts on exponent differences >= 768 and on operands below 2^-970; a = +0
// gives +0 either way; a = -0 would give +0 here and -0 there: callers pass sums and differences, where x - x = +0).
// The moments, covariances and pixel coordinates of the quad path are within 2^+-80.  tools/div_test.hip compares the two
// forms on the device.
struct Recip {
    double d, r;
};
__device__ __forceinline__ Recip recip_of(double d)
{
    double r = __builtin_amdgcn_rcp(d);
    double e = __builtin_fma(-d, r, 1.0);
    r = __builtin_fma(r, e, r);
    e = __builtin_fma(-d, r, 1.0);
    r = __builtin_fma(r, e, r);
    return Recip{d, r};
}
__device__ __forceinline__ double div_by(double a, const Recip &k)
{
    const double q = a * k.r;
    const double rem = __builtin_fma(-k.d, q, a);
    return __builtin_fma(rem, k.r, q);
}

// A batch whose work buffers overflowed is re-run by the host after growing them; until then its cluster
// lists and quad records have holes, so every consumer kernel backs out first thing.
__device__ __forceinline__ bool batch_poisoned(const long long *counters)
{
    return (counters[CNT_OVERFLOW_HASH] | counters[CNT_OVERFLOW_POINTS] | counters[CNT_OVERFLOW_CLUSTERS] | counters[CNT_OVERFLOW_DETS]) != 0;
}

// Diagnostic build only (-DASL_PHASE_TIMING): per-phase shader-clock sums, one stamp per block.
// The shipped library compiles these to nothing.
#define PHASE_SPREAD 512  /* copies of every counter (one address sustains ~90 atomics/us); the host adds them up */
__device__ unsigned long long g_phase_cycles[64 * PHASE_SPREAD];
#ifdef ASL_PHASE_TIMING
// stamps are summed in LDS by thread 0 and leave with one atomic per phase when the workgroup ends: an atomic per
// stamp would queue behind the other workgroups' (one address sustains ~90 atomics/us) and the next global load of
// the wave would wait for it, charging the queueing to whichever phase touches memory first
#define PHASE_DECL()                                                                  \
    __shared__ unsigned long long ph_s__[16];                                         \
    long long ph_t__ = 0;                                                             \
    if (threadIdx.x == 0 && threadIdx.y == 0)                                         \
        for (int ph_i__ = 0; ph_i__ < 16; ph_i__++) ph_s__[ph_i__] = 0
#define PHASE_INIT() ph_t__ = clock64()
#define PHASE(k)                                                                      \
    do {                                                                              \
        if (threadIdx.x == 0 && threadIdx.y == 0) {                                   \
            long long now__ = clock64();                                              \
            ph_s__[(k) & 15] += (unsigned long long)(now__ - ph_t__);                 \
            ph_t__ = now__;                                                           \
        }                                                                             \
    } while (0)
#define PHASE_COUNT(k, v) do { if (threadIdx.x == 0 && threadIdx.y == 0) ph_s__[(k) & 15] += (unsigned long long)(v); } while (0)
#define PHASE_FLUSH(base)                                                             \
    do {                                                                              \
        if (threadIdx.x == 0 && threadIdx.y == 0)                                     \
            for (int ph_i__ = 0; ph_i__ < 16; ph_i__++)                               \
                if (ph_s__[ph_i__]) atomicAdd(&g_phase_cycles[((base) + ph_i__) * PHASE_SPREAD + ((blockIdx.x + 7u * blockIdx.y + 13u * blockIdx.z) & (PHASE_SPREAD - 1))], ph_s__[ph_i__]); \
    } while (0)
#else
#define PHASE_DECL() do {} while (0)
#define PHASE_INIT() do {} while (0)
#define PHASE(k) do {} while (0)
#define PHASE_COUNT(k, v) do {} while (0)
#define PHASE_FLUSH(base) do {} while (0)
#endif

// pixel_fetch + pixel_gray = gray_at split in two, so that callers can issue many fetches before converting.
// Branch-free on purpose (these sit in the innermost loops of the per-quad kernels): one possibly unaligned 4-byte
// fetch covers B, G, R (the 4th byte is the next pixel's B); for the last pixel of a row the fetch starts one byte
// earlier and is shifted, so it never leaves the frame.  Offsets are 32-bit: a frame is at most 16384 x 16384 x 3 bytes.
// CH (1 = gray, 3 = BGR) is a template parameter of the kernels on this path, so no branch on it splits their loops.
template <int CH>
__device__ __forceinline__ unsigned int pixel_fetch(const uint8_t *frame, const Geom &g, int x, int y, unsigned int *shift)
{
    // returns the raw fetch; *shift (0 or 8) is applied by pixel_gray, so nothing here depends on the loaded value
    // and a caller can issue many fetches back to back before the first conversion waits for memory
    if (CH == 1) { *shift = 0; return frame[(unsigned int)y * (unsigned int)g.stride + (unsigned int)x]; }
    const unsigned int sh = (x + 1 < g.w) ? 0u : 1u;
    unsigned int u;
    __builtin_memcpy(&u, frame + ((unsigned int)y * (unsigned int)g.stride + 3u * (unsigned int)x - sh), 4);
    *shift = 8u * sh;
    return u;
}

// cv2 BGR2GRAY fixed point, (3735 B + 19235 G + 9798 R + 16384) >> 15, of the three low bytes of u.  The 15-bit weights
// are split into a high and a low byte (3735 = 14 * 256 + 151, 19235 = 75 * 256 + 35, 9798 = 38 * 256 + 70) so that two
// v_dot4_u32_u8 do the three byte extractions and multiply-adds: the same integer, in four instructions instead of eight
// (the probes of the edge refinement and of the decoder convert ~2.5 K pixels per quad).
__device__ __forceinline__ int bgr_gray(unsigned int u)
{
    const unsigned int hi = __builtin_amdgcn_udot4(u, 0x00264B0Eu, 0u, false), lo = __builtin_amdgcn_udot4(u, 0x00462397u, 16384u, false);
    return (int)(((hi << 8) + lo) >> 15);
}

template <int CH>
__device__ __forceinline__ int pixel_gray(unsigned int u, unsigned int shift)
{
    if (CH == 1) return (int)u;
    return bgr_gray(u >> shift);
}

template <int CH>
__device__ __forceinline__ int gray_at(const uint8_t *frame, const Geom &g, int x, int y)
{
    unsigned int sh;
    unsigned int u = pixel_fetch<CH>(frame, g, x, y, &sh);
    return pixel_gray<CH>(u, sh);
}
