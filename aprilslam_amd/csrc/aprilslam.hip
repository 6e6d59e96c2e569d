// libaprilslam.so -- host side of the C ABI declared in include/aprilslam.h.
// gfx950 only.  One asl_detector owns a device workspace sized for its largest batch and
// submits the whole detector (+ optional PnP) as one chain of launches on one HIP stream.
#include "../../include/aprilslam.h"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "asl_common.h"
#include "tag_standard41h12.inc"

#include "k_threshold.inc"
#include "k_cc.inc"
#include "k_cluster.inc"
#include "k_seg.inc"
#include "k_xchg.inc"
#include "k_quad.inc"
#include "k_decode.inc"
#include "k_pnp.inc"
#include "k_dedup.inc"
#include "k_graph.inc"
#include "k_render.inc"
#include "k_gn.inc"

static thread_local std::string g_err;

// Optional ROCTX ranges around the stage groups, for `rocprofv3 --marker-trace`: ASL_ROCTX=1 in the
// environment loads the marker library at first use; without it these are two predictable branches.
#include <dlfcn.h>
static struct Markers {
    int (*push)(const char *) = nullptr;
    int (*pop)() = nullptr;
    bool tried = false;
    void load()
    {
        tried = true;
        const char *e = getenv("ASL_ROCTX");
        if (!e || !*e || *e == '0') return;
        void *h = dlopen("librocprofiler-sdk-roctx.so", RTLD_NOW | RTLD_GLOBAL);
        if (!h) h = dlopen("libroctx64.so", RTLD_NOW | RTLD_GLOBAL);
        if (!h) return;
        push = reinterpret_cast<int (*)(const char *)>(dlsym(h, "roctxRangePushA"));
        pop = reinterpret_cast<int (*)()>(dlsym(h, "roctxRangePop"));
        if (!push || !pop) { push = nullptr; pop = nullptr; }
    }
} g_markers;
static inline void range_push(const char *name) { if (!g_markers.tried) g_markers.load(); if (g_markers.push) g_markers.push(name); }
static inline void range_pop() { if (g_markers.pop) g_markers.pop(); }

static int fail(int code, const char *fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}

#define HIPCHK(expr)                                                                                   \
    do {                                                                                               \
        hipError_t e__ = (expr);                                                                       \
        if (e__ != hipSuccess) return fail(ASL_EDEVICE, "%s failed: %s", #expr, hipGetErrorString(e__)); \
    } while (0)

#define MAX_STAGES 24

struct asl_detector {
    int device = 0;
    int maxhamming = 1;
    int decimate = 2;
    int refine = 1;
    int pnp_both_minima = 0;
    FamilyDev fam;
    unsigned long long *d_codes = nullptr;
    DevBuf<unsigned short> idx_start;        // code-book index of the family (FamilyDev), rebuilt when the id limit changes
    DevBuf<unsigned long long> idx_entries;

    // capacities (grow on overflow)
    unsigned int hash_slots_per_frame = 1024;
    unsigned int clusters_per_frame = 2048;
    unsigned int dets_per_frame = 256;
    double points_per_pixel = 0.5;

    // workspace
    DevBuf<uint8_t> in, dgray, tmin, tmax, tcut;
    DevBuf<uint8_t> dbg_thresh;         // asl_debug_fetch only: the threshold image as bytes
    DevBuf<unsigned int> dbg_labels;    // asl_debug_fetch only: per-pixel labels
    DevBuf<unsigned int> parent, sizes;
    DevBuf<unsigned long long> hkeys, points, rootmask, wmask, bmask;
    DevBuf<unsigned int> hcounts, class_lists, stage_pos, frame_cursor, dense_tiles, dense_seg, quad_list;
    DevBuf<unsigned long long> stage_rec;
    DevBuf<uint4> seg_edges;  // per labelling tile: its first / last pixel column, a bit per row and colour
    unsigned int stage_cap = 0;  // staged points per frame
    DevBuf<unsigned long long> slot_cluster;  // per hash slot: offset | count << 32 of its cluster's segment
    DevBuf<ClusterRec> clusters;
    DevBuf<QuadRec> quads;
    DevBuf<double> scratch, quadH, wtab, side_mom;  // side_mom: 4 x 6 moments per cluster, k_fit_quads -> k_quad_finish
    DevBuf<DetRec> dets;
    // S8 on the device: per-frame index lists, counts and offsets, and the results in the ABI's layout
    DevBuf<unsigned int> frame_ndets, frame_idx, frame_nkeep, frame_off;
    DevBuf<DetOut> out_det;
    DevBuf<PoseOut> out_pose;
    DevBuf<long long> counters;
    DevBuf<float> pnp_corners;
    DevBuf<double> pnp_out;
    DevBuf<uint8_t> pnp_ok;
    GnWorkspace gn;
    hipStream_t copy_stream = nullptr, host_stream = nullptr;  // host frames: transfers and the chunks' kernels (detect_host_frames)
    std::vector<hipEvent_t> copy_done;
    hipStream_t aux_stream = nullptr;  // highest priority, for the small latency-bound jobs next to a running batch (pose-graph LM)

    // sizes used by the last batch
    Geom last{};
    unsigned int nslots = 0, max_clusters = 0, max_points = 0, max_dets = 0;
    long long last_counters[CNT__N] = {0};
    // batch in flight (asl_submit_batch_device .. asl_collect_batch)
    bool pending = false;
    const uint8_t *p_frames = nullptr;
    Geom p_geom{};
    hipStream_t p_stream = nullptr;
    bool p_has_cam = false;
    CamDev p_cam{};
    size_t prefetched = 0, nd_guess = 0;
    std::chrono::steady_clock::time_point t_submit, t_enqueued;
    long long *pinned_counters = nullptr;  // D2H target that does not force a blocking staging copy
    DetOut *host_det = nullptr;    // pinned staging of the results
    PoseOut *host_pose = nullptr;
    size_t host_cap = 0;
    unsigned int *host_nkeep = nullptr;  // pinned, per frame
    size_t host_nkeep_cap = 0;

    // profiling
    int profiling = 0;
    hipEvent_t ev[MAX_STAGES + 1] = {nullptr};
    int nev = 0;
    const char *stage_names[MAX_STAGES] = {nullptr};
    float stage_ms[MAX_STAGES] = {0};
    int nstages = 0;
    float host_ms[4] = {0, 0, 0, 0};  // enqueue, wait, copy, post-process of the last batch
};

static const char *kVersion = "aprilslam 0.1 gfx950 (HIP, tagStandard41h12)";

extern "C" const char *asl_last_error(void) { return g_err.c_str(); }
extern "C" const char *asl_version(void) { return kVersion; }

// Code-book index for k_decode (asl_common.h: FamilyDev): maxhamming + 1 chunks of the code bits; a family wider than 48 bits
// or with more than 65535 ids keeps idx_nch = 0 and is searched as a whole.
static int build_code_index(asl_detector *d)
{
    FamilyDev &f = d->fam;
    f.idx_nch = 0; f.idx_start = nullptr; f.idx_entries = nullptr;
    const int nch = d->maxhamming + 1;
    if (f.nbits > 48 || f.ncodes > 65535 || nch > IDX_MAX_CHUNKS) return 0;
    if (getenv("ASL_NO_CODE_INDEX")) return 0;  // tests: the search of the whole book must agree with the index
    std::vector<unsigned short> start((size_t)nch * (IDX_BUCKETS + 1), 0);
    std::vector<unsigned long long> entries((size_t)nch * f.ncodes);
    int lo = 0;
    for (int c = 0; c < nch; c++) {
        const int w = f.nbits / nch + (c < f.nbits % nch ? 1 : 0);
        f.idx_lo[c] = lo; f.idx_w[c] = w;
        std::vector<unsigned int> bucket(f.ncodes);
        unsigned short *st = start.data() + (size_t)c * (IDX_BUCKETS + 1);
        for (int i = 0; i < f.ncodes; i++) {
            bucket[i] = idx_bucket((kTag41h12Codes[i] >> lo) & ((1ull << w) - 1ull), w);
            st[bucket[i] + 1]++;
        }
        for (int b = 0; b < IDX_BUCKETS; b++) st[b + 1] = (unsigned short)(st[b + 1] + st[b]);
        std::vector<unsigned short> fill(st, st + IDX_BUCKETS);
        for (int i = 0; i < f.ncodes; i++)  // ids ascending inside a bucket
            entries[(size_t)c * f.ncodes + fill[bucket[i]]++] = ((unsigned long long)i << 48) | kTag41h12Codes[i];
        lo += w;
    }
    for (int c = nch; c < IDX_MAX_CHUNKS; c++) { f.idx_lo[c] = 0; f.idx_w[c] = 1; }
    if (d->idx_start.ensure(start.size()) || d->idx_entries.ensure((size_t)IDX_MAX_CHUNKS * kTag41h12NCodes)) return -1;
    if (hipMemcpy(d->idx_start.p, start.data(), start.size() * sizeof(unsigned short), hipMemcpyHostToDevice) != hipSuccess) return -1;
    if (hipMemcpy(d->idx_entries.p, entries.data(), entries.size() * sizeof(unsigned long long), hipMemcpyHostToDevice) != hipSuccess) return -1;
    f.idx_start = d->idx_start.p; f.idx_entries = d->idx_entries.p; f.idx_nch = nch;
    return 0;
}

extern "C" int asl_detector_create(const char *family, int nthreads, int maxhamming, float decimate, float blur,
                                   int refine_edges, int device, asl_detector **out)
{
    (void)nthreads;  // host threading is meaningless here: the parallelism is the GPU grid
    if (!out) return fail(ASL_EINVAL, "out is NULL");
    *out = nullptr;
    if (!family || strcmp(family, "tagStandard41h12") != 0)
        return fail(ASL_EINVAL, "unknown tag family '%s' (supported: tagStandard41h12)", family ? family : "(null)");
    if (!(decimate >= 1.0f) || decimate != std::floor(decimate) || decimate > 8.0f)
        return fail(ASL_EINVAL, "decimate must be an integer value in [1, 8] (got %g)", (double)decimate);
    if (blur != 0.0f) return fail(ASL_EINVAL, "blur (quad_sigma) != 0 is not supported (the reference never sets it)");
    if (maxhamming < 0 || maxhamming > 3) return fail(ASL_EINVAL, "maxhamming must be in [0, 3]");
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0) return fail(ASL_EDEVICE, "no HIP device available (%s)", hipGetErrorString(e));
    if (device < 0 || device >= ndev) return fail(ASL_EINVAL, "device %d out of range (have %d)", device, ndev);
    HIPCHK(hipSetDevice(device));
    asl_detector *d = new asl_detector();
    d->device = device;
    d->maxhamming = maxhamming;
    d->decimate = (int)decimate;
    d->refine = refine_edges ? 1 : 0;
    memset(&d->fam, 0, sizeof d->fam);
    d->fam.nbits = kTag41h12NBits;
    d->fam.width_at_border = kTag41h12WidthAtBorder;
    d->fam.total_width = kTag41h12TotalWidth;
    d->fam.reversed_border = 1;
    d->fam.ncodes = kTag41h12PinnedIds;  // ids the reference's own tag images pin; asl_detector_set_id_limit opens the rest
    for (int i = 0; i < kTag41h12NBits; i++) { d->fam.bit_x[i] = kTag41h12BitX[i]; d->fam.bit_y[i] = kTag41h12BitY[i]; }
    if (hipMalloc((void **)&d->d_codes, sizeof(unsigned long long) * kTag41h12NCodes) != hipSuccess) {
        delete d;
        return fail(ASL_ENOMEM, "hipMalloc(code book) failed");
    }
    if (hipMemcpy(d->d_codes, kTag41h12Codes, sizeof(unsigned long long) * kTag41h12NCodes, hipMemcpyHostToDevice) != hipSuccess) {
        (void)hipFree(d->d_codes);
        delete d;
        return fail(ASL_EDEVICE, "hipMemcpy(code book) failed");
    }
    d->fam.codes = d->d_codes;
    if (build_code_index(d)) {
        d->idx_start.release(); d->idx_entries.release();
        (void)hipFree(d->d_codes);
        delete d;
        return fail(ASL_ENOMEM, "code-book index allocation failed");
    }
    if (d->wtab.ensure(WEIGHT_TABLE_N)) {
        (void)hipFree(d->d_codes);
        delete d;
        return fail(ASL_ENOMEM, "hipMalloc(weight table) failed");
    }
    hipLaunchKernelGGL(k_weight_table, dim3((WEIGHT_TABLE_N + 255) / 256), dim3(256), 0, 0, d->wtab.p);
    if (hipDeviceSynchronize() != hipSuccess) {
        d->wtab.release();
        (void)hipFree(d->d_codes);
        delete d;
        return fail(ASL_EDEVICE, "weight table kernel failed");
    }
    // class-3 quad fit uses 64 KB of dynamic LDS on top of a few hundred static bytes
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_fit_quads<256, true, CLASS3_CAP / 256>), hipFuncAttributeMaxDynamicSharedMemorySize, QUAD_LDS_BYTES(CLASS3_CAP));
    *out = d;
    return ASL_OK;
}

extern "C" void asl_detector_destroy(asl_detector *d)
{
    if (!d) return;
    (void)hipSetDevice(d->device);
    if (d->pending) (void)hipStreamSynchronize(d->p_stream);  // a batch still in flight reads and writes the workspace
    d->rootmask.release(); d->quad_list.release(); d->dense_tiles.release(); d->dense_seg.release(); d->seg_edges.release(); d->wmask.release(); d->bmask.release();
    d->dbg_thresh.release(); d->dbg_labels.release();
    d->frame_ndets.release(); d->frame_idx.release(); d->frame_nkeep.release(); d->frame_off.release(); d->out_det.release(); d->out_pose.release();
    if (d->host_pose) (void)hipHostFree(d->host_pose);
    if (d->host_nkeep) (void)hipHostFree(d->host_nkeep);
    d->in.release(); d->dgray.release(); d->tmin.release(); d->tmax.release(); d->tcut.release();
    d->parent.release(); d->sizes.release(); d->hkeys.release(); d->points.release(); d->hcounts.release(); d->class_lists.release(); d->stage_pos.release(); d->frame_cursor.release(); d->stage_rec.release();
    d->slot_cluster.release(); d->clusters.release(); d->quads.release(); d->scratch.release(); d->side_mom.release(); d->quadH.release(); d->wtab.release(); d->dets.release();
    d->counters.release(); d->pnp_corners.release(); d->pnp_out.release(); d->pnp_ok.release();
    d->gn.release();
    if (d->aux_stream) (void)hipStreamDestroy(d->aux_stream);
    if (d->copy_stream) (void)hipStreamDestroy(d->copy_stream);
    if (d->host_stream) (void)hipStreamDestroy(d->host_stream);
    for (hipEvent_t e : d->copy_done) (void)hipEventDestroy(e);
    if (d->d_codes) (void)hipFree(d->d_codes);
    d->idx_start.release(); d->idx_entries.release();
    if (d->host_det) (void)hipHostFree(d->host_det);
    if (d->pinned_counters) (void)hipHostFree(d->pinned_counters);
    for (int i = 0; i <= MAX_STAGES; i++) if (d->ev[i]) (void)hipEventDestroy(d->ev[i]);
    delete d;
}

extern "C" int asl_detector_set_id_limit(asl_detector *d, int n_ids)
{
    if (!d) return fail(ASL_EINVAL, "detector is NULL");
    if (n_ids > kTag41h12NCodes) return fail(ASL_EINVAL, "the code table holds %d ids (asked for %d)", kTag41h12NCodes, n_ids);
    if (d->pending) return fail(ASL_EINVAL, "a batch is in flight on this detector");
    d->fam.ncodes = n_ids <= 0 ? kTag41h12NCodes : n_ids;
    HIPCHK(hipSetDevice(d->device));
    if (build_code_index(d)) return fail(ASL_ENOMEM, "code-book index allocation failed");
    return ASL_OK;
}

extern "C" int asl_detector_set_pnp_both_minima(asl_detector *d, int enabled)
{
    if (!d) return fail(ASL_EINVAL, "detector is NULL");
    d->pnp_both_minima = enabled ? 1 : 0;
    return ASL_OK;
}

extern "C" int asl_set_profiling(asl_detector *d, int enabled)
{
    if (!d) return fail(ASL_EINVAL, "detector is NULL");
    d->profiling = enabled ? 1 : 0;
    if (enabled)
        for (int i = 0; i <= MAX_STAGES; i++)
            if (!d->ev[i]) HIPCHK(hipEventCreate(&d->ev[i]));
    return ASL_OK;
}

extern "C" int asl_stage_times(asl_detector *d, const char **names, float *ms, int max_n, int *n)
{
    if (!d || !n) return fail(ASL_EINVAL, "NULL argument");
    int k = std::min(max_n, d->nstages);
    for (int i = 0; i < k; i++) { if (names) names[i] = d->stage_names[i]; if (ms) ms[i] = d->stage_ms[i]; }
    static const char *hn[4] = {"host_enqueue", "host_wait", "host_copy", "host_post"};
    for (int i = 0; i < 4 && k < max_n; i++, k++) { if (names) names[k] = hn[i]; if (ms) ms[k] = d->host_ms[i]; }
    *n = k;
    return ASL_OK;
}

// Tags per 64-lane wave of the PnP kernels (k_pnp.inc: a quad of lanes per tag, so at most 16): fewer when the launch
// cannot fill the chip anyway, so that a wave is not held up by the slowest of 16 tags.  `expected` = tags in the launch.
static int pnp_tpw(size_t expected)
{
    const char *e = getenv("ASL_PNP_TPW");  // tuning override
    if (e) { int x = atoi(e); return x < 1 ? 1 : (x > 16 ? 16 : x); }
    int v = 1;  // about a wave per SIMD (1024 of them) before the waves fill up: 20 K tags ran 0.153 / 0.193 / 0.256 ms at 16 / 8 / 4 tags per wave
    while (v < 16 && (size_t)v * 2 * 1024 <= expected) v <<= 1;
    return v;
}

static unsigned int next_pow2(unsigned long long v)
{
    unsigned long long p = 1;
    while (p < v) p <<= 1;
    return (unsigned int)p;
}

static int make_geom(asl_detector *d, int n_frames, int channels, int w, int h, int stride, size_t frame_pitch, Geom *g)
{
    if (n_frames <= 0 || n_frames > 65535) return fail(ASL_EINVAL, "n_frames must be in [1, 65535] (got %d)", n_frames);
    if (channels != 1 && channels != 3) return fail(ASL_EINVAL, "channels must be 1 (gray) or 3 (BGR), got %d", channels);
    if (w < 8 || h < 8 || w > 16384 || h > 16384) return fail(ASL_EINVAL, "unsupported image size %dx%d", w, h);
    if (stride < w * channels) return fail(ASL_EINVAL, "stride %d smaller than a row (%d bytes)", stride, w * channels);
    g->w = w; g->h = h; g->stride = stride; g->channels = channels; g->f = d->decimate;
    g->sw = 1 + (w - 1) / g->f; g->sh = 1 + (h - 1) / g->f;
    g->tw = g->sw / TILESZ; g->th = g->sh / TILESZ;
    g->nframes = n_frames; g->frame_pitch = frame_pitch;
    g->npix = (size_t)g->sw * g->sh;
    if (g->npix >= (1u << 24)) return fail(ASL_EINVAL, "decimated frame has %zu pixels; the cluster key holds 2^24", g->npix);
    return ASL_OK;
}

static int seg_nwx(const Geom &g) { return (g.sw + 63) / 64; }                       // 64-pixel words per row
static int seg_point_tiles_y(const Geom &g) { return std::max(1, (g.sh - 1 + SEG_PH - 1) / SEG_PH); }  // k_seg_points tiles: rows 63k .. 63k+62 emit
static size_t count_tiles(const Geom &g) { return (size_t)seg_nwx(g) * (size_t)seg_point_tiles_y(g); }

static int ensure_workspace(asl_detector *d, const Geom &g)
{
    size_t B = (size_t)g.nframes;
    size_t total = B * g.npix;
    d->nslots = next_pow2(std::max<unsigned long long>(16384ull, (unsigned long long)B * d->hash_slots_per_frame));
    unsigned long long mc = (unsigned long long)B * d->clusters_per_frame;
    d->max_clusters = (unsigned int)std::min<unsigned long long>(mc, 0x7FFFFFFFull);
    unsigned long long mp = (unsigned long long)((double)total * d->points_per_pixel) + 65536ull;
    d->max_points = (unsigned int)std::min<unsigned long long>(mp, 0xFFFFFFF0ull);
    d->max_dets = (unsigned int)std::min<unsigned long long>((unsigned long long)B * d->dets_per_frame, 0x7FFFFFFFull);
    int bad = 0;
    bad |= d->dgray.ensure(total);
    bad |= d->tmin.ensure(B * (size_t)std::max(1, g.tw * g.th) + 8);  // + 8: k_tile_cut's last 8-byte load
    bad |= d->tmax.ensure(B * (size_t)std::max(1, g.tw * g.th) + 8);
    bad |= d->tcut.ensure(B * (size_t)std::max(1, g.tw * g.th));
    bad |= d->parent.ensure(total);
    bad |= d->sizes.ensure(total);
    bad |= d->rootmask.ensure(B * (size_t)g.sh * (size_t)seg_nwx(g));
    bad |= d->wmask.ensure(B * (size_t)g.sh * (size_t)seg_nwx(g));
    bad |= d->bmask.ensure(B * (size_t)g.sh * (size_t)seg_nwx(g));
    bad |= d->hkeys.ensure(d->nslots);
    bad |= d->hcounts.ensure(d->nslots);
    bad |= d->class_lists.ensure((size_t)NCLASSES * d->max_clusters);
    bad |= d->slot_cluster.ensure(d->nslots);
    bad |= d->clusters.ensure(d->max_clusters);
    bad |= d->quads.ensure(d->max_clusters);
    bad |= d->quad_list.ensure(d->max_clusters);
    bad |= d->quadH.ensure((size_t)10 * d->max_clusters);
    bad |= d->side_mom.ensure((size_t)24 * d->max_clusters);
    bad |= d->points.ensure(d->max_points);
    d->stage_cap = (unsigned int)((double)g.npix * d->points_per_pixel) + 1024u;
    bad |= d->stage_rec.ensure((size_t)B * d->stage_cap);
    bad |= d->stage_pos.ensure((size_t)B * d->stage_cap);
    bad |= d->frame_cursor.ensure(B);
    bad |= d->dense_tiles.ensure(B * count_tiles(g));
    bad |= d->dense_seg.ensure(B * (size_t)seg_nwx(g) * (size_t)((g.sh + SEG_TH - 1) / SEG_TH));
    bad |= d->seg_edges.ensure(B * (size_t)seg_nwx(g) * (size_t)((g.sh + SEG_TH - 1) / SEG_TH));
    bad |= d->scratch.ensure((size_t)d->max_points * 8);
    bad |= d->dets.ensure(d->max_dets);
    bad |= d->frame_ndets.ensure(B);
    bad |= d->frame_nkeep.ensure(B);
    bad |= d->frame_off.ensure(B);
    bad |= d->frame_idx.ensure(B * (size_t)d->dets_per_frame);
    bad |= d->out_det.ensure(d->max_dets);
    bad |= d->out_pose.ensure(d->max_dets);
    bad |= d->counters.ensure(CNT__N);
    if (bad) return fail(ASL_ENOMEM, "device workspace allocation failed (%zu frames of %dx%d)", B, g.sw, g.sh);
    return ASL_OK;
}

#define STAGE(name)                                                        \
    do {                                                                   \
        if (d->profiling && d->nev < MAX_STAGES) {                         \
            d->stage_names[d->nev] = name;                                 \
            HIPCHK(hipEventRecord(d->ev[d->nev], st));                     \
            d->nev++;                                                      \
        }                                                                  \
    } while (0)

// Quad fit of one size class (k_quad.inc): grid-stride kernels over the class's device-side cluster list -- many more
// workgroups than fit on the chip, so the heavy-tailed per-cluster costs balance out (workgroups without work leave at once).
static void launch_fit_class(asl_detector *d, const Geom &g, int cls, unsigned int B, hipStream_t st)
{
    const int want_rev = d->fam.reversed_border ? 1 : 0, want_norm = d->fam.reversed_border ? 0 : 1;
    int tag_width = d->fam.width_at_border / g.f;
    if (tag_width < 3) tag_width = 3;
    const unsigned int qgrid = std::min<unsigned int>(d->max_clusters, std::max<unsigned int>(16384u, 32u * B));
    const unsigned int q2grid = std::min<unsigned int>(d->max_clusters, 2048u);
    const unsigned int *list = d->class_lists.p + (size_t)cls * d->max_clusters;
    switch (cls) {
    case 0:
        hipLaunchKernelGGL((k_fit_quads<64, true, CLASS0_CAP / 64>), dim3(qgrid), dim3(64), QUAD_LDS_BYTES(CLASS0_CAP), st, d->clusters.p, list, d->counters.p, 0,
                           d->max_clusters, CLASS0_CAP, d->points.p, d->dgray.p, g, tag_width, want_rev, want_norm, d->scratch.p, d->quads.p, d->wtab.p, d->side_mom.p);
        break;
    case 1:  // two wavefronts per cluster: the 16 KB slab limits a CU to 7 workgroups, so wider workgroups keep more waves in flight
        hipLaunchKernelGGL((k_fit_quads<128, true, CLASS1_CAP / 128>), dim3(qgrid), dim3(128), QUAD_LDS_BYTES(CLASS1_CAP), st, d->clusters.p, list, d->counters.p, 1,
                           d->max_clusters, CLASS1_CAP, d->points.p, d->dgray.p, g, tag_width, want_rev, want_norm, d->scratch.p, d->quads.p, d->wtab.p, d->side_mom.p);
        break;
    case 2:
        hipLaunchKernelGGL((k_fit_quads<256, true, CLASS2_CAP / 256>), dim3(q2grid), dim3(256), QUAD_LDS_BYTES(CLASS2_CAP), st, d->clusters.p, list, d->counters.p, 2,
                           d->max_clusters, CLASS2_CAP, d->points.p, d->dgray.p, g, tag_width, want_rev, want_norm, d->scratch.p, d->quads.p, d->wtab.p, d->side_mom.p);
        break;
    case 3:
        hipLaunchKernelGGL((k_fit_quads<256, true, CLASS3_CAP / 256>), dim3(q2grid), dim3(256), QUAD_LDS_BYTES(CLASS3_CAP), st, d->clusters.p, list, d->counters.p, 3,
                           d->max_clusters, CLASS3_CAP, d->points.p, d->dgray.p, g, tag_width, want_rev, want_norm, d->scratch.p, d->quads.p, d->wtab.p, d->side_mom.p);
        break;
    default:
        hipLaunchKernelGGL((k_fit_quads<256, false, 0>), dim3(q2grid), dim3(256), 0, st, d->clusters.p, list, d->counters.p, 4, d->max_clusters, 0, d->points.p,
                           d->dgray.p, g, tag_width, want_rev, want_norm, d->scratch.p, d->quads.p, d->wtab.p, d->side_mom.p);
        break;
    }
}

static void launch_quad_finish(asl_detector *d, const Geom &g, hipStream_t st)
{
    int tag_width = d->fam.width_at_border / g.f;
    if (tag_width < 3) tag_width = 3;
    hipLaunchKernelGGL(k_quad_finish, dim3(std::min<unsigned int>((d->max_clusters + 63) / 64, 4096u)), dim3(256), 0, st, d->quads.p, d->side_mom.p, d->counters.p,
                       d->max_clusters, tag_width);
}

// enqueue the whole detector for frames resident at d_frames; no host sync
static int enqueue_detect(asl_detector *d, const uint8_t *d_frames, const Geom &g, hipStream_t st, const CamDev *cam)
{
    dim3 blk(64, 4, 1);
    int thx = (g.sh + TILESZ - 1) / TILESZ;  // generic decimation kernel: tile rows
    unsigned int B = (unsigned int)g.nframes;
    d->nev = 0;
    range_push("S0-S4 gray, decimate, threshold, segmentation, clusters");
    STAGE("k_hash_clear");
    hipLaunchKernelGGL(k_hash_clear, dim3((std::max<unsigned int>(d->nslots, std::max<unsigned int>(B, CNT__N)) + 255) / 256), dim3(256), 0, st, d->hkeys.p,
                       d->hcounts.p, d->nslots, d->counters.p, d->frame_cursor.p, d->frame_ndets.p, B);

    STAGE("k_decimate_minmax");
    if (g.f == 2) {
        const unsigned int ntile = (unsigned int)(g.tw * g.th);
        const unsigned int nrest = (unsigned int)(g.sw - 4 * g.tw) * (unsigned int)g.sh + (unsigned int)(g.sh - 4 * g.th) * (unsigned int)(4 * g.tw);
        if (ntile)
            hipLaunchKernelGGL((g.channels == 1 ? k_decimate2_tiles<1> : k_decimate2_tiles<3>), dim3((ntile + 255) / 256, B), dim3(256), 0, st, d_frames, g,
                               d->dgray.p, d->tmin.p, d->tmax.p);
        if (nrest)
            hipLaunchKernelGGL((g.channels == 1 ? k_decimate_rest<1> : k_decimate_rest<3>), dim3((nrest + 255) / 256, B), dim3(256), 0, st, d_frames, g,
                               d->dgray.p);
    } else
        hipLaunchKernelGGL((g.channels == 1 ? k_decimate_minmax<1> : k_decimate_minmax<3>), dim3((g.sw + 63) / 64, (thx + 3) / 4, B), blk, 0, st, d_frames, g, d->dgray.p, d->tmin.p, d->tmax.p);

    const int nwx = seg_nwx(g), pty = seg_point_tiles_y(g);
    const size_t nwords = (size_t)B * g.sh * nwx;
    STAGE("k_tile_cut");
    if (g.tw > 0 && g.th > 0)
        hipLaunchKernelGGL(k_tile_cut, dim3((((g.tw + 3) / 4) * ((g.th + 3) / 4) + 255) / 256, B), dim3(256), 0, st, d->tmin.p, d->tmax.p, g, d->tcut.p);
    STAGE("k_seg_tile");
    {
        const int ntiles = nwx * ((g.sh + SEG_TH - 1) / SEG_TH);
        hipLaunchKernelGGL(k_seg_tile, dim3((ntiles + SEG_TILE_WAVES - 1) / SEG_TILE_WAVES, 1, B), dim3(64 * SEG_TILE_WAVES), 0, st, d->dgray.p, d->tcut.p, g, nwx, ntiles,
                           d->wmask.p, d->bmask.p, d->parent.p, d->sizes.p, d->rootmask.p, d->seg_edges.p, d->dense_seg.p, d->counters.p);
        // tiles with more runs or links than the common launch's tables hold (none in ordinary frames: the launch finds an empty list)
        hipLaunchKernelGGL(k_seg_tile_dense, dim3(1024), dim3(64), 0, st, g, nwx, ntiles, d->wmask.p, d->bmask.p, d->parent.p, d->sizes.p,
                           d->rootmask.p, d->dense_seg.p, d->counters.p);
    }
    STAGE("k_seg_border");
    {
        const size_t nseams = nwx > 1 ? (size_t)B * g.sh * (nwx - 1) : 0;
        const unsigned int ncol_blocks = (unsigned int)((nseams + 63) / 64);
        const int nrb = (g.sh - 1) / SEG_TH;
        const unsigned int nrow_blocks = (unsigned int)nwx * (unsigned int)nrb * (unsigned int)B;
        if (ncol_blocks + nrow_blocks > 0)
            hipLaunchKernelGGL(k_seg_border, dim3(ncol_blocks + nrow_blocks), dim3(64), 0, st, d->wmask.p, d->bmask.p, d->seg_edges.p,
                               nwx * ((g.sh + SEG_TH - 1) / SEG_TH), g, nwx, d->parent.p, d->counters.p, ncol_blocks, nrb > 0 ? nrb : 1);
    }
    STAGE("k_seg_roots");
    hipLaunchKernelGGL(k_seg_roots, dim3((unsigned int)((nwords + 255) / 256)), dim3(256), 0, st, d->rootmask.p, g, nwx, d->parent.p, d->sizes.p);

    STAGE("k_seg_points");
    hipLaunchKernelGGL((k_seg_points<SEGP_PCAP, SEGP_RUNCAP, SEGP_NW, 1>), dim3(B, (nwx + SEGP_NW - 1) / SEGP_NW, pty), dim3(64 * SEGP_NW), 0, st,
                       d->wmask.p, d->bmask.p, g, nwx, d->parent.p, d->sizes.p, d->hkeys.p, d->hcounts.p, d->nslots - 1, d->stage_rec.p,
                       d->stage_pos.p, d->frame_cursor.p, d->stage_cap, d->dense_tiles.p, pty, d->counters.p);
    // tiles of workgroups that ran out of staging space (none in ordinary frames: the launch finds an empty list)
    hipLaunchKernelGGL((k_seg_points<SEGP_PCAP_DENSE, SEGP_RUNCAP_DENSE, 1, 2>), dim3(256), dim3(64), 0, st, d->wmask.p, d->bmask.p, g, nwx,
                       d->parent.p, d->sizes.p, d->hkeys.p, d->hcounts.p, d->nslots - 1, d->stage_rec.p, d->stage_pos.p, d->frame_cursor.p,
                       d->stage_cap, d->dense_tiles.p, pty, d->counters.p);
    STAGE("k_cluster_filter");
    hipLaunchKernelGGL(k_cluster_filter, dim3((d->nslots + 1023) / 1024), dim3(1024), 0, st, d->hkeys.p, d->hcounts.p, d->nslots, g,
                       d->clusters.p, d->slot_cluster.p, d->class_lists.p, d->max_clusters, d->max_points, d->counters.p);
    STAGE("k_point_place");
    hipLaunchKernelGGL(k_point_place, dim3(16, B), dim3(256), 0, st, d->stage_rec.p, d->stage_pos.p, d->frame_cursor.p, d->stage_cap,
                       d->slot_cluster.p, d->points.p, d->counters.p);

    // one launch per size class; each walks its own cluster list (grid-stride)
    range_pop();
    range_push("S5 quad fit");
    for (int cls = 0; cls < NCLASSES; cls++) {
        static const char *const kFitStage[NCLASSES] = {"k_fit_quads<0>", "k_fit_quads<1>", "k_fit_quads<2>", "k_fit_quads<3>", "k_fit_quads<4>"};
        STAGE(kFitStage[cls]);
        launch_fit_class(d, g, cls, B, st);
    }

    STAGE("k_quad_finish");
    launch_quad_finish(d, g, st);
    range_pop();
    range_push("S6-S7 edge refinement, homography, decode");
    STAGE("k_quad_compact");
    hipLaunchKernelGGL(k_quad_compact, dim3((d->max_clusters + 1023) / 1024), dim3(1024), 0, st, d->quads.p, d->counters.p, d->max_clusters,
                       d->quad_list.p);
    unsigned int dgrid = std::min<unsigned int>(d->max_clusters, std::max<unsigned int>(8192u, 64u * B));
    STAGE("k_refine");
    hipLaunchKernelGGL((g.channels == 1 ? k_refine<1> : k_refine<3>), dim3(dgrid), dim3(64), 0, st, d->quads.p, d->counters.p, d->max_clusters, d_frames, g,
                       d->fam.reversed_border ? 1 : 0, d->refine, d->quadH.p, d->quad_list.p, d->side_mom.p);
    STAGE("k_homography");
    hipLaunchKernelGGL(k_homography, dim3(std::min<unsigned int>((d->max_clusters + 31) / 32, 2048u)), dim3(256), 0, st, d->quadH.p, d->counters.p, d->max_clusters,
                       d->quad_list.p, d->side_mom.p);
    STAGE("k_decode");
    hipLaunchKernelGGL((g.channels == 1 ? k_decode<1> : k_decode<3>), dim3(dgrid), dim3(64), 0, st, d->quads.p, d->quadH.p, d->counters.p, d->max_clusters, d_frames, g, d->fam,
                       d->maxhamming, d->dets.p, d->max_dets, d->counters.p, d->quad_list.p);

    range_pop();
    range_push("S9 PnP, S8 de-duplication");
    if (cam) {
        STAGE("k_pnp_dets");
        const int tpw = pnp_tpw(d->nd_guess ? d->nd_guess : (size_t)20 * B);  // detections of the previous batch, else a guess
        hipLaunchKernelGGL(k_pnp_dets, dim3((d->max_dets + tpw - 1) / tpw), dim3(64), 0, st, d->dets.p, d->counters.p, d->max_dets, *cam, tpw);
    }
    // ---- S8: de-duplicate, order by id, lay out the results
    STAGE("k_det_dedup");
    {
        const unsigned int cap_f = d->dets_per_frame;
        hipLaunchKernelGGL(k_det_bucket, dim3((d->max_dets + 255) / 256), dim3(256), 0, st, d->dets.p, d->counters.p, d->max_dets, d->frame_ndets.p,
                           d->frame_idx.p, cap_f, d->counters.p);
        hipLaunchKernelGGL(k_det_dedup, dim3(B), dim3(256), 0, st, d->dets.p, d->counters.p, d->frame_ndets.p, d->frame_idx.p, cap_f,
                           d->frame_nkeep.p, d->counters.p);
        hipLaunchKernelGGL(k_det_offsets, dim3(1), dim3(1024), 0, st, d->frame_nkeep.p, B, d->frame_off.p, d->counters.p);
        hipLaunchKernelGGL(k_det_gather, dim3((cap_f + 255) / 256, B), dim3(256), 0, st, d->dets.p, d->frame_idx.p, cap_f, d->frame_nkeep.p,
                           d->frame_off.p, d->out_det.p, d->out_pose.p, d->max_dets, cam ? 1 : 0);
    }
    range_pop();
    if (d->profiling && d->nev <= MAX_STAGES) HIPCHK(hipEventRecord(d->ev[d->nev], st));
    HIPCHK(hipGetLastError());
    return ASL_OK;
}

static CamDev make_cam(const asl_detector *d, const double *K, const double *dist, int n_dist, double tag_size)
{
    CamDev c;
    c.both_minima = d->pnp_both_minima; c.pad = 0;
    c.fx = K[0]; c.fy = K[4]; c.cx = K[2]; c.cy = K[5];
    c.k1 = c.k2 = c.p1 = c.p2 = c.k3 = 0;
    if (dist && n_dist >= 4) { c.k1 = dist[0]; c.k2 = dist[1]; c.p1 = dist[2]; c.p2 = dist[3]; }
    if (dist && n_dist >= 5) c.k3 = dist[4];
    c.half = (double)(float)(tag_size / 2);  // object corners are float32 in the reference
    return c;
}

using clk = std::chrono::steady_clock;
static float msf(clk::time_point a, clk::time_point b) { return std::chrono::duration<float, std::milli>(b - a).count(); }

static int ensure_host_out(asl_detector *d, size_t want, size_t nframes)
{
    if (nframes > d->host_nkeep_cap) {
        if (d->host_nkeep) (void)hipHostFree(d->host_nkeep);
        d->host_nkeep = nullptr; d->host_nkeep_cap = 0;
        HIPCHK(hipHostMalloc((void **)&d->host_nkeep, nframes * sizeof(unsigned int), hipHostMallocDefault));
        d->host_nkeep_cap = nframes;
    }
    if (want <= d->host_cap) return ASL_OK;
    if (d->host_det) (void)hipHostFree(d->host_det);
    if (d->host_pose) (void)hipHostFree(d->host_pose);
    d->host_det = nullptr; d->host_pose = nullptr; d->host_cap = 0;
    want = std::max<size_t>(want * 2, 4096);
    HIPCHK(hipHostMalloc((void **)&d->host_det, want * sizeof(DetOut), hipHostMallocNonCoherent));  // coarse-grained: CPU-cached
    HIPCHK(hipHostMalloc((void **)&d->host_pose, want * sizeof(PoseOut), hipHostMallocNonCoherent));
    d->host_cap = want;
    return ASL_OK;
}

// enqueue one batch and the asynchronous read-back of its counters (and of as many detection records as the
// previous batch produced, so that the usual case needs no second copy); returns without waiting
static int submit_batch(asl_detector *d, const uint8_t *d_frames, const Geom &g, hipStream_t st, const CamDev *cam)
{
    if (d->pending) return fail(ASL_EINVAL, "a batch is already in flight on this detector: collect it first");
    int rc = ensure_workspace(d, g);
    if (rc) return rc;
    d->t_submit = clk::now();
    rc = enqueue_detect(d, d_frames, g, st, cam);
    if (rc) return rc;
    if (!d->pinned_counters) HIPCHK(hipHostMalloc((void **)&d->pinned_counters, sizeof(long long) * CNT__N, hipHostMallocDefault));
    HIPCHK(hipMemcpyAsync(d->pinned_counters, d->counters.p, sizeof(long long) * CNT__N, hipMemcpyDeviceToHost, st));
    d->prefetched = 0;
    rc = ensure_host_out(d, std::min<size_t>(d->nd_guess, d->max_dets), (size_t)g.nframes);
    if (rc) return rc;
    HIPCHK(hipMemcpyAsync(d->host_nkeep, d->frame_nkeep.p, sizeof(unsigned int) * (size_t)g.nframes, hipMemcpyDeviceToHost, st));
    if (d->nd_guess > 0) {  // as many results as the previous batch produced: the usual case needs no second copy
        size_t guess = std::min<size_t>(d->nd_guess, d->max_dets);
        HIPCHK(hipMemcpyAsync(d->host_det, d->out_det.p, guess * sizeof(DetOut), hipMemcpyDeviceToHost, st));
        if (cam) HIPCHK(hipMemcpyAsync(d->host_pose, d->out_pose.p, guess * sizeof(PoseOut), hipMemcpyDeviceToHost, st));
        d->prefetched = guess;
    }
    d->pending = true;
    d->p_frames = d_frames; d->p_geom = g; d->p_stream = st; d->p_has_cam = cam != nullptr;
    if (cam) d->p_cam = *cam;
    d->t_enqueued = clk::now();
    return ASL_OK;
}

static int collect_batch(asl_detector *d, asl_detection *out, asl_pose *poses, int max_out, int *n_per_frame, int *n_out)
{
    if (!d->pending) return fail(ASL_EINVAL, "no batch in flight on this detector");
    const Geom g = d->p_geom;
    hipStream_t st = d->p_stream;
    const CamDev *cam = d->p_has_cam ? &d->p_cam : nullptr;
    clk::time_point t0 = d->t_submit, t1 = d->t_enqueued, t2 = t1;
    for (int attempt = 0; attempt < 4; attempt++) {
        if (attempt > 0) {  // a work buffer overflowed: grow it and run the batch again, synchronously
            d->pending = false;
            d->nd_guess = 0;
            int rc = submit_batch(d, d->p_frames, g, st, cam);
            if (rc) return rc;
        }
        HIPCHK(hipStreamSynchronize(st));
        t2 = clk::now();
        memcpy(d->last_counters, d->pinned_counters, sizeof(long long) * CNT__N);
        d->last = g;
        long long *c = d->last_counters;
        if (c[CNT_UF_GUARD]) {
            d->pending = false;
            return fail(ASL_ECAPACITY, "a union-find loop hit its iteration guard (%lld times): labels of this batch are not trustworthy", c[CNT_UF_GUARD]);
        }
        bool again = false;
        if (c[CNT_OVERFLOW_HASH]) { d->hash_slots_per_frame *= 4; again = true; }
        if (c[CNT_OVERFLOW_CLUSTERS] || c[CNT_NCLUSTERS] > (long long)d->max_clusters) { d->clusters_per_frame *= 4; again = true; }
        if (c[CNT_OVERFLOW_POINTS]) { d->points_per_pixel *= 2; again = true; }
        if (c[CNT_OVERFLOW_DETS]) { d->dets_per_frame *= 4; again = true; }
        if (again) {
            if (attempt == 3) {
                d->pending = false;
                return fail(ASL_ECAPACITY, "work buffers still overflow after growing (hash %lld clusters %lld points %lld dets %lld)",
                            c[CNT_OVERFLOW_HASH], c[CNT_OVERFLOW_CLUSTERS], c[CNT_OVERFLOW_POINTS], c[CNT_OVERFLOW_DETS]);
            }
            continue;
        }
        break;
    }
    d->pending = false;
    if (d->profiling) {
        d->nstages = d->nev;
        for (int i = 0; i < d->nev; i++) HIPCHK(hipEventElapsedTime(&d->stage_ms[i], d->ev[i], d->ev[i + 1]));
    }
    if (d->last_counters[CNT_DEDUP_LIMIT])
        return fail(ASL_ECAPACITY, "%lld frame(s) hold more than %d detections: the de-duplication does not sort that many", d->last_counters[CNT_DEDUP_LIMIT], DEDUP_MAX);
    size_t nd = (size_t)d->last_counters[CNT_NKEEP];
    if (nd > d->max_dets) nd = d->max_dets;
    if (nd > d->prefetched) {
        size_t have = d->prefetched;
        if (nd > d->host_cap) have = 0;  // the staging buffers are about to be replaced
        int rc = ensure_host_out(d, nd, (size_t)g.nframes);
        if (rc) return rc;
        HIPCHK(hipMemcpy(d->host_det + have, d->out_det.p + have, (nd - have) * sizeof(DetOut), hipMemcpyDeviceToHost));
        if (cam) HIPCHK(hipMemcpy(d->host_pose + have, d->out_pose.p + have, (nd - have) * sizeof(PoseOut), hipMemcpyDeviceToHost));
    }
    d->nd_guess = nd + nd / 4 + 256;
    clk::time_point t3 = clk::now();
    // the device has de-duplicated, ordered by (frame, id) and laid the results out in the ABI's structs
    static_assert(sizeof(DetOut) == sizeof(asl_detection) && sizeof(PoseOut) == sizeof(asl_pose), "device results are copied verbatim");
    int total = (int)nd;
    int nw = std::min(total, max_out);
    if (out && nw > 0) memcpy(out, d->host_det, (size_t)nw * sizeof(asl_detection));
    if (poses && cam && nw > 0) memcpy(poses, d->host_pose, (size_t)nw * sizeof(asl_pose));
    if (n_per_frame) for (int f = 0; f < g.nframes; f++) n_per_frame[f] = (int)d->host_nkeep[f];
    if (n_out) *n_out = total;
    d->host_ms[0] = msf(t0, t1); d->host_ms[1] = msf(t1, t2); d->host_ms[2] = msf(t2, t3); d->host_ms[3] = msf(t3, clk::now());
    return ASL_OK;
}

static int check_device_args(asl_detector *d, const void *d_frames, int n_frames, int channels, int w, int h, int stride, size_t frame_pitch,
                             int n_dist, Geom *g)
{
    if (!d || !d_frames) return fail(ASL_EINVAL, "NULL detector or frames");
    if (n_dist != 0 && n_dist != 4 && n_dist != 5) return fail(ASL_EINVAL, "n_dist must be 0, 4 or 5");
    HIPCHK(hipSetDevice(d->device));
    int rc = make_geom(d, n_frames, channels, w, h, stride, frame_pitch, g);
    if (rc) return rc;
    if (frame_pitch < (size_t)stride * (size_t)h) return fail(ASL_EINVAL, "frame_pitch smaller than one frame");
    return ASL_OK;
}

extern "C" int asl_submit_batch_device(asl_detector *d, const void *d_frames, int n_frames, int channels, int w, int h, int stride,
                                       size_t frame_pitch, void *stream, const double *K, const double *dist, int n_dist, double tag_size)
{
    Geom g;
    int rc = check_device_args(d, d_frames, n_frames, channels, w, h, stride, frame_pitch, n_dist, &g);
    if (rc) return rc;
    CamDev cam;
    if (K) cam = make_cam(d, K, dist, n_dist, tag_size);
    return submit_batch(d, (const uint8_t *)d_frames, g, (hipStream_t)stream, K ? &cam : nullptr);
}

extern "C" int asl_collect_batch(asl_detector *d, asl_detection *out, asl_pose *poses, int max_out, int *n_per_frame, int *n_out)
{
    if (!d) return fail(ASL_EINVAL, "NULL detector");
    if (max_out < 0 || (max_out > 0 && !out)) return fail(ASL_EINVAL, "out is NULL");
    HIPCHK(hipSetDevice(d->device));
    return collect_batch(d, out, poses, max_out, n_per_frame, n_out);
}

extern "C" int asl_collect_batch_view(asl_detector *d, const asl_detection **out, const asl_pose **poses, const uint32_t **n_per_frame, int *n_out)
{
    if (!d || !out || !n_per_frame || !n_out) return fail(ASL_EINVAL, "NULL argument");
    HIPCHK(hipSetDevice(d->device));
    const bool with_poses = d->p_has_cam;
    int rc = collect_batch(d, nullptr, nullptr, 0, nullptr, n_out);  // waits, checks, fills the detector's page-locked result buffers
    if (rc) return rc;
    *out = reinterpret_cast<const asl_detection *>(d->host_det);
    if (poses) *poses = with_poses ? reinterpret_cast<const asl_pose *>(d->host_pose) : nullptr;
    *n_per_frame = d->host_nkeep;
    return ASL_OK;
}

extern "C" int asl_detect_batch_device(asl_detector *d, const void *d_frames, int n_frames, int channels, int w, int h, int stride,
                                       size_t frame_pitch, void *stream, const double *K, const double *dist, int n_dist,
                                       double tag_size, asl_detection *out, asl_pose *poses, int max_out, int *n_per_frame, int *n_out)
{
    if (max_out < 0 || (max_out > 0 && !out)) return fail(ASL_EINVAL, "out is NULL");
    int rc = asl_submit_batch_device(d, d_frames, n_frames, channels, w, h, stride, frame_pitch, stream, K, dist, n_dist, tag_size);
    if (rc) return rc;
    return collect_batch(d, out, poses, max_out, n_per_frame, n_out);
}

// Frames that start in host memory (the reference's callers hand over numpy frames: simulation_engine.py:219,
// video_detection.py:247-258).  All copies are issued up front on a copy stream, one event per chunk of frames; the
// detector works through the chunks on its own stream, each as soon as its frames have landed, so chunk i's kernels run
// under chunk i+1's transfer and the call takes the time of the transfer plus one chunk's kernels (the PCIe link is the
// bound: 2.76 MB per 720p frame against 3.5 us of kernels).  Results are appended chunk by chunk with the frame index
// of the whole call.  A call of fewer than two chunks is one batch, as before.
#define HOST_CHUNK_FRAMES 64
// frames [f0, f1) into the staging buffer; frames that follow each other in host memory (one array, the usual case) go as
// one transfer (a call per 2.76 MB frame costs ~12 us: 6 ms on 512 frames against 25 ms of transfer)
static hipError_t copy_frames(asl_detector *d, const uint8_t *const *frames, int f0, int f1, size_t pitch, hipStream_t st)
{
    int i = f0;
    while (i < f1) {
        int j = i + 1;
        while (j < f1 && frames[j] == frames[j - 1] + pitch) j++;
        hipError_t e = hipMemcpyAsync(d->in.p + (size_t)i * pitch, frames[i], pitch * (size_t)(j - i), hipMemcpyHostToDevice, st);
        if (e != hipSuccess) return e;
        i = j;
    }
    return hipSuccess;
}

static int detect_host_frames(asl_detector *d, const uint8_t *const *frames, int n_frames, int channels, int w, int h, int stride,
                              const CamDev *cam, asl_detection *out, asl_pose *poses, int max_out, int *n_per_frame, int *n_out)
{
    Geom g;
    const size_t pitch = (size_t)stride * (size_t)h;
    int rc = make_geom(d, n_frames, channels, w, h, stride, pitch, &g);
    if (rc) return rc;
    if (d->in.ensure(pitch * (size_t)n_frames)) return fail(ASL_ENOMEM, "input staging allocation failed");
    for (int i = 0; i < n_frames; i++)
        if (!frames[i]) return fail(ASL_EINVAL, "frames[%d] is NULL", i);
    int chunk = HOST_CHUNK_FRAMES;
    { const char *e = getenv("ASL_HOST_CHUNK"); if (e && atoi(e) > 0) chunk = atoi(e); }  // tuning override
    const int nchunks = n_frames >= 2 * chunk ? (n_frames + chunk - 1) / chunk : 1;
    if (nchunks == 1) {
        HIPCHK(copy_frames(d, frames, 0, n_frames, pitch, nullptr));
        int rcs = submit_batch(d, d->in.p, g, nullptr, cam);
        if (rcs) return rcs;
        return collect_batch(d, out, poses, max_out, n_per_frame, n_out);
    }
    if (!d->copy_stream) {
        HIPCHK(hipStreamCreateWithFlags(&d->copy_stream, hipStreamNonBlocking));
        HIPCHK(hipStreamCreateWithFlags(&d->host_stream, hipStreamNonBlocking));
    }
    while ((int)d->copy_done.size() < nchunks) {
        hipEvent_t e;
        HIPCHK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        d->copy_done.push_back(e);
    }
    HIPCHK(hipStreamSynchronize(nullptr));  // earlier work of this detector on the null stream may still read d->in
    for (int c = 0; c < nchunks; c++) {
        const int f0 = c * chunk, f1 = std::min(n_frames, f0 + chunk);
        HIPCHK(copy_frames(d, frames, f0, f1, pitch, d->copy_stream));
        HIPCHK(hipEventRecord(d->copy_done[c], d->copy_stream));
    }
    int total = 0;
    for (int c = 0; c < nchunks; c++) {
        const int f0 = c * chunk, f1 = std::min(n_frames, f0 + chunk);
        Geom gc = g;
        gc.nframes = f1 - f0;
        HIPCHK(hipStreamWaitEvent(d->host_stream, d->copy_done[c], 0));
        int rcs = submit_batch(d, d->in.p + (size_t)f0 * pitch, gc, d->host_stream, cam);
        if (rcs) { (void)hipStreamSynchronize(d->copy_stream); return rcs; }
        int nc = 0;
        const int room = std::max(0, max_out - total);
        rcs = collect_batch(d, out ? out + std::min(total, max_out) : nullptr, (poses && cam) ? poses + std::min(total, max_out) : nullptr, room,
                            n_per_frame ? n_per_frame + f0 : nullptr, &nc);
        if (rcs) { (void)hipStreamSynchronize(d->copy_stream); return rcs; }
        for (int k = total; k < std::min(total + nc, max_out); k++) out[k].frame += f0;  // frame index inside the whole call
        total += nc;
    }
    if (n_out) *n_out = total;
    return ASL_OK;
}

extern "C" int asl_detect_batch_u8(asl_detector *d, const uint8_t *const *frames, int n_frames, int channels, int w, int h, int stride,
                                   asl_detection *out, int max_out, int *n_per_frame, int *n_out)
{
    if (!d || !frames) return fail(ASL_EINVAL, "NULL detector or frames");
    if (max_out < 0 || (max_out > 0 && !out)) return fail(ASL_EINVAL, "out is NULL");
    HIPCHK(hipSetDevice(d->device));
    return detect_host_frames(d, frames, n_frames, channels, w, h, stride, nullptr, out, nullptr, max_out, n_per_frame, n_out);
}

extern "C" int asl_detect_batch_pose_u8(asl_detector *d, const uint8_t *const *frames, int n_frames, int channels, int w, int h, int stride,
                                        const double *K, const double *dist, int n_dist, double tag_size, asl_detection *out, asl_pose *poses,
                                        int max_out, int *n_per_frame, int *n_out)
{
    if (!d || !frames || !K || !poses) return fail(ASL_EINVAL, "NULL detector, frames, K or poses");
    if (max_out < 0 || (max_out > 0 && !out)) return fail(ASL_EINVAL, "out is NULL");
    if (n_dist != 0 && n_dist != 4 && n_dist != 5) return fail(ASL_EINVAL, "n_dist must be 0, 4 or 5");
    if (n_dist && !dist) return fail(ASL_EINVAL, "dist is NULL but n_dist = %d", n_dist);
    HIPCHK(hipSetDevice(d->device));
    CamDev cam = make_cam(d, K, dist, n_dist, tag_size);
    return detect_host_frames(d, frames, n_frames, channels, w, h, stride, &cam, out, poses, max_out, n_per_frame, n_out);
}

extern "C" int asl_detect_gray_u8(asl_detector *d, const uint8_t *gray, int w, int h, int stride, asl_detection *out, int max_out, int *n_out)
{
    const uint8_t *fr[1] = {gray};
    return asl_detect_batch_u8(d, fr, 1, 1, w, h, stride, out, max_out, nullptr, n_out);
}

extern "C" int asl_detect_bgr_u8(asl_detector *d, const uint8_t *bgr, int w, int h, int stride, asl_detection *out, int max_out, int *n_out)
{
    const uint8_t *fr[1] = {bgr};
    return asl_detect_batch_u8(d, fr, 1, 3, w, h, stride, out, max_out, nullptr, n_out);
}

extern "C" int asl_render_frames_device(asl_detector *d, void *d_frames, int n_frames, int w, int h, int stride, size_t frame_pitch,
                                        const void *d_planes, int max_planes, const void *d_textures, int tw, int th, double half,
                                        const double *K, const double *dist, int n_dist, void *stream)
{
    if (!d || !d_frames || !d_planes || !d_textures) return fail(ASL_EINVAL, "NULL argument");
    if (n_frames <= 0 || w <= 0 || h <= 0 || max_planes <= 0 || tw <= 0 || th <= 0) return fail(ASL_EINVAL, "sizes must be positive");
    if (stride < 3 * w || frame_pitch < (size_t)stride * (size_t)h) return fail(ASL_EINVAL, "stride / frame_pitch smaller than a BGR row / frame");
    if (dist && n_dist != 4 && n_dist != 5) return fail(ASL_EINVAL, "n_dist must be 4 or 5");
    if (dist && !K) return fail(ASL_EINVAL, "lens coefficients need the camera matrix");
    static_assert(sizeof(RenderPlane) == sizeof(asl_render_plane) && sizeof(asl_render_plane) == 96, "asl_render_plane layout");
    HIPCHK(hipSetDevice(d->device));
    RenderCam cam;
    memset(&cam, 0, sizeof cam);
    cam.iters = 8;
    if (K && dist) {
        cam.fx = K[0]; cam.fy = K[4]; cam.cx = K[2]; cam.cy = K[5];
        cam.k1 = dist[0]; cam.k2 = dist[1]; cam.p1 = dist[2]; cam.p2 = dist[3]; cam.k3 = n_dist >= 5 ? dist[4] : 0.0;
        cam.distort = 1;
    }
    hipLaunchKernelGGL(k_render, dim3((w + RENDER_TW - 1) / RENDER_TW, (h + 4 * RENDER_TH - 1) / (4 * RENDER_TH), (unsigned int)n_frames), dim3(64, 4), 0, (hipStream_t)stream, (uint8_t *)d_frames, w, h,
                       stride, frame_pitch, (const RenderPlane *)d_planes, max_planes, (const uint8_t *)d_textures, tw, th, half, cam);
    HIPCHK(hipGetLastError());
    return ASL_OK;
}

extern "C" int asl_pack_observations_device(asl_detector *d, void *d_obs, int max_tags, void *stream)
{
    if (!d || !d_obs) return fail(ASL_EINVAL, "NULL detector or output");
    if (max_tags <= 0) return fail(ASL_EINVAL, "max_tags must be positive");
    if (!d->pending) return fail(ASL_EINVAL, "no batch in flight on this detector: pack between submit and the next submit");
    static_assert(sizeof(ObsRec) == sizeof(asl_obs) && sizeof(asl_obs) == 136, "asl_obs layout");
    HIPCHK(hipSetDevice(d->device));
    const int nf = d->p_geom.nframes;
    const unsigned int total = (unsigned int)nf * (unsigned int)max_tags;
    hipLaunchKernelGGL(k_obs_pack, dim3((total + 255) / 256), dim3(256), 0, (hipStream_t)stream, d->dets.p, d->frame_idx.p, d->dets_per_frame,
                       d->frame_nkeep.p, d->counters.p, (ObsRec *)d_obs, nf, max_tags);
    HIPCHK(hipGetLastError());
    return ASL_OK;
}

extern "C" int asl_graph_frames_device(asl_detector *d, const void *d_obs, int world, int n_frames, int max_tags, int coordinate_id,
                                       double *d_pose, uint8_t *d_status, uint32_t *d_last, int n_ids, void *d_picks, void *stream)
{
    if (!d || !d_obs || !d_pose || !d_status || !d_last) return fail(ASL_EINVAL, "NULL argument");
    if (world <= 0 || n_frames <= 0 || max_tags <= 0 || n_ids <= 0) return fail(ASL_EINVAL, "sizes must be positive");
    HIPCHK(hipSetDevice(d->device));
    const unsigned int total = (unsigned int)world * (unsigned int)n_frames;
    const int lds_ids = (size_t)n_ids * sizeof(unsigned int) <= 48 * 1024 ? n_ids : 0;  // the table as an LDS copy per workgroup, if it fits
    hipLaunchKernelGGL(k_graph_frames, dim3((total + 255) / 256), dim3(256), sizeof(unsigned int) * (size_t)lds_ids, (hipStream_t)stream, (const ObsRec *)d_obs,
                       world, n_frames, max_tags, coordinate_id, d_pose, d_status, (unsigned int *)d_last, n_ids, lds_ids);
    if (d_picks)
        hipLaunchKernelGGL(k_graph_pick, dim3((n_ids + 255) / 256), dim3(256), 0, (hipStream_t)stream, (const ObsRec *)d_obs, world, n_frames, max_tags,
                           (const unsigned int *)d_last, n_ids, (ObsRec *)d_picks);
    HIPCHK(hipGetLastError());
    return ASL_OK;
}

extern "C" int asl_graph_picks_device(asl_detector *d, const void *d_obs, int world, int n_frames, int max_tags, const uint8_t *d_status,
                                      unsigned int order_lo, unsigned int order_hi, uint32_t *d_last, int n_ids, void *d_picks, void *stream)
{
    if (!d || !d_obs || !d_status || !d_last || !d_picks) return fail(ASL_EINVAL, "NULL argument");
    if (world <= 0 || n_frames <= 0 || max_tags <= 0 || n_ids <= 0) return fail(ASL_EINVAL, "sizes must be positive");
    if (order_lo > order_hi) return fail(ASL_EINVAL, "empty or reversed range");
    HIPCHK(hipSetDevice(d->device));
    HIPCHK(hipMemsetAsync(d_last, 0, sizeof(uint32_t) * (size_t)n_ids, (hipStream_t)stream));
    const unsigned int total = (unsigned int)world * (unsigned int)n_frames;
    const int lds_ids = (size_t)n_ids * sizeof(unsigned int) <= 48 * 1024 ? n_ids : 0;
    hipLaunchKernelGGL(k_graph_last_range, dim3((total + 255) / 256), dim3(256), sizeof(unsigned int) * (size_t)lds_ids, (hipStream_t)stream, (const ObsRec *)d_obs,
                       world, n_frames, max_tags, d_status, order_lo, order_hi, (unsigned int *)d_last, n_ids, lds_ids);
    hipLaunchKernelGGL(k_graph_pick, dim3((n_ids + 255) / 256), dim3(256), 0, (hipStream_t)stream, (const ObsRec *)d_obs, world, n_frames, max_tags,
                       (const unsigned int *)d_last, n_ids, (ObsRec *)d_picks);
    HIPCHK(hipGetLastError());
    return ASL_OK;
}

extern "C" int asl_solve_pnp_batch(asl_detector *d, const float *corners, const double *K, const double *dist, int n_dist,
                                   double tag_size, double *rvec, double *tvec, double *T, uint8_t *ok, int N)
{
    if (!d || !corners || !K || !rvec || !tvec || !T || !ok) return fail(ASL_EINVAL, "NULL argument");
    if (N < 0) return fail(ASL_EINVAL, "N < 0");
    if (n_dist != 0 && n_dist != 4 && n_dist != 5) return fail(ASL_EINVAL, "n_dist must be 0, 4 or 5");
    if (N == 0) return ASL_OK;
    HIPCHK(hipSetDevice(d->device));
    if (d->pnp_corners.ensure((size_t)N * 8) || d->pnp_out.ensure((size_t)N * 22) || d->pnp_ok.ensure((size_t)N))
        return fail(ASL_ENOMEM, "PnP workspace allocation failed");
    CamDev cam = make_cam(d, K, dist, n_dist, tag_size);
    HIPCHK(hipMemcpy(d->pnp_corners.p, corners, sizeof(float) * 8 * (size_t)N, hipMemcpyHostToDevice));
    double *dr = d->pnp_out.p, *dt = dr + 3 * (size_t)N, *dT = dt + 3 * (size_t)N;
    hipLaunchKernelGGL(k_pnp_batch, dim3((N + pnp_tpw((size_t)N) - 1) / pnp_tpw((size_t)N)), dim3(64), 0, nullptr, d->pnp_corners.p, N, cam, dr, dt, dT, d->pnp_ok.p, pnp_tpw((size_t)N));
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpy(rvec, dr, sizeof(double) * 3 * (size_t)N, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(tvec, dt, sizeof(double) * 3 * (size_t)N, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(T, dT, sizeof(double) * 16 * (size_t)N, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(ok, d->pnp_ok.p, (size_t)N, hipMemcpyDeviceToHost));
    return ASL_OK;
}

// asl_debug_fetch item 8: div_by(a, recip_of(d)) (asl_common.h) against a / d, as compiled into this library: log-uniform
// magnitudes with exponents within +-lim, both signs, a = 0 now and then
__global__ void __launch_bounds__(256) k_div_check(unsigned long long seed, int per_thread, int lim, unsigned long long *bad)
{
    unsigned long long s = seed + 0x9E3779B97F4A7C15ull * (blockIdx.x * (unsigned long long)blockDim.x + threadIdx.x + 1);
    auto rng = [&]() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return s; };
    for (int i = 0; i < 8; i++) rng();
    unsigned long long nbad = 0;
    for (int i = 0; i < per_thread; i++) {
        const unsigned long long u = rng(), v = rng(), w = rng();
        const int ea = (int)(u % (unsigned)(2 * lim + 1)) - lim, ed = (int)((u >> 20) % (unsigned)(2 * lim + 1)) - lim;
        double a = ldexp(1.0 + (double)(v & 0xFFFFFFFFFFFFFull) * 0x1p-52, ea);
        double dd = ldexp(1.0 + (double)(w & 0xFFFFFFFFFFFFFull) * 0x1p-52, ed);
        if (u & (1ull << 60)) a = -a;
        if (u & (1ull << 61)) dd = -dd;
        if ((u >> 40) % 257 == 0) a = 0.0;
        const double want = a / dd, got = div_by(a, recip_of(dd));
        if (__double_as_longlong(want) != __double_as_longlong(got)) nbad++;
    }
    if (nbad) atomicAdd(bad, nbad);
}

extern "C" int asl_debug_fetch(asl_detector *d, int what, void *dst, size_t bytes, size_t *n_items)
{
    if (!d || !dst || !n_items) return fail(ASL_EINVAL, "NULL argument");
    HIPCHK(hipSetDevice(d->device));
    const Geom &g = d->last;
    size_t total = (size_t)g.nframes * g.npix;
    if (what >= 0 && what <= 3 && total == 0) return fail(ASL_EINVAL, "no batch has run yet");
    switch (what) {
    case 0: {
        if (bytes < total) return fail(ASL_EINVAL, "dst too small: need %zu bytes", total);
        HIPCHK(hipMemcpy(dst, d->dgray.p, total, hipMemcpyDeviceToHost));
        *n_items = total;
        return ASL_OK;
    }
    case 1: {  // the pipeline keeps the threshold image as two bit masks per 64 pixels: expand it for the caller
        if (bytes < total) return fail(ASL_EINVAL, "dst too small: need %zu bytes", total);
        if (d->dbg_thresh.ensure(total)) return fail(ASL_ENOMEM, "debug buffer allocation failed");
        hipLaunchKernelGGL(k_seg_debug_thresh, dim3((g.sw + 63) / 64, (g.sh + 3) / 4, (unsigned int)g.nframes), dim3(64, 4), 0, nullptr,
                           d->wmask.p, d->bmask.p, g, seg_nwx(g), d->dbg_thresh.p);
        HIPCHK(hipGetLastError());
        HIPCHK(hipMemcpy(dst, d->dbg_thresh.p, total, hipMemcpyDeviceToHost));
        *n_items = total;
        return ASL_OK;
    }
    case 2: {  // labels live at run starts (run start -> tile-local root -> global root): resolve them per pixel for the caller
        if (bytes < total * 4) return fail(ASL_EINVAL, "dst too small: need %zu bytes", total * 4);
        if (d->dbg_labels.ensure(total)) return fail(ASL_ENOMEM, "debug buffer allocation failed");
        hipLaunchKernelGGL(k_seg_debug_labels, dim3((g.sw + 63) / 64, (g.sh + 3) / 4, (unsigned int)g.nframes), dim3(64, 4), 0, nullptr,
                           d->wmask.p, d->bmask.p, g, seg_nwx(g), d->parent.p, d->dbg_labels.p);
        HIPCHK(hipGetLastError());
        HIPCHK(hipMemcpy(dst, d->dbg_labels.p, total * 4, hipMemcpyDeviceToHost));
        *n_items = total;
        return ASL_OK;
    }
    case 3: {  // sizes are kept at the global roots; "no contrast" pixels are singletons whose size is implied
        if (bytes < total * 4) return fail(ASL_EINVAL, "dst too small: need %zu bytes", total * 4);
        if (d->dbg_thresh.ensure(total)) return fail(ASL_ENOMEM, "debug buffer allocation failed");
        hipLaunchKernelGGL(k_seg_debug_thresh, dim3((g.sw + 63) / 64, (g.sh + 3) / 4, (unsigned int)g.nframes), dim3(64, 4), 0, nullptr,
                           d->wmask.p, d->bmask.p, g, seg_nwx(g), d->dbg_thresh.p);
        HIPCHK(hipGetLastError());
        std::vector<uint8_t> th(total);
        HIPCHK(hipMemcpy(th.data(), d->dbg_thresh.p, total, hipMemcpyDeviceToHost));
        HIPCHK(hipMemcpy(dst, d->sizes.p, total * 4, hipMemcpyDeviceToHost));
        unsigned int *o = (unsigned int *)dst;
        for (size_t i = 0; i < total; i++)
            if (th[i] == 127) o[i] = 1u;
        *n_items = total;
        return ASL_OK;
    }
    case 4: {
        size_t ncl = (size_t)std::min<long long>(d->last_counters[CNT_NCLUSTERS], (long long)d->max_clusters);
        std::vector<QuadRec> q(ncl);
        if (ncl) HIPCHK(hipMemcpy(q.data(), d->quads.p, ncl * sizeof(QuadRec), hipMemcpyDeviceToHost));
        std::sort(q.begin(), q.end(), [](const QuadRec &a, const QuadRec &b) { return a.key < b.key; });
        asl_debug_quad *o = (asl_debug_quad *)dst;
        size_t cap = bytes / sizeof(asl_debug_quad), k = 0;
        for (size_t i = 0; i < ncl; i++) {
            if (!q[i].valid) continue;
            if (k >= cap) return fail(ASL_EINVAL, "dst too small for the quads");
            for (int a = 0; a < 4; a++) { o[k].p[a][0] = q[i].p[a][0]; o[k].p[a][1] = q[i].p[a][1]; }
            unsigned long long key = q[i].key;
            o[k].frame = (int)(key >> 48);
            o[k].cluster = (((key >> 24) & 0xFFFFFFull) << 32) + (key & 0xFFFFFFull);
            o[k].reversed_border = q[i].reversed_border;
            k++;
        }
        *n_items = k;
        return ASL_OK;
    }
    case 5: {
        if (bytes < sizeof(long long) * 18) return fail(ASL_EINVAL, "dst too small");
        long long *o = (long long *)dst;
        o[0] = g.nframes; o[1] = g.sw; o[2] = g.sh;
        o[3] = d->last_counters[CNT_NCLUSTERS]; o[4] = d->last_counters[CNT_NPOINTS]; o[5] = d->last_counters[CNT_NQUADS];
        o[6] = d->last_counters[CNT_NDETS]; o[7] = d->nslots; o[8] = d->max_clusters; o[9] = d->max_points; o[10] = d->max_dets;
        o[11] = d->last_counters[CNT_OVERFLOW_HASH]; o[12] = d->last_counters[CNT_OVERFLOW_CLUSTERS];
        o[13] = d->last_counters[CNT_OVERFLOW_POINTS]; o[14] = d->last_counters[CNT_OVERFLOW_DETS]; o[15] = d->last_counters[CNT_CLASS0];
        o[16] = d->last_counters[CNT_DENSE_TILES]; o[17] = d->last_counters[CNT_DENSE_SEG];  // tiles that took the dense launches
        *n_items = 18;
        return ASL_OK;
    }
    case 6: {  // clusters as the quad fit receives them: (key, count, hash of the sorted point records), ordered by key
        size_t ncl = (size_t)std::min<long long>(d->last_counters[CNT_NCLUSTERS], (long long)d->max_clusters);
        size_t npts = (size_t)std::min<long long>(d->last_counters[CNT_NPOINTS], (long long)d->max_points);
        if (bytes < ncl * 24) return fail(ASL_EINVAL, "dst too small: need %zu bytes", ncl * 24);
        std::vector<ClusterRec> cl(ncl);
        std::vector<unsigned long long> pts(npts);
        if (ncl) HIPCHK(hipMemcpy(cl.data(), d->clusters.p, ncl * sizeof(ClusterRec), hipMemcpyDeviceToHost));
        if (npts) HIPCHK(hipMemcpy(pts.data(), d->points.p, npts * sizeof(unsigned long long), hipMemcpyDeviceToHost));
        std::sort(cl.begin(), cl.end(), [](const ClusterRec &a, const ClusterRec &b) { return a.key < b.key; });
        unsigned long long *o = (unsigned long long *)dst;
        for (size_t i = 0; i < ncl; i++) {
            unsigned long long h = 0xcbf29ce484222325ull;
            if ((size_t)cl[i].offset + cl[i].count <= npts) {
                std::sort(pts.begin() + cl[i].offset, pts.begin() + cl[i].offset + cl[i].count);
                for (unsigned int k = 0; k < cl[i].count; k++) { h ^= pts[(size_t)cl[i].offset + k]; h *= 0x100000001b3ull; }
            }
            o[3 * i] = cl[i].key; o[3 * i + 1] = cl[i].count; o[3 * i + 2] = h;
        }
        *n_items = ncl;
        return ASL_OK;
    }
    case 7: {  // run the quad fit of the last batch again, `bytes` times, on the buffers it left behind: dst (int64[2 + NCLASSES])
        // receives the repetitions, the quads that came out differently from the first repetition, and those by size class.
        // The fit is a pure function of the clusters, so any difference is a race.
        if (d->pending) return fail(ASL_EINVAL, "a batch is in flight on this detector");
        size_t ncl = (size_t)std::min<long long>(d->last_counters[CNT_NCLUSTERS], (long long)d->max_clusters);
        if (!ncl) return fail(ASL_EINVAL, "no clusters in the last batch");
        const size_t reps = bytes;
        std::vector<QuadRec> ref(ncl), cur(ncl);
        std::vector<ClusterRec> cl(ncl);
        HIPCHK(hipMemcpy(cl.data(), d->clusters.p, ncl * sizeof(ClusterRec), hipMemcpyDeviceToHost));
        long long *o = (long long *)dst;
        for (int k = 0; k < 2 + NCLASSES; k++) o[k] = 0;
        for (size_t r = 0; r < reps; r++) {
            // the all-LDS classes only: the global-slab class (more than 1024 points) sorts and de-duplicates inside its
            // clusters' point records, so a second run of it would not see the first one's input
            for (int cls = 0; cls < NCLASSES - 1; cls++) launch_fit_class(d, g, cls, (unsigned int)g.nframes, nullptr);
            launch_quad_finish(d, g, nullptr);
            HIPCHK(hipGetLastError());
            HIPCHK(hipMemcpy((r ? cur : ref).data(), d->quads.p, ncl * sizeof(QuadRec), hipMemcpyDeviceToHost));
            if (!r) continue;
            for (size_t i = 0; i < ncl; i++) {
                const bool same = cur[i].valid == ref[i].valid && (!ref[i].valid || memcmp(cur[i].p, ref[i].p, sizeof ref[i].p) == 0);
                if (same) continue;
                const unsigned int cnt = cl[i].count;
                if (cnt > CLASS3_CAP) continue;  // not re-run (above)
                o[1]++;
                o[2 + (cnt <= CLASS0_CAP ? 0 : (cnt <= CLASS1_CAP ? 1 : (cnt <= CLASS2_CAP ? 2 : (cnt <= CLASS3_CAP ? 3 : 4))))]++;
            }
        }
        o[0] = (long long)reps;
        *n_items = 2 + NCLASSES;
        return ASL_OK;
    }
    case 8: {  // the shared-reciprocal division of the line fits against the compiler's division: dst int64[2] = pairs, mismatches
        unsigned long long *d_bad = nullptr, h_bad = 0;
        const int lim = bytes > 0 && bytes <= 900 ? (int)bytes : 100, blocks = 2048, per = 1024;
        HIPCHK(hipMalloc((void **)&d_bad, sizeof h_bad));
        HIPCHK(hipMemset(d_bad, 0, sizeof h_bad));
        hipLaunchKernelGGL(k_div_check, dim3(blocks), dim3(256), 0, nullptr, 20260301ull, per, lim, d_bad);
        hipError_t e8 = hipMemcpy(&h_bad, d_bad, sizeof h_bad, hipMemcpyDeviceToHost);
        (void)hipFree(d_bad);
        if (e8 != hipSuccess) return fail(ASL_EDEVICE, "division check failed: %s", hipGetErrorString(e8));
        ((long long *)dst)[0] = (long long)blocks * 256 * per;
        ((long long *)dst)[1] = (long long)h_bad;
        *n_items = 2;
        return ASL_OK;
    }
    default:
        return fail(ASL_EINVAL, "unknown debug item %d", what);
    }
}

extern "C" int asl_debug_phase_cycles(asl_detector *d, unsigned long long *out64, int reset)
{
    if (!d || !out64) return fail(ASL_EINVAL, "NULL argument");
    HIPCHK(hipSetDevice(d->device));
    std::vector<unsigned long long> all((size_t)64 * PHASE_SPREAD);
    HIPCHK(hipMemcpyFromSymbol(all.data(), HIP_SYMBOL(g_phase_cycles), sizeof(unsigned long long) * all.size()));
    for (int i = 0; i < 64; i++) {
        out64[i] = 0;
        for (int k = 0; k < PHASE_SPREAD; k++) out64[i] += all[(size_t)i * PHASE_SPREAD + k];
    }
    if (reset) {
        std::fill(all.begin(), all.end(), 0ull);
        HIPCHK(hipMemcpyToSymbol(HIP_SYMBOL(g_phase_cycles), all.data(), sizeof(unsigned long long) * all.size()));
    }
    return ASL_OK;
}

#include "gn_host.inc"
