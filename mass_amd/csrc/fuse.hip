// fuse.hip — the per-frame voxel-map update for gfx950 (MI355X).
//
// Replaces, per call, the torch op sequences of the reference's
//   transform_rays      /root/reference/mass/utils/projection.py:77-110
//   bin_rays            projection.py:113-230
//   update_feature_map  projection.py:233-351
// as driven by BaseProjectionLayer.update (mass/nn/base_projection_layer.py:282-343).
//
// Design: the reference's read-modify-write blend
//     new[v] = sum_k ((1 - iw*w_k)*old[v] + iw*w_k*f_k) * w_k / W[v]
// has the closed form
//     new[v] = a*old[v] + g*U ,  a = 1 - iw*S2/W, g = iw/W, W = sum w_k, S2 = sum w_k^2, U = sum w_k^2 f_k
// and a sequence of frames unrolls to  m_n = s_n * (m_0 + sum_f (g_f / s_f) U_f),  s_f = prod a.
// One workgroup per *map tile* keeps the tile in LDS across all frames of the call:
//   1. count     : every pixel is unprojected and binned (bit-exact integer path), its 2x2x2
//                  footprint is mapped to the <= 8 tiles it overlaps, and (tile, frame) bucket
//                  sizes are counted through a per-block LDS hash, so a block issues one global
//                  integer atomic per distinct bucket (blocks own 16 x 16 pixel patches); the
//                  binned pixel is written out for step 3;
//   2. scan      : exclusive prefix sum of the bucket sizes; list of non-empty tiles in eight load
//                  classes, heaviest first; the call's density picks the tile kernel;
//   3. scatter   : writes the point record (16 bytes, + a 4-byte class id / feature index where the record
//                  format has no room for it) of every (pixel, tile) pair into its bucket slot
//                  (slot = bucket base + LDS-local rank);
//   4. tile kernel, one of
//        fuse_cells_kernel   sequential frames of class ids / ones: all frames of a 4 x 4 x 8 tile at once through
//                  compact (voxel, frame) cells, integer sums only, three 256-thread workgroups per CU (see there).
//                  A batch of unrelated frames reaches it as CONTRIBUTIONS (make_contribution: scatter_kernel expands a
//                  point into one 8-byte entry per corner of its footprint, in the bucket of the corner's tile), a real
//                  scene as AGGREGATED entries (bucket_agg_kernel: a block's corners summed per (tile, frame, voxel,
//                  class) in LDS before anything is written - collision compaction; fuse_cells_kernel<AGG>);
//        fuse_dense_kernel   the same feature kinds, real scenes as 16-byte tile-local records (MF_AGG=0, maps of more
//                  than 2^17 tiles, merged batches): everything accumulated as integers on 4 x 4 x 8 tiles, suffix form
//                  of the unrolled blend (see there);
//                  (probe_kernel samples the call's points and picks the entry format, tile_list_kernel the tile kernel
//                  that goes with it: a function of the call's data alone, no state is kept between calls)
//        fuse_tiles_kernel   dense fp32 features, blend weights outside [0, 1], any tile shape: persistent
//                  workgroups walk the tile list (ticket counter; the next tile's ticket, id,
//                  offsets and first records are fetched one tile ahead).  The tile's old map
//                  values are preloaded into LDS by LDS-DMA while pass 1 runs; frames are taken
//                  in chunks: pass 1 accumulates W, S2 as 64-bit fixed-point LDS integer atomics
//                  (ds_add_f32 is 29x slower than ds_add_u32 on gfx950), pass 2 turns them into
//                  k_f = g_f / s_f per voxel, pass 3 adds w^2 k_f feat (compare-and-swap, float
//                  atomic only on a lost race); the final pass writes s * D row by row;
//        fuse_single_kernel / fuse_single_dense_kernel   single-group calls: one pass, integer sums.
// No global float atomics are used (guide: ~1.3 TB/s, 17x slower when scattered); HBM sees each
// tile once per call, coalesced along z.
//
// mf_fuse_frame_maps (the agent's loop over its maps per simulator step, navigation_policy.py:164-171): steps 1-3 do
// not depend on the features, so a single group is bucketed ONCE for up to four maps that share grid and frames
// (count_kernel also checks the other maps' class ids, finds their feature range and zeroes their counters;
// tile_list_kernel makes one list per map; scatter_kernel writes one aux word per record and map), and the maps'
// tile kernels run side by side (MultiCtx, map_streams).
//
// Tuning / diagnostics, all off by default and range-checked (env_int): MF_TILE="s0 s1 s2 threads [gc]"
// overrides the tile shape of fuse_tiles_kernel, MF_DENSE=0 keeps calls off the 4 x 4 x 8 integer kernels,
// MF_DENSE_FORCE / MF_CELLS_FORCE give every eligible call to that kernel, MF_CELLS=0 keeps the cells kernel out,
// MF_FORMAT=contributions / records / aggregated overrides the probe's entry format per call, MF_AGG=0 keeps real scenes on records,
// MF_DENSE_GC / MF_DENSE_NT / MF_CELLS_PER_CU size them, MF_BLOCKS caps the workgroups,
// MF_STAMPS=1 prints the share of each phase of the tile kernel (dev builds of bench runs).
#include <algorithm>
#include <cstdlib>
#include <type_traits>
#include <mutex>
#include <unordered_map>
#include <vector>
#include "common.h"
#include "geometry.h"

namespace mf {

// ----------------------------------------------------------------------------
// parameters shared by the pipeline kernels (passed by value)
// ----------------------------------------------------------------------------
struct FuseParams {
    // grid
    int size0, size1, size2, C;
    Bins bins;                 // b0 = bins_x, b1 = bins_y, b2 = bins_z
    float *map;
    // frames (front end 0)
    int n_frames, H, W;
    const float *cam, *poses, *depth;
    int pose_inline;           // the one frame's pose rides in pose0 (mf_frames.poses_on_host), `poses` is not read
    float pose0[12];
    const void *feat;
    int feat_kind, fh, fw, rep_y, rep_x;
    float min_d, max_d;
    int *label_status;         // optional: set to 1 (and the call aborted) on a class id outside [0, C)
    // binned points (front end 1)
    const int64_t *i0, *i1, *i2;
    const float *q0, *q1, *q2;
    // common
    long long n_points;
    int G;                     // sequential groups
    float iw;
    // tiling
    int s0, s1, s2;            // log2 tile extents
    int nt0, nt1, nt2, n_tiles, n_keys;
    int gc;                    // frames per chunk in the tile kernel
    int vec4;                  // final pass may use 16-byte accesses
    int meta;                  // entries in a tile-local format (all-integer tile kernels, no aux words): contributions or meta records, see use_contributions
    int fmt_force;             // entry format of a tile-local call: see entry_format
    unsigned magicC;           // ceil(2^32 / C) (0 when C == 1)
    // workspace
    int *cursor;               // [n_keys + 1]
    int *block_sums;
    int *ticket;               // [1 + TILE_CLASSES]
    int *active;               // [TILE_CLASSES][n_tiles]
    uint4 *rec;
    uint32_t *aux;
    uint4 *pts;                // [n_points] (front end 0) the binned pixels, written by count_kernel for scatter_kernel
    unsigned *absmax;          // front end 0, dense features on the single-pass path: bits of max |feature| (count_kernel), else NULL
    // further maps updated from the same frames (mf_fuse_frame_maps): they share the records, each has its own word per record
    int n_extra;
    struct ExtraMap {
        const void *feat;
        int feat_kind, C, fh, fw, rep_y, rep_x;
        uint32_t *aux;         // [cap] class id / feature pixel per record (NULL: ones)
        int *label_status;     // optional, as above
        int *abort;            // its tile_list_kernel lists nothing when this word is set
        unsigned *absmax;      // as above, for this map (a word of the first map's counter block), else NULL
        uint4 *zero;           // its counters and split scratch, zeroed here (count_kernel) ...
        unsigned zero16;       // ... this many 16-byte words
    } extra[3];
};
constexpr int MAX_EXTRA_MAPS = 3;

// What the tile kernel needs (a subset: fewer scalar registers held across its loop).
struct TileParams {
    int size0, size1, size2, C;
    float *map;
    const void *feat;
    int G;
    float iw;
    int s0, s1, s2;
    int nt1, nt2, n_tiles;
    unsigned magicC;
    int gc, vec4, fx_shift;
    int meta;                  // the entries are tile-local contributions (see make_contribution)
    int cells_cap;             // fuse_cells_kernel: (voxel, frame) cells that fit its LDS
    const int *cursor;
    int *ticket;
    int *ctr;                  // this kernel's work counter (one word of `ticket` per tile kernel)
    const int *active;
    const uint4 *rec;
    const uint32_t *aux;
    const int4 *light;         // fuse_cells_kernel's work items [CELL_CLASSES][n_tiles] (tile_list_kernel): {tile, first entry, entries, origin 10 + 10 + 10 bits}
};

struct Point {
    int k0, k1, k2;            // map dims: 0 = y (flipped), 1 = x, 2 = z
    float r0, r1, r2;
    int group;
};

__device__ __forceinline__ uint32_t read_label(const void *feat, int kind, long long i)
{
    if (kind == MF_FEAT_LABEL_U8) return ((const uint8_t *)feat)[i];
    if (kind == MF_FEAT_LABEL_I32) return (uint32_t)((const int32_t *)feat)[i];
    const long long v = ((const int64_t *)feat)[i];
    return (v < 0 || v > 0x7fffffffLL) ? 0xffffffffu : (uint32_t)v;
}

// Pixel (y, x) of the frame a block works on (front end 0: blocks own 16 x 16 patches, so row and column come
// from the block and thread index: no division by the image width).
struct Pix { int y, x; };
__device__ __forceinline__ Pix patch_pixel(const FuseParams &P);
__device__ __forceinline__ Pix patch_pixel_at(const FuseParams &P, int bx);

// Index of the feature pixel under frame pixel (y, x) of frame f: features may be coarser than the frame by whole
// factors (repeat_interleave upsampling, base_projection_layer.py:322-325); the usual factor 1 takes no division.
__device__ __forceinline__ long long feature_pixel(int f, int y, int x, int fh, int fw, int rep_y, int rep_x)
{
    if (rep_y != 1 || rep_x != 1) { y /= rep_y; x /= rep_x; }       // (uniform)
    return ((long long)f * fh + y) * fw + x;
}

// Front end 0: pixel of a posed frame -> binned point (a3 + a4).
// Front end 1: already binned point arrays (functional update_feature_map).
// (bx, by: the block's patch and frame - blockIdx.x / blockIdx.y in the kernels launched as (patches, n_frames) blocks,
// the work item of a kernel that loops over them)
template <int FRONT>
__device__ __forceinline__ bool get_point_at(const FuseParams &P, long long idx, Point &pt, uint32_t &aux, int bx, int by)
{
    if (FRONT == 0) {
        // frames are launched on blockIdx.y (point_index), so no 64-bit divide is needed here
        const int HW = P.H * P.W;
        const int f = by;
        const int pix = (int)(idx - (long long)f * HW);
        const float d = P.depth[idx];
        float pose[12];
#pragma unroll
        for (int k = 0; k < 12; ++k) pose[k] = P.pose_inline ? P.pose0[k] : P.poses[f * 12 + k];
        float q0, q1, q2;
        rotate_ray(pose + 3, P.cam[pix * 3], P.cam[pix * 3 + 1], P.cam[pix * 3 + 2], q0, q1, q2);
        const float m0 = q0 * d, m1 = q1 * d, m2 = q2 * d;
        const float p0 = pose[0] + m0, p1 = pose[1] + m1, p2 = pose[2] + m2;
        int kx, ky, kz; float rx, ry, rz;
        const bool ok = bin_point(P.bins, p0, p1, p2, d, P.min_d, P.max_d, kx, ky, kz, rx, ry, rz);
        pt.k0 = ky; pt.k1 = kx; pt.k2 = kz; pt.r0 = ry; pt.r1 = rx; pt.r2 = rz;
        pt.group = P.G == 1 ? 0 : f;
        // The class id is read here only when ids are being checked, then for every pixel, also one that misses the
        // map: the reference's one_hot looks at the whole image (semantic_projection_layer.py:203-209).
        // (scatter_kernel reads the word that goes with the record itself.)
        if (P.label_status && P.feat_kind >= MF_FEAT_LABEL_U8 && P.feat_kind <= MF_FEAT_LABEL_I64) {
            const Pix px = patch_pixel_at(P, bx);
            aux = read_label(P.feat, P.feat_kind, feature_pixel(f, px.y, px.x, P.fh, P.fw, P.rep_y, P.rep_x));
        }
        return ok;
    } else {
        const long long a = P.i0[idx], b = P.i1[idx], c = P.i2[idx];
        const bool ok = a >= 0 && a < P.size0 && b >= 0 && b < P.size1 && c >= 0 && c < P.size2;
        pt.k0 = (int)a; pt.k1 = (int)b; pt.k2 = (int)c;
        pt.r0 = P.q0[idx]; pt.r1 = P.q1[idx]; pt.r2 = P.q2[idx];
        pt.group = 0;
        if (ok && P.feat_kind != MF_FEAT_ONES)
            aux = P.feat_kind == MF_FEAT_DENSE_F32 ? (uint32_t)idx : read_label(P.feat, P.feat_kind, idx);
        return ok;
    }
}
template <int FRONT>
__device__ __forceinline__ bool get_point(const FuseParams &P, long long idx, Point &pt, uint32_t &aux)
{
    return get_point_at<FRONT>(P, idx, pt, aux, (int)blockIdx.x, (int)blockIdx.y);
}

// Global point index of this thread, or -1.  Front end 0 is launched as (patches, n_frames)
// blocks where a 256-thread block owns a 16 x 16 pixel patch: a compact patch is a narrow
// pencil in space, so its points fall into few map tiles (fewer distinct buckets per block =
// fewer global atomics and longer contiguous record runs than a 256 x 1 pixel strip).
// Front end 1 is a flat grid.
constexpr int PATCH = 16;

template <int FRONT>
__device__ __forceinline__ long long point_index(const FuseParams &P, int threads)
{
    if (FRONT == 0) {
        const int pw = (P.W + PATCH - 1) / PATCH;
        const int py = blockIdx.x / pw, px = blockIdx.x - py * pw;
        const int y = py * PATCH + (int)(threadIdx.x / PATCH), x = px * PATCH + (int)(threadIdx.x % PATCH);
        return (y < P.H && x < P.W) ? (long long)blockIdx.y * (P.H * P.W) + y * P.W + x : -1;
    }
    const long long idx = (long long)blockIdx.x * threads + threadIdx.x;
    return idx < P.n_points ? idx : -1;
}

__device__ __forceinline__ Pix patch_pixel_at(const FuseParams &P, int bx)
{
    const int pw = (P.W + PATCH - 1) / PATCH;
    const int py = bx / pw, px = bx - py * pw;       // (scalar: once per block)
    Pix p;
    p.y = py * PATCH + (int)(threadIdx.x / PATCH);
    p.x = px * PATCH + (int)(threadIdx.x % PATCH);
    return p;
}
__device__ __forceinline__ Pix patch_pixel(const FuseParams &P) { return patch_pixel_at(P, (int)blockIdx.x); }

// The <= 8 (tile, group) buckets a point's footprint overlaps, at fixed positions: key[4 a + 2 b + c] for the lower / upper tile per axis, bit j of the result set
// where the combination is a tile of its own (an axis whose two corners share a tile counts once, as a = 0).  Fixed
// positions keep keys, slots and ranks of count_kernel / scatter_kernel in registers (unrolled loops, no indexed arrays)
// and let scatter_kernel have the returning atomics of all of a point's buckets in flight before the first record store.
struct TileKeys {
    uint32_t key[8];
    unsigned mask;
    unsigned straddle;         // bit 2 / 1 / 0: the footprint's two corners of axis 0 / 1 / 2 lie in different tiles
    AxisFoot a0, a1, a2;
    int t0[2], t1[2], t2[2];
};
__device__ __forceinline__ void point_keys8(const FuseParams &P, const Point &pt, TileKeys &K)
{
    K.a0 = axis_foot(pt.k0, pt.r0, P.size0); K.a1 = axis_foot(pt.k1, pt.r1, P.size1); K.a2 = axis_foot(pt.k2, pt.r2, P.size2);
    K.t0[0] = K.a0.lo >> P.s0; K.t0[1] = K.a0.hi >> P.s0;
    K.t1[0] = K.a1.lo >> P.s1; K.t1[1] = K.a1.hi >> P.s1;
    K.t2[0] = K.a2.lo >> P.s2; K.t2[1] = K.a2.hi >> P.s2;
    const bool m0 = K.t0[0] != K.t0[1], m1 = K.t1[0] != K.t1[1], m2 = K.t2[0] != K.t2[1];
    K.straddle = (m0 ? 4u : 0u) | (m1 ? 2u : 0u) | (m2 ? 1u : 0u);
    K.mask = 0u;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int a = j >> 2, b = (j >> 1) & 1, c = j & 1;
        const bool valid = (a == 0 || m0) && (b == 0 || m1) && (c == 0 || m2);
        K.key[j] = (uint32_t)(((K.t0[a] * P.nt1 + K.t1[b]) * P.nt2 + K.t2[c]) * P.G + pt.group);
        K.mask |= valid ? 1u << j : 0u;
    }
}

// ----------------------------------------------------------------------------
// per-block LDS hash: bucket key -> (count, base)
// ----------------------------------------------------------------------------
constexpr int TILE_CLASSES = 8;          // load classes of the tile work list (tile_list_kernel)
constexpr int ABORT_SLOT = 1 + TILE_CLASSES;   // ticket[ABORT_SLOT] != 0: a class id was out of range, the call is called off
constexpr int BIN_THREADS = 256;
constexpr int HS_BITS = 8;
constexpr int HS = 1 << HS_BITS;
constexpr uint32_t EMPTY = 0xffffffffu;

// Entry format of a call bucketed on 4 x 4 x 8 tiles (P.meta): probe_kernel counts, over a sample of pixel patches, the valid
// points and the distinct tiles their voxels lie in; count_kernel, scatter_kernel and tile_list_kernel all read the two words
// and agree: a real scene (tens of points of a 16 x 16 patch in one tile) keeps 16-byte point records and goes to
// fuse_dense_kernel, a batch of unrelated frames (about one point per tile) is expanded into contributions for
// fuse_cells_kernel.  A function of the call's data alone.
constexpr int PROBE_POINTS = ABORT_SLOT + 10, PROBE_TILES = ABORT_SLOT + 11;
constexpr int FORMAT_SLOT = ABORT_SLOT + 12;    // diagnostics: the call's entry format, FMT_* (tile_list_kernel)
constexpr int PROBE_DENSE_RATIO = 8;            // points per distinct tile of a patch from which a scene counts as real
// Entry format of a call bucketed on 4 x 4 x 8 tiles.  fmt_force: 0 the probe decides between contributions (sparse) and records
// (real scene), 1 contributions, 2 records, 3 aggregated entries, 4 the probe decides between contributions and aggregated entries.
constexpr int FMT_RECORDS = 0, FMT_CONTRIB = 1, FMT_AGG = 2;
__device__ __forceinline__ int entry_format(const int *ticket, int meta, int fmt_force)
{
    if (!meta) return FMT_RECORDS;
    if (fmt_force == 1) return FMT_CONTRIB;
    if (fmt_force == 2) return FMT_RECORDS;
    if (fmt_force == 3) return FMT_AGG;
    const bool sparse = ticket[PROBE_POINTS] < PROBE_DENSE_RATIO * ticket[PROBE_TILES];
    return sparse ? FMT_CONTRIB : fmt_force == 4 ? FMT_AGG : FMT_RECORDS;
}
__device__ __forceinline__ bool use_contributions(const int *ticket, int meta, int fmt_force)
{
    return entry_format(ticket, meta, fmt_force) == FMT_CONTRIB;
}

// The table pays when the keys of a block repeat (neighbouring pixels of a real scene share their tiles: tens of points
// per key).  A block of unrelated depths brings ~450 distinct keys for 256 slots: once HS_FILL slots are claimed the
// table is closed - later keys go to the global counters directly without a search (a wave runs a probe loop as long as
// its slowest lane, for every one of the eight key positions; measured: searching the closed table for keys that are
// in it costs the headline 6 %, closing it earlier (64 / 32 slots) 7 / 16 %, 512 slots change nothing).
#ifndef HS_FILL_DEF
#define HS_FILL_DEF (HS / 2)
#endif
#ifndef HS_PROBES_DEF
#define HS_PROBES_DEF 8
#endif
#ifndef HS_CLOSED_SKIP_DEF
#define HS_CLOSED_SKIP_DEF 1
#endif
constexpr int HS_FILL = HS_FILL_DEF, HS_PROBES = HS_PROBES_DEF;

// `open`: the table still takes new keys (hash_open, read ONCE per point: eight LDS round trips per thread otherwise)
__device__ __forceinline__ bool hash_open(const int *hfill) { return *(volatile const int *)hfill < HS_FILL; }

__device__ __forceinline__ int hash_insert(uint32_t *hkey, int *hcnt, int *hfill, uint32_t key, int &rank, bool open, int inc = 1)
{
    uint32_t h = (key * 2654435761u) >> (32 - HS_BITS);
#if HS_CLOSED_SKIP_DEF
    if (!open) return -1;          // a closed table is not even searched: the block's keys are (nearly) all different
#endif
    for (int probe = 0; probe < HS_PROBES; ++probe) {
        uint32_t cur = *(volatile uint32_t *)&hkey[h];
        if (cur == EMPTY) {
            if (!open) return -1;
            cur = atomicCAS(&hkey[h], EMPTY, key);
            if (cur == EMPTY) { atomicAdd(hfill, 1); cur = key; }
        }
        if (cur == key) { rank = atomicAdd(&hcnt[h], inc); return (int)h; }
        h = (h + 1) & (HS - 1);
    }
    return -1;
}

// The 16-byte record of a binned point: 10 + 10 + 10 voxel index bits and the three ratios.  The group
// (frame) rides in the two top bits of the four words: a ratio in [0, 1] has its sign and top exponent bit
// clear (MAX_GROUPS = 256 = 8 bits).
__device__ __forceinline__ uint4 make_record(const Point &pt)
{
    uint4 r;
    r.x = (uint32_t)pt.k0 | ((uint32_t)pt.k1 << 10) | ((uint32_t)pt.k2 << 20);
    r.y = __float_as_uint(pt.r0); r.z = __float_as_uint(pt.r1); r.w = __float_as_uint(pt.r2);
    const uint32_t g = (uint32_t)pt.group;
    r.x |= (g & 3u) << 30; r.y |= ((g >> 2) & 3u) << 30; r.z |= ((g >> 4) & 3u) << 30; r.w |= ((g >> 6) & 3u) << 30;
    return r;
}

// ----------------------------------------------------------------------------
// tile-local records ("meta" format) of the all-integer tile kernels
// ----------------------------------------------------------------------------
// A record is written per (point, tile) pair anyway, so for the calls that go to fuse_dense_kernel /
// fuse_cells_kernel (sequential or merged frames of class ids / ones, front end 0) scatter_kernel stores what the
// tile kernels would otherwise derive again in every pass, in the same 16 bytes and with no separate class-id word:
//   x  bits  0..7   corner c = 4 ca + 2 cb + cd lies inside the tile (ca / cb / cd pick the upper corner of axis 0 / 1 / 2)
//      bits  8..10  d0 d1 d2: upper minus lower voxel index per axis (0 where the footprint is clamped at the map border)
//      bits 11..20  tile-local id of the all-lower corner + META_VOFF (it may lie one voxel before the tile)
//      bits 21..28  class id (255 = outside [0, C), counts as an all-zero feature row), bits 29..30 frame bits 6..7
//   y, z, w  the three ratios (in [0, 1]: sign and top exponent bit clear); their two top bits carry frame bits 0..5.
constexpr int META_VOFF = 128;

struct AxisLocal { int l, d; bool in_lo, in_hi; };
__device__ __forceinline__ AxisLocal axis_local(const AxisFoot &a, int origin, int shift)
{
    AxisLocal x;
    x.l = a.lo - origin;
    x.d = a.hi - a.lo;
    x.in_lo = ((unsigned)x.l >> shift) == 0u;
    x.in_hi = ((unsigned)(a.hi - origin) >> shift) == 0u;
    return x;
}

__device__ __forceinline__ uint4 make_meta_record(const Point &pt, const AxisFoot &a0, const AxisFoot &a1, const AxisFoot &a2,
                                                  int o0, int o1, int o2, int s0, int s1, int s2, uint32_t label)
{
    const AxisLocal x0 = axis_local(a0, o0, s0), x1 = axis_local(a1, o1, s1), x2 = axis_local(a2, o2, s2);
    const uint32_t in8 = ((x0.in_lo ? 0x0fu : 0u) | (x0.in_hi ? 0xf0u : 0u)) & ((x1.in_lo ? 0x33u : 0u) | (x1.in_hi ? 0xccu : 0u)) &
                         ((x2.in_lo ? 0x55u : 0u) | (x2.in_hi ? 0xaau : 0u));
    const int v000 = x0.l * (1 << (s1 + s2)) + x1.l * (1 << s2) + x2.l;
    const uint32_t g = (uint32_t)pt.group;
    uint4 r;
    r.x = in8 | ((uint32_t)x0.d << 8) | ((uint32_t)x1.d << 9) | ((uint32_t)x2.d << 10) | ((uint32_t)(v000 + META_VOFF) << 11) |
          ((label > 255u ? 255u : label) << 21) | (((g >> 6) & 3u) << 29);
    r.y = __float_as_uint(pt.r0) | ((g & 3u) << 30);
    r.z = __float_as_uint(pt.r1) | (((g >> 2) & 3u) << 30);
    r.w = __float_as_uint(pt.r2) | (((g >> 4) & 3u) << 30);
    return r;
}

__device__ __forceinline__ int meta_frame(const uint4 &r)
{
    return (int)((r.y >> 30) | ((r.z >> 30) << 2) | ((r.w >> 30) << 4) | (((r.x >> 29) & 3u) << 6));
}
__device__ __forceinline__ uint32_t meta_label(const uint4 &r) { return (r.x >> 21) & 255u; }

// What a pass needs of a meta record: tile-local ids and weights of the eight corners (static register picks once
// the corner loop is unrolled), the inside mask.
template <int S1, int S2>
struct MetaCorners {
    int v[8];
    float w[8];
    uint32_t in8;
    __device__ __forceinline__ explicit MetaCorners(const uint4 &r)
    {
        in8 = r.x & 255u;
        const int v000 = (int)((r.x >> 11) & 1023u) - META_VOFF;
        const int e0 = (int)((r.x >> 8) & 1u) << (S1 + S2), e1 = (int)((r.x >> 9) & 1u) << S2, e2 = (int)((r.x >> 10) & 1u);
        v[0] = v000; v[1] = v000 + e2; v[2] = v000 + e1; v[3] = v[2] + e2;
        v[4] = v000 + e0; v[5] = v[4] + e2; v[6] = v[4] + e1; v[7] = v[6] + e2;
        const float r0 = __uint_as_float(r.y & 0x3fffffffu), r1 = __uint_as_float(r.z & 0x3fffffffu), r2 = __uint_as_float(r.w & 0x3fffffffu);
        // per axis (projection.py:280-316): r < 0.5: (0.5 - r, r + 0.5), else (1.5 - r, r - 0.5)
        const float l0 = (r0 < 0.5f ? 0.5f : 1.5f) - r0, h0 = r0 + (r0 < 0.5f ? 0.5f : -0.5f);
        const float l1 = (r1 < 0.5f ? 0.5f : 1.5f) - r1, h1 = r1 + (r1 < 0.5f ? 0.5f : -0.5f);
        const float l2 = (r2 < 0.5f ? 0.5f : 1.5f) - r2, h2 = r2 + (r2 < 0.5f ? 0.5f : -0.5f);
        // (w0 * w1) * w2 + 1e-9, the reference's product order (projection.py:319-323)
        const float w00 = l0 * l1, w01 = l0 * h1, w10 = h0 * l1, w11 = h0 * h1;
        w[0] = 1e-9f + w00 * l2; w[1] = 1e-9f + w00 * h2; w[2] = 1e-9f + w01 * l2; w[3] = 1e-9f + w01 * h2;
        w[4] = 1e-9f + w10 * l2; w[5] = 1e-9f + w10 * h2; w[6] = 1e-9f + w11 * l2; w[7] = 1e-9f + w11 * h2;
    }
};

// body(cc, v, w) for the corners of a meta record that lie inside the tile
template <int S1, int S2, class F>
__device__ __forceinline__ void meta_corners_idx(const uint4 &r, F body)
{
    const MetaCorners<S1, S2> m(r);
#pragma unroll
    for (int cc = 0; cc < 8; ++cc)
        if (m.in8 & (1u << cc)) body(cc, m.v[cc], m.w[cc]);
}

// ----------------------------------------------------------------------------
// tile-local entries ("meta" format) of the all-integer tile kernels: CONTRIBUTIONS
// ----------------------------------------------------------------------------
// For the calls that go to fuse_dense_kernel / fuse_cells_kernel (sequential or merged frames of class ids / ones,
// front end 0) scatter_kernel expands a point where its geometry is at hand anyway: one 8-byte contribution per
// corner of the 2 x 2 x 2 footprint, written to the bucket of the tile the corner lies in (rounds 2-3 wrote one
// 16-byte record per (point, tile) and the tile kernels worked the eight corners out again in each of their passes):
//   x  bits  0..6   voxel inside the tile (4 x 4 x 8: l0 * 32 + l1 * 8 + l2)
//      bits  7..14  class id (255 = outside [0, C), counts as an all-zero feature row)
//      bits 15..22  frame (sequential group)
//   y  the corner weight 1e-9 + (w0 * w1) * w2 (projection.py:319-323), fp32
__device__ __forceinline__ uint2 make_contribution(int v, uint32_t label, int frame, float w)
{
    return make_uint2((uint32_t)v | ((label > 255u ? 255u : label) << 7) | ((uint32_t)frame << 15), __float_as_uint(w));
}

// The sample behind use_contributions: block (i, f) takes patch i of `n_probe` evenly spread 16 x 16 patches of frame f and
// counts its valid points and the distinct tiles their voxels lie in (an LDS set).
__global__ __launch_bounds__(BIN_THREADS) void probe_kernel(FuseParams P, int n_probe)
{
    constexpr int PS = 512;
    __shared__ uint32_t pset[PS];
    __shared__ int n_valid, n_distinct;
    for (int s = threadIdx.x; s < PS; s += BIN_THREADS) pset[s] = EMPTY;
    if (threadIdx.x == 0) { n_valid = 0; n_distinct = 0; }
    __syncthreads();
    const int pw = (P.W + PATCH - 1) / PATCH, ph = (P.H + PATCH - 1) / PATCH, np = pw * ph;
    const int patch = (int)(((long long)blockIdx.x * np + np / 2) / n_probe) % np;
    const int py = patch / pw, px = patch - py * pw;
    const int y = py * PATCH + (int)(threadIdx.x / PATCH), x = px * PATCH + (int)(threadIdx.x % PATCH), f = blockIdx.y;
    if (y < P.H && x < P.W) {
        const int pix = y * P.W + x;
        const float d = P.depth[(long long)f * (P.H * P.W) + pix];
        float pose[12];
#pragma unroll
        for (int k = 0; k < 12; ++k) pose[k] = P.pose_inline ? P.pose0[k] : P.poses[f * 12 + k];
        float q0, q1, q2;
        rotate_ray(pose + 3, P.cam[pix * 3], P.cam[pix * 3 + 1], P.cam[pix * 3 + 2], q0, q1, q2);
        const float m0 = q0 * d, m1 = q1 * d, m2 = q2 * d;
        int kx, ky, kz; float rx, ry, rz;
        if (bin_point(P.bins, pose[0] + m0, pose[1] + m1, pose[2] + m2, d, P.min_d, P.max_d, kx, ky, kz, rx, ry, rz)) {
            const uint32_t key = (uint32_t)(((ky >> P.s0) * P.nt1 + (kx >> P.s1)) * P.nt2 + (kz >> P.s2));
            atomicAdd(&n_valid, 1);
            uint32_t h = (key * 2654435761u) >> (32 - 9);
            for (int probe = 0; probe < PS; ++probe) {
                const uint32_t cur = atomicCAS(&pset[h], EMPTY, key);
                if (cur == EMPTY) { atomicAdd(&n_distinct, 1); break; }
                if (cur == key) break;
                h = (h + 1) & (PS - 1);
            }
        }
    }
    __syncthreads();
    if (threadIdx.x == 0 && n_valid > 0) { atomicAdd(&P.ticket[PROBE_POINTS], n_valid); atomicAdd(&P.ticket[PROBE_TILES], n_distinct); }
}

template <int FRONT>
__global__ __launch_bounds__(BIN_THREADS) void count_kernel(FuseParams P)
{
    __shared__ uint32_t hkey[HS];
    __shared__ int hcnt[HS];
    __shared__ int hfill;
    if (FRONT == 0 && entry_format(P.ticket, P.meta, P.fmt_force) == FMT_AGG) return;      // bucket_agg_kernel<false> takes the call (uniform)
    for (int s = threadIdx.x; s < HS; s += BIN_THREADS) { hkey[s] = EMPTY; hcnt[s] = 0; }
    if (threadIdx.x == 0) hfill = 0;
    __syncthreads();
    const long long idx = point_index<FRONT>(P, BIN_THREADS);
    if (idx >= 0) {
        Point pt; uint32_t aux = 0;
        const bool ok = get_point<FRONT>(P, idx, pt, aux);
        if (P.label_status && P.feat_kind >= MF_FEAT_LABEL_U8 && P.feat_kind <= MF_FEAT_LABEL_I64 && aux >= (uint32_t)P.C) {
            *P.label_status = 1;                        // reported to the host ...
            P.ticket[ABORT_SLOT] = 1;                   // ... and the rest of the pipeline is called off
        }
        if (FRONT == 0) {
            for (int m = 0; m < P.n_extra; ++m) {       // class ids of the further maps: every pixel is looked at, as above
                const FuseParams::ExtraMap &E = P.extra[m];
                if (!E.label_status || E.feat_kind < MF_FEAT_LABEL_U8 || E.feat_kind > MF_FEAT_LABEL_I64) continue;
                const Pix px = patch_pixel(P);
                const long long fi = feature_pixel((int)blockIdx.y, px.y, px.x, E.fh, E.fw, E.rep_y, E.rep_x);
                if (read_label(E.feat, E.feat_kind, fi) >= (uint32_t)E.C) { *E.label_status = 1; *E.abort = 1; }
            }
        }
        if (FRONT == 0) {                               // scatter_kernel takes the binned pixel from here (no second unprojection)
            uint4 r = make_record(pt);
            if (!ok) r.y = 0xffffffffu;                // no ratio in [0, 1] has these low 30 bits
            P.pts[idx] = r;
        }
        if (ok) {
            TileKeys K;
            point_keys8(P, pt, K);
            // entries of a bucket: one record per (point, tile), or - tile-local "meta" entries - one contribution per corner
            // inside the tile: the eight corners split evenly over the point's tiles (an axis whose two corners straddle a
            // tile face halves the share)
            const int inc = use_contributions(P.ticket, P.meta, P.fmt_force) ? 8 >> __popc(K.straddle) : 1;
            const bool open = hash_open(&hfill);
#pragma unroll
            for (int j = 0; j < 8; ++j)
                if (K.mask & (1u << j)) {
                    int rank;
                    if (hash_insert(hkey, hcnt, &hfill, K.key[j], rank, open, inc) < 0) atomicAdd(&P.cursor[K.key[j]], inc);
                }
        }
    }
    if (FRONT == 0) {
        // the counters and split scratch of the further maps (their own memset otherwise)
        const unsigned gtid = (blockIdx.y * gridDim.x + blockIdx.x) * BIN_THREADS + threadIdx.x, gsize = gridDim.x * gridDim.y * BIN_THREADS;
        for (int m = 0; m < P.n_extra; ++m)
            for (unsigned i = gtid; i < P.extra[m].zero16; i += gsize) P.extra[m].zero[i] = make_uint4(0u, 0u, 0u, 0u);
        // max |feature| of the maps whose dense features take the single-pass kernel (its fixed-point scale): every
        // feature pixel is some pixel's (the feature image divides the frame), so the maximum over the pixels is exact
        for (int m = -1; m < P.n_extra; ++m) {
            unsigned *out = m < 0 ? P.absmax : P.extra[m].absmax;
            if (!out) continue;                                    // (uniform)
            const float *f = (const float *)(m < 0 ? P.feat : P.extra[m].feat);
            const int Cm = m < 0 ? P.C : P.extra[m].C, fh = m < 0 ? P.fh : P.extra[m].fh, fw = m < 0 ? P.fw : P.extra[m].fw;
            const int ry = m < 0 ? P.rep_y : P.extra[m].rep_y, rx = m < 0 ? P.rep_x : P.extra[m].rep_x;
            unsigned mx = 0u;
            if (idx >= 0) {
                const Pix px = patch_pixel(P);
                const float *row = f + feature_pixel((int)blockIdx.y, px.y, px.x, fh, fw, ry, rx) * Cm;
                for (int c = 0; c < Cm; ++c) mx = max(mx, __float_as_uint(row[c]) & 0x7fffffffu);
            }
            for (int o = 32; o > 0; o >>= 1) mx = max(mx, (unsigned)__shfl_down((int)mx, o, 64));
            if ((threadIdx.x & 63) == 0 && mx > *(volatile unsigned *)out) atomicMax(out, mx);
        }
    }
    __syncthreads();
    for (int s = threadIdx.x; s < HS; s += BIN_THREADS)
        if (hkey[s] != EMPTY) atomicAdd(&P.cursor[hkey[s]], hcnt[s]);
}

template <int FRONT>
__global__ __launch_bounds__(BIN_THREADS) void scatter_kernel(FuseParams P)
{
    __shared__ uint32_t hkey[HS];
    __shared__ int hcnt[HS];
    __shared__ int hfill;
    if (FRONT == 0 && entry_format(P.ticket, P.meta, P.fmt_force) == FMT_AGG) return;      // bucket_agg_kernel<true> takes the call (uniform)
    for (int s = threadIdx.x; s < HS; s += BIN_THREADS) { hkey[s] = EMPTY; hcnt[s] = 0; }
    if (threadIdx.x == 0) hfill = 0;
    __syncthreads();
    const long long idx = point_index<FRONT>(P, BIN_THREADS);
    Point pt; uint32_t aux = 0;
    uint32_t xaux[MAX_EXTRA_MAPS] = {0u, 0u, 0u};
    TileKeys K;
    K.mask = 0u;
    int slot[8], rank[8];
    uint4 r = make_uint4(0u, 0xffffffffu, 0u, 0u);
    bool ok = false;
    if (FRONT == 0) {
        if (idx >= 0) r = P.pts[idx];
        ok = (r.y & 0x3fffffffu) != 0x3fffffffu;
        if (ok) {
            const unsigned rm = 0x3fffffffu;
            pt.k0 = r.x & 1023; pt.k1 = (r.x >> 10) & 1023; pt.k2 = (r.x >> 20) & 1023;
            pt.r0 = __uint_as_float(r.y & rm); pt.r1 = __uint_as_float(r.z & rm); pt.r2 = __uint_as_float(r.w & rm);
            pt.group = P.G == 1 ? 0 : (int)blockIdx.y;
            const Pix px = patch_pixel(P);
            if (P.feat_kind != MF_FEAT_ONES) {
                const long long fi = feature_pixel((int)blockIdx.y, px.y, px.x, P.fh, P.fw, P.rep_y, P.rep_x);
                aux = P.feat_kind == MF_FEAT_DENSE_F32 ? (uint32_t)fi : read_label(P.feat, P.feat_kind, fi);
            }
            for (int m = 0; m < P.n_extra; ++m) {
                const FuseParams::ExtraMap &E = P.extra[m];
                if (!E.aux) continue;
                const long long fi = feature_pixel((int)blockIdx.y, px.y, px.x, E.fh, E.fw, E.rep_y, E.rep_x);
                xaux[m] = E.feat_kind == MF_FEAT_DENSE_F32 ? (uint32_t)fi : read_label(E.feat, E.feat_kind, fi);
            }
        }
    } else {
        ok = idx >= 0 && get_point<FRONT>(P, idx, pt, aux);
        if (ok) r = make_record(pt);
    }
    if (ok) {
        point_keys8(P, pt, K);
    }
    const bool contrib = use_contributions(P.ticket, P.meta, P.fmt_force);       // (uniform)
    const int inc_k = ok && contrib ? 8 >> __popc(K.straddle) : 1;
    if (ok) {
        const bool open = hash_open(&hfill);
#pragma unroll
        for (int j = 0; j < 8; ++j)
            if (K.mask & (1u << j)) slot[j] = hash_insert(hkey, hcnt, &hfill, K.key[j], rank[j], open, inc_k);
    }
    __syncthreads();
    // one returning global atomic per distinct bucket of this block; hcnt becomes the base
    for (int s = threadIdx.x; s < HS; s += BIN_THREADS)
        if (hkey[s] != EMPTY) hcnt[s] = atomicAdd(&P.cursor[hkey[s]], hcnt[s]);
    __syncthreads();
    if (K.mask) {
        // positions first (the returning atomics of the buckets that are not in the table all in flight), then the stores
        int pos[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            pos[j] = 0;
            if (K.mask & (1u << j)) pos[j] = slot[j] >= 0 ? hcnt[slot[j]] + rank[j] : atomicAdd(&P.cursor[K.key[j]], inc_k);
        }
        if (contrib) {
            // Tile-local entries for fuse_cells_kernel: one CONTRIBUTION per corner, written to the bucket of the
            // corner's own tile (make_contribution: voxel inside the tile, class id, frame, corner weight).  Corner c = 4 ca +
            // 2 cb + cd belongs to key position c & straddle; inside a (point, tile) share the corners are numbered by
            // their free bits.  A footprint clamped at the map border has two corners on one voxel: both are written,
            // like the reference scatters both (projection.py:294-298).
            const unsigned sd = K.straddle;
            const bool m0 = sd & 4u, m1 = sd & 2u, m2 = sd & 1u;
            const int s2n = m2 ? 1 : 2, u1 = m1 ? 0 : s2n, u0 = m0 ? 0 : s2n * (m1 ? 1 : 2);     // (the z pair of a share is adjacent)
            const float w00 = K.a0.wlo * K.a1.wlo, w01 = K.a0.wlo * K.a1.whi, w10 = K.a0.whi * K.a1.wlo, w11 = K.a0.whi * K.a1.whi;
            const int x0[2] = {(K.a0.lo & ((1 << P.s0) - 1)) << (P.s1 + P.s2), (K.a0.hi & ((1 << P.s0) - 1)) << (P.s1 + P.s2)};
            const int x1[2] = {(K.a1.lo & ((1 << P.s1) - 1)) << P.s2, (K.a1.hi & ((1 << P.s1) - 1)) << P.s2};
            const int x2[2] = {K.a2.lo & ((1 << P.s2) - 1), K.a2.hi & ((1 << P.s2) - 1)};
            uint2 *out = reinterpret_cast<uint2 *>(P.rec);
#pragma unroll
            for (int cab = 0; cab < 4; ++cab) {
                const int ca = cab >> 1, cb = cab & 1;
                // position of the share: pos[c & straddle], picked with static selects
                const bool ja = ca && m0, jb = cb && m1;
                const int base_lo = ja ? (jb ? pos[6] : pos[4]) : (jb ? pos[2] : pos[0]);        // corner cd = 0 (and cd = 1 when z does not straddle)
                const int base_hi = ja ? (jb ? pos[7] : pos[5]) : (jb ? pos[3] : pos[1]);        // corner cd = 1 when z straddles
                const int idx = ca * u0 + cb * u1;
                // (w0 * w1) * w2 + 1e-9, the reference's product order (projection.py:319-323)
                const float wab = ca ? (cb ? w11 : w10) : (cb ? w01 : w00);
                const uint2 lo = make_contribution(x0[ca] | x1[cb] | x2[0], aux, pt.group, 1e-9f + wab * K.a2.wlo);
                const uint2 hi = make_contribution(x0[ca] | x1[cb] | x2[1], aux, pt.group, 1e-9f + wab * K.a2.whi);
                // The z pair of a share is adjacent: ONE 16-byte store unless z straddles a tile face (one time in eight).
                // The write requests that reach the L2, not the instructions, bound this kernel: 8-byte stores of a wave's
                // 64 lanes are 64 requests each.
                if (!m2) {
                    uint4 q; q.x = lo.x; q.y = lo.y; q.z = hi.x; q.w = hi.y;
                    __builtin_memcpy(out + base_lo + idx, &q, 16);       // (8-byte aligned)
                } else {
                    out[base_lo + idx] = lo;
                    out[base_hi + idx] = hi;
                }
            }
        } else if (P.meta) {
            // tile-local records for fuse_dense_kernel (no separate class-id word)
#pragma unroll
            for (int j = 0; j < 8; ++j)
                if (K.mask & (1u << j)) {
                    const int a = j >> 2, b = (j >> 1) & 1, c = j & 1;
                    P.rec[pos[j]] = make_meta_record(pt, K.a0, K.a1, K.a2, K.t0[a] << P.s0, K.t1[b] << P.s1, K.t2[c] << P.s2,
                                                     P.s0, P.s1, P.s2, aux);
                }
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j)
                if (K.mask & (1u << j)) {
                    P.rec[pos[j]] = r;
                    if (P.feat_kind != MF_FEAT_ONES) P.aux[pos[j]] = aux;
                    if (FRONT == 0)
                        for (int m = 0; m < P.n_extra; ++m)
                            if (P.extra[m].aux) P.extra[m].aux[pos[j]] = xaux[m];
                }
        }
    }
}

// w * 2^shift as a 64-bit integer (truncated), for 0 <= w < 2^(40 - shift + ...): built from the
// float's bits with one 64-bit shift instead of the seven-instruction float -> u64 conversion.
// fx_c = 182 - shift; a result below one unit (or w == 0) comes out as 0.
__device__ __forceinline__ unsigned long long to_fixed(float w, int fx_c)
{
    const unsigned b = __float_as_uint(w);
    const unsigned m = (b & 0x7fffffu) | 0x800000u;             // 24-bit significand
    int amt = fx_c - (int)(b >> 23);                            // (m << 32) >> amt == m * 2^(e - 150 + shift)
    amt = amt > 63 ? 63 : amt;                                  // m << 32 < 2^56, so 63 yields 0
    return ((unsigned long long)m << 32) >> amt;
}

// ----------------------------------------------------------------------------
// AGGREGATED entries (round 4): collision compaction for real scenes
// ----------------------------------------------------------------------------
// In a real scene the 256 points of a 16 x 16 pixel patch land on a few dozen voxels: 30-140 corners of a frame on
// the same (voxel, class).  The all-integer tile kernels only need, per (voxel, class, frame), the SUMS W = sum w and
// S2 = sum w^2 of those corners (W and S2 of the voxel's cell are sums over the classes, the delta of (voxel, class) is
// t_f * S2).  So for calls the probe finds to be a real scene the two bucketing kernels aggregate a block's 2,048
// corners per (tile, frame, voxel, class) in LDS before anything is written, and the tile kernel (fuse_cells_kernel<AGG>)
// reads one 16-byte entry per group instead of one 16-byte record per (point, tile):
//   x  bits  0..6 voxel inside the tile, 7..14 class id (255 = outside [0, C)), 15..22 frame, 23..31 W bits 32..40
//   y  W bits 0..31      z, w  S2 (64 bits)
// W is the exact integer sum of to_fixed(w) with 32 fraction bits, S2 that of to_fixed(w * w) with 55 (a block puts at
// most 256 x 1.0 on one voxel: below 2^41 / 2^63), so the entries - and with them the map - are run-to-run identical.
// S2 needs its 55 bits: a voxel that a frame touches with ONE far corner of weight w ~ 1e-7 receives iw * w^2 / W = iw * w,
// which is not small against the map's tolerance although w^2 ~ 1e-14 is (with 40 fraction bits such corners were lost:
// measured, 3e-6 off on the border voxels of a room); W only enters as S2 / W and 1 / W, where 2^-32 absolute is plenty.
// count (SCATTER = false) and scatter (SCATTER = true) must agree on the number of entries of every bucket, whatever
// the order their threads run in: the table is DIRECT MAPPED and a slot goes to the SMALLEST key that hashes to it
// (atomicMin, order independent).  Corners whose key won their slot are summed there and leave as ONE entry; a
// corner that lost is written as an entry of its own.  Both kernels see the same keys, hence the same winners.
constexpr int AGG_BITS = 10, AGG_SLOTS = 1 << AGG_BITS;
constexpr int AGG2_BITS = 8, AGG2_SLOTS = 1 << AGG2_BITS;           // the second table, for the keys that lost in the first
#ifndef AGG_ABL
#define AGG_ABL 0
#endif
#ifndef AGG_ITEMS_DEF
#define AGG_ITEMS_DEF 4
#endif
constexpr int AGG_ITEMS = AGG_ITEMS_DEF;            // (patch, frame) items per workgroup
constexpr int AGG_FW = 32, AGG_FS = 55;            // fraction bits of an entry's W / S2
#ifndef AGG_DEFAULT
#define AGG_DEFAULT 1
#endif

__device__ __forceinline__ uint4 make_agg_entry(unsigned hdr, unsigned long long W, unsigned long long S)
{
    uint4 e;
    e.x = hdr | ((unsigned)(W >> 32) << 23);
    e.y = (unsigned)W;
    e.z = (unsigned)S;
    e.w = (unsigned)(S >> 32);
    return e;
}
__device__ __forceinline__ unsigned long long agg_W(const uint4 &e) { return (unsigned long long)e.y | ((unsigned long long)(e.x >> 23) << 32); }
__device__ __forceinline__ unsigned long long agg_S(const uint4 &e) { return (unsigned long long)e.z | ((unsigned long long)e.w << 32); }

// position i of lane l <- position i ^ r: neighbouring pixels of a real scene have the SAME footprint, and in corner order
// the lanes of one LDS atomic would all hit the same few words (same-address LDS atomics are served one lane at a time:
// measured, 1.0 ms of the first version's 1.0 ms).  With the corners rotated per lane, lanes with the same footprint add
// DIFFERENT corners in the same instruction.
template <class T>
__device__ __forceinline__ void xor_permute8(T (&a)[8], unsigned r)
{
#pragma unroll
    for (int b = 1; b < 8; b <<= 1) {
        const bool sw = r & b;
#pragma unroll
        for (int i = 0; i < 8; ++i)
            if (!(i & b)) { const T x = a[i], y = a[i | b]; a[i] = sw ? y : x; a[i | b] = sw ? x : y; }
    }
}

template <bool SCATTER>
__global__ __launch_bounds__(BIN_THREADS) void bucket_agg_kernel(FuseParams P)
{
    __shared__ uint32_t akey[AGG_SLOTS], bkey[AGG2_SLOTS];
    __shared__ unsigned long long aW[SCATTER ? AGG_SLOTS : 1], aS[SCATTER ? AGG_SLOTS : 1], bW[SCATTER ? AGG2_SLOTS : 1], bS[SCATTER ? AGG2_SLOTS : 1];
    __shared__ uint32_t hkey[HS];
    __shared__ int hcnt[HS];
    __shared__ int hfill;
    if (entry_format(P.ticket, P.meta, P.fmt_force) != FMT_AGG) return;        // count_kernel / scatter_kernel take the call (uniform)
    // Work items = (16 x 16 pixel patch, frame) pairs, AGG_ITEMS per workgroup (grid stride): launched for every call that
    // MAY be a real scene, the kernel costs a call that is not (the headline) a quarter of the 77 k workgroups that return
    // at once.  (All items in a grid-stride loop of six workgroups per CU: room batch 1.07 -> 1.33 ms per step - the loop's
    // back edge waits for the item's stores and atomics, which separate workgroups overlap.)
    const int pw = (P.W + PATCH - 1) / PATCH, n_patches = pw * ((P.H + PATCH - 1) / PATCH), n_items = n_patches * P.n_frames;
    for (int item = blockIdx.x; item < n_items; item += gridDim.x) {
    const int by = item / n_patches, bx = item - by * n_patches;               // (scalar)
    for (int s = threadIdx.x; s < AGG_SLOTS; s += BIN_THREADS) {
        akey[s] = EMPTY;
        if (SCATTER) { aW[s] = 0ull; aS[s] = 0ull; }
    }
    for (int s = threadIdx.x; s < AGG2_SLOTS; s += BIN_THREADS) {
        bkey[s] = EMPTY;
        if (SCATTER) { bW[s] = 0ull; bS[s] = 0ull; }
    }
    for (int s = threadIdx.x; s < HS; s += BIN_THREADS) { hkey[s] = EMPTY; hcnt[s] = 0; }
    if (threadIdx.x == 0) hfill = 0;
    __syncthreads();
    const Pix pix = patch_pixel_at(P, bx);
    const long long idx = (pix.y < P.H && pix.x < P.W) ? (long long)by * (P.H * P.W) + pix.y * P.W + pix.x : -1;
    const int group = P.G == 1 ? 0 : by;
    Point pt;
    bool ok = false;
    if (!SCATTER) {
        if (idx >= 0) {
            uint32_t aux = 0;
            ok = get_point_at<0>(P, idx, pt, aux, bx, by);
            if (P.label_status && P.feat_kind >= MF_FEAT_LABEL_U8 && P.feat_kind <= MF_FEAT_LABEL_I64 && aux >= (uint32_t)P.C) {
                *P.label_status = 1;                    // (as in count_kernel)
                P.ticket[ABORT_SLOT] = 1;
            }
            uint4 r = make_record(pt);
            if (!ok) r.y = 0xffffffffu;
            P.pts[idx] = r;
        }
    } else {
        uint4 r = make_uint4(0u, 0xffffffffu, 0u, 0u);
        if (idx >= 0) r = P.pts[idx];
        ok = (r.y & 0x3fffffffu) != 0x3fffffffu;
        if (ok) {
            const unsigned rm = 0x3fffffffu;
            pt.k0 = r.x & 1023; pt.k1 = (r.x >> 10) & 1023; pt.k2 = (r.x >> 20) & 1023;
            pt.r0 = __uint_as_float(r.y & rm); pt.r1 = __uint_as_float(r.z & rm); pt.r2 = __uint_as_float(r.w & rm);
        }
    }
    // The eight corners.  key = tile << 15 | class << 7 | voxel inside the tile (the frame is the block's: 32 bits do for
    // maps of up to 2^17 tiles, the host offers the format to no other; a class id outside [0, C) is kept as C <= 64, so no
    // key is all ones = EMPTY).
    uint32_t ck[8];
    int cs[8];
    float cw[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) { ck[c] = EMPTY; cs[c] = 0; cw[c] = 0.0f; }
    if (ok) {
        uint32_t label = 0u;
        if (P.feat_kind != MF_FEAT_ONES) {
            label = read_label(P.feat, P.feat_kind, feature_pixel(by, pix.y, pix.x, P.fh, P.fw, P.rep_y, P.rep_x));
        }
        const uint32_t cls = label > (uint32_t)P.C ? (uint32_t)P.C : label;
        const AxisFoot a0 = axis_foot(pt.k0, pt.r0, P.size0), a1 = axis_foot(pt.k1, pt.r1, P.size1), a2 = axis_foot(pt.k2, pt.r2, P.size2);
        const float w0[2] = {a0.wlo, a0.whi}, w1[2] = {a1.wlo, a1.whi}, w2[2] = {a2.wlo, a2.whi};
        // per axis: the corner's tile coordinate (pre-multiplied) and its voxel bits inside the tile (4 x 4 x 8 tiles)
        const uint32_t t0[2] = {(uint32_t)(a0.lo >> 2) * (uint32_t)(P.nt1 * P.nt2), (uint32_t)(a0.hi >> 2) * (uint32_t)(P.nt1 * P.nt2)};
        const uint32_t t1[2] = {(uint32_t)(a1.lo >> 2) * (uint32_t)P.nt2, (uint32_t)(a1.hi >> 2) * (uint32_t)P.nt2};
        const uint32_t t2[2] = {(uint32_t)(a2.lo >> 3), (uint32_t)(a2.hi >> 3)};
        const uint32_t x0[2] = {(uint32_t)(a0.lo & 3) << 5, (uint32_t)(a0.hi & 3) << 5};
        const uint32_t x1[2] = {(uint32_t)(a1.lo & 3) << 3, (uint32_t)(a1.hi & 3) << 3};
        const uint32_t x2[2] = {(uint32_t)(a2.lo & 7), (uint32_t)(a2.hi & 7)};
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            const int ca = c >> 2, cb = (c >> 1) & 1, cd = c & 1;
            // (w0 * w1) * w2 + 1e-9, the reference's product order (projection.py:319-323)
            cw[c] = 1e-9f + (w0[ca] * w1[cb]) * w2[cd];
            const uint32_t tile = t0[ca] + t1[cb] + t2[cd], v = x0[ca] | x1[cb] | x2[cd];
            ck[c] = (tile << 15) | (cls << 7) | v;
        }
    }
    {
        const unsigned lane = threadIdx.x & 63u, rot = (lane ^ (lane >> 3)) & 7u;
        xor_permute8(ck, rot);
        if (SCATTER) xor_permute8(cw, rot);
    }
    // (a plain multiplicative hash: slots made of the voxel bits and a few hash bits of (tile, class) were measured too - the
    // tiles along a wall use the same voxels of their tiles, and when two of them share the hash bits one loses everything)
#pragma unroll
    for (int c = 0; c < 8; ++c) cs[c] = (int)((ck[c] * 2654435761u) >> (32 - AGG_BITS));
    // a slot goes to the smallest key that hashes to it (a slot that already holds a key this small is left alone: plain
    // reads of one word are broadcast, atomics on it are not)
    {
        uint32_t cur[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) cur[c] = *(volatile uint32_t *)&akey[cs[c]];
#pragma unroll
        for (int c = 0; c < 8; ++c)
            if (AGG_ABL < 3 && ck[c] < cur[c]) atomicMin(&akey[cs[c]], ck[c]);
    }
    __syncthreads();
    // A corner whose key holds its slot is summed there.  One that lost (~5 % of a real scene's corners: ~100 distinct keys
    // on 1,024 slots) tries a second, small table the same way; what loses there too is an entry of its own, its place in
    // the bucket from the global counter directly (rare: without the second table these atomics - all corners of a losing
    // key on the same counter - were a third of both kernels' time).
    unsigned lose = 0u;
    {
        uint32_t cur[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) cur[c] = akey[cs[c]];
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            if (ck[c] == EMPTY) continue;
            if (cur[c] == ck[c]) {
                if (SCATTER && AGG_ABL < 2) {
                    atomicAdd(&aW[cs[c]], to_fixed(cw[c], 182 - AGG_FW));
                    atomicAdd(&aS[cs[c]], to_fixed(cw[c] * cw[c], 182 - AGG_FS));
                }
            } else {
                lose |= 1u << c;
                atomicMin(&bkey[(ck[c] * 0x85ebca6bu) >> (32 - AGG2_BITS)], ck[c]);
            }
        }
    }
    __syncthreads();
    const unsigned fbits = (unsigned)group << 15;
    if (lose && AGG_ABL < 1) {
#pragma unroll
        for (int c = 0; c < 8; ++c)
            if (lose & (1u << c)) {
                const int s2 = (int)((ck[c] * 0x85ebca6bu) >> (32 - AGG2_BITS));
                if (bkey[s2] == ck[c]) {
                    if (SCATTER) {
                        atomicAdd(&bW[s2], to_fixed(cw[c], 182 - AGG_FW));
                        atomicAdd(&bS[s2], to_fixed(cw[c] * cw[c], 182 - AGG_FS));
                    }
                } else {
                    const int pos = atomicAdd(&P.cursor[(ck[c] >> 15) * (uint32_t)P.G + (uint32_t)group], 1);
                    if (SCATTER)
                        P.rec[pos] = make_agg_entry((ck[c] & 0x7fffu) | fbits, to_fixed(cw[c], 182 - AGG_FW), to_fixed(cw[c] * cw[c], 182 - AGG_FS));
                }
            }
    }
    // one entry per claimed slot: the slots of a bucket are ranked through the block's table of (tile, frame) buckets
    constexpr int NS = (AGG_SLOTS + AGG2_SLOTS + BIN_THREADS - 1) / BIN_THREADS;
    auto slot_key = [&](int s) { return s < AGG_SLOTS ? akey[s] : s < AGG_SLOTS + AGG2_SLOTS ? bkey[s - AGG_SLOTS] : EMPTY; };
    int sslot[NS], srank[NS];
    {
        const bool open = hash_open(&hfill);
#pragma unroll
        for (int i = 0; i < NS; ++i) {
            const uint32_t k = slot_key(threadIdx.x + BIN_THREADS * i);
            sslot[i] = -1; srank[i] = 0;
            if (k != EMPTY) {
                const uint32_t tk = (k >> 15) * (uint32_t)P.G + (uint32_t)group;
                sslot[i] = hash_insert(hkey, hcnt, &hfill, tk, srank[i], open, 1);
                if (!SCATTER && sslot[i] < 0) atomicAdd(&P.cursor[tk], 1);
            }
        }
    }
    __syncthreads();
    if (!SCATTER) {
        for (int s = threadIdx.x; s < HS; s += BIN_THREADS)
            if (hkey[s] != EMPTY) atomicAdd(&P.cursor[hkey[s]], hcnt[s]);
    } else {
        for (int s = threadIdx.x; s < HS; s += BIN_THREADS)
            if (hkey[s] != EMPTY) hcnt[s] = atomicAdd(&P.cursor[hkey[s]], hcnt[s]);
        __syncthreads();
#pragma unroll
        for (int i = 0; i < NS; ++i) {
            const int s = threadIdx.x + BIN_THREADS * i;
            const uint32_t k = slot_key(s);
            if (k != EMPTY) {
                const int pos = sslot[i] >= 0 ? hcnt[sslot[i]] + srank[i] : atomicAdd(&P.cursor[(k >> 15) * (uint32_t)P.G + (uint32_t)group], 1);
                P.rec[pos] = s < AGG_SLOTS ? make_agg_entry((k & 0x7fffu) | fbits, aW[s], aS[s])
                                           : make_agg_entry((k & 0x7fffu) | fbits, bW[s - AGG_SLOTS], bS[s - AGG_SLOTS]);
            }
        }
    }
    __syncthreads();                                    // the next item clears the tables
    }
}

// ----------------------------------------------------------------------------
// exclusive scan of cursor[0 .. n] (n = n_keys + 1 items, last one is 0)
// ----------------------------------------------------------------------------
constexpr int SCAN_THREADS = 256;
constexpr int SCAN_ITEMS = 8;
constexpr int SCAN_TILE = SCAN_THREADS * SCAN_ITEMS;

__device__ __forceinline__ int block_reduce_sum(int v, int *sh /* [SCAN_THREADS/64] */)
{
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    int t = 0;
    for (int w = 0; w < SCAN_THREADS / 64; ++w) t += sh[w];
    __syncthreads();
    return t;
}

__global__ __launch_bounds__(SCAN_THREADS) void scan_sums_kernel(const int *__restrict__ data, int n, int *block_sums)
{
    __shared__ int sh[SCAN_THREADS / 64];
    const int base = blockIdx.x * SCAN_TILE + threadIdx.x * SCAN_ITEMS;
    int s = 0;
    for (int i = 0; i < SCAN_ITEMS; ++i) if (base + i < n) s += data[base + i];
    s = block_reduce_sum(s, sh);
    if (threadIdx.x == 0) block_sums[blockIdx.x] = s;
}

__global__ __launch_bounds__(SCAN_THREADS) void scan_apply_kernel(int *data, int n, const int *__restrict__ block_sums,
                                                                  int *nonempty /* += buckets with entries, or NULL */)
{
    __shared__ int sh[SCAN_THREADS / 64];
    __shared__ int wsum[SCAN_THREADS / 64];
    int pre = 0;
    for (int b = threadIdx.x; b < (int)blockIdx.x; b += SCAN_THREADS) pre += block_sums[b];
    pre = block_reduce_sum(pre, sh);
    const int base = blockIdx.x * SCAN_TILE + threadIdx.x * SCAN_ITEMS;
    int v[SCAN_ITEMS];
    int s = 0;
    for (int i = 0; i < SCAN_ITEMS; ++i) { v[i] = base + i < n ? data[base + i] : 0; s += v[i]; }
    // inclusive scan of the per-thread sums inside the wave, then across waves
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    int inc = s;
    for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(inc, o, 64); if (lane >= o) inc += t; }
    if (lane == 63) wsum[wid] = inc;
    __syncthreads();
    int woff = 0;
    for (int w = 0; w < wid; ++w) woff += wsum[w];
    int run = pre + woff + inc - s;
    int filled = 0;
    for (int i = 0; i < SCAN_ITEMS; ++i) if (base + i < n) { data[base + i] = run; run += v[i]; filled += v[i] > 0; }
    if (nonempty) {                                     // one atomic per block (all of them land on one word)
        for (int o = 32; o > 0; o >>= 1) filled += __shfl_down(filled, o, 64);
        __syncthreads();
        if (lane == 0) wsum[wid] = filled;
        __syncthreads();
        if (threadIdx.x == 0) {
            int t = 0;
            for (int w = 0; w < SCAN_THREADS / 64; ++w) t += wsum[w];
            if (t) atomicAdd(nonempty, t);
        }
    }
}

// ----------------------------------------------------------------------------
// work list: the non-empty tiles, heaviest class first
// ----------------------------------------------------------------------------
// ticket[0] = work counter of the tile kernel, ticket[1 + c] = tiles in class c.
// Class 0 holds the tiles with the most entries; the tile kernel walks class 0,
// 1, 2, 3 in that order so the long tiles start first and the tail is short.

__device__ __forceinline__ int tile_class(int n)
{
    return n >= 131072 ? 0 : n >= 65536 ? 1 : n >= 32768 ? 2 : n >= 16384 ? 3 : n >= 8192 ? 4 : n >= 2048 ? 5 : n >= 512 ? 6 : 7;
}

// Work items of fuse_single_kernel (single-group calls with class-id / ones features, split_min > 0):
// every non-empty tile, one with more than split_min records cut into `nparts` record ranges.
constexpr int SPLIT_ITEMS = ABORT_SLOT + 1, SPLIT_TILES = ABORT_SLOT + 2, SPLIT_NONEMPTY = ABORT_SLOT + 3;
constexpr int FEAT_ABSMAX = ABORT_SLOT + 4;     // bits of max |feature| of the call (dense single-pass path)
constexpr int MODE_SLOT = ABORT_SLOT + 5;       // which tile kernel takes the call: 0 fuse_tiles_kernel, 2 fuse_dense_kernel, 3 fuse_cells_kernel
constexpr int HINT_SLOT = ABORT_SLOT + 6;       // [2] records listed, non-empty buckets x tile voxels / 2 (diagnostics)
constexpr int TICKET_DENSE = ABORT_SLOT + 8;    // work counter of fuse_dense_kernel (ticket[0] is fuse_tiles_kernel's)
constexpr int TICKET_CELLS = ABORT_SLOT + 9;    // work counter of fuse_cells_kernel
constexpr int ABORT_MAPS = 32;                  // [MAX_EXTRA_MAPS] abort words of the further maps of a mf_fuse_frame_maps call
constexpr int ABSMAX_MAPS = 36;                 // [MAX_EXTRA_MAPS] their FEAT_ABSMAX words
// fuse_cells_kernel deals its work list statically (no ticket: see there), which balances only if the tiles of a load
// class cost about the same: its list has half-octave classes of the entry count, heaviest first
constexpr int CELL_CLASSES = 40;                // entries >= 2^19.5 ... < 2
constexpr int CELL_COUNT = 64;                  // ticket[CELL_COUNT + c]: tiles in fine class c (the counter block has 128 words)
__device__ __forceinline__ int cell_class(int n)     // n >= 1
{
    const int l = 31 - __clz(n), key = 2 * l + (l > 0 ? (n >> (l - 1)) & 1 : 0);
    return max(0, CELL_CLASSES - 1 - key);
}
constexpr int MODE_TILES = 0, MODE_DENSE = 2, MODE_CELLS = 3, MODE_CELLS_AGG = 4;      // (4: fuse_cells_kernel over aggregated entries)
constexpr int SINGLE_DENSE_MAX_C = 16;          // dense features take the single-pass path up to this many channels
constexpr int SINGLE_MIN_MEAN = 96;       // class ids: mean records per non-empty tile below which a call stays with the tile kernel
constexpr int SPLIT_PARTS_MAX = 64;
constexpr long long SINGLE_MAX_POINTS = 1 << 21;   // calls with more points (a merged multi-frame batch) keep the tile kernel

// One list per map (blockIdx.y): the maps of a mf_fuse_frame_maps call share cursor and records, each has its counters,
// lists, thresholds (they depend on the feature kind) and abort word.
struct ListMap {
    int *ticket, *active, *items;
    int split_min, split_slots, min_mean, first_ticket, dense_tv, first_ticket_dense, first_ticket_cells;
    const int *abort;              // a class id of this map was out of range
    int4 *items4;                  // [CELL_CLASSES][n_tiles] work items of fuse_cells_kernel (in place of `active` when it takes the call)
};
struct ListParams {
    const int *cursor;             // exclusive offsets
    int n_tiles, G, split_part;
    int nt1, nt2, s0, s1, s2;      // tile grid (the origin of a tile rides in fuse_cells_kernel's work item)
    int meta, fmt_force;           // entry format of the call (use_contributions)
    const int *nonempty;           // buckets with entries (scan_apply_kernel)
    ListMap map[1 + MAX_EXTRA_MAPS];
};

__global__ __launch_bounds__(256) void tile_list_kernel(ListParams LP)
{
    const int *__restrict__ cursor = LP.cursor;
    const int n_tiles = LP.n_tiles, G = LP.G, split_part = LP.split_part;
    const int *nonempty = LP.nonempty;
    const ListMap &M = LP.map[blockIdx.y];
    int *ticket = M.ticket, *active = M.active, *items = M.items;
    int split_min = M.split_min;
    const int split_slots = M.split_slots, min_mean = M.min_mean, first_ticket = M.first_ticket, dense_tv = M.dense_tv;
    const int first_ticket_dense = M.first_ticket_dense, first_ticket_cells = M.first_ticket_cells;
    const int *abort = M.abort;
    const int t = blockIdx.x * 256 + threadIdx.x;
    // Which tile kernel takes the call (every thread works the choice out: which list a tile goes to depends on it).
    // Calls bucketed on the 4 x 4 x 8 tiles of the all-integer kernels in a tile-local format (LP.meta): the format -
    // contributions or records, use_contributions - IS the choice.  Other calls that fuse_dense_kernel is offered
    // (a merged batch of frames: generic records): by the call's density, records per non-empty (tile, frame) bucket
    // and tile voxel - half a record per voxel and frame or more is a real scene.
    // dense_tv: low bits the tiles' voxel count, bit 20: fuse_dense_kernel is offered the call, bit 21: fuse_cells_kernel is
    const bool dense_ok = dense_tv & (1 << 20), cells_ok = dense_tv & (1 << 21);
    const int fmt = entry_format(ticket, LP.meta, LP.fmt_force);
    const long long total = cursor[n_tiles * G];
    const long long half = (long long)*nonempty * ((dense_tv & 0xfffff) ? (dense_tv & 0xfffff) : 512) / 2;
    const bool dense = dense_ok && (LP.meta ? fmt == FMT_RECORDS : (total >= half || (dense_tv & (1 << 22))));
    const int tile_mode = dense ? MODE_DENSE : fmt != FMT_RECORDS && cells_ok ? (fmt == FMT_AGG ? MODE_CELLS_AGG : MODE_CELLS) : MODE_TILES;
    if (t == 0) {
        ticket[MODE_SLOT] = tile_mode;
        ticket[FORMAT_SLOT] = fmt;
        ticket[HINT_SLOT] = (int)(total > 0x7fffffffLL ? 0x7fffffffLL : total);
        ticket[HINT_SLOT + 1] = (int)(half > 0x7fffffffLL ? 0x7fffffffLL : half);
        // every tile kernel deals its first items statically (see there) and has a work counter of its own
        ticket[0] = first_ticket;
        ticket[TICKET_DENSE] = first_ticket_dense;
        ticket[TICKET_CELLS] = first_ticket_cells;
    }
    int n = 0;
    if (*abort) return;                             // a class id was out of range: no tile is listed, the map stays as it is
    if (t < n_tiles) n = cursor[(t + 1) * G] - cursor[t * G];
    // A call whose tiles are sparse on average (a synthetic frame of unrelated depths: ~40 records per
    // tile) is better off in the tile kernel, which spends less per tile; a real frame (hundreds to
    // thousands of records per tile) goes to the single-pass kernel.  The whole call goes one way.
    // (With ones features the single-pass kernel has no per-class state and wins on both.)
    if (split_min > 0 && (long long)cursor[n_tiles * G] < (long long)*nonempty * min_mean) split_min = 0;
    if (split_min > 0 && n > 0) {                   // single-pass kernel: every tile is an item, a big one several
        int nparts = 1, slot = 0xffff;
        if (n > split_min) {
            slot = atomicAdd(&ticket[SPLIT_TILES], 1);
            if (slot < split_slots) {
                nparts = (n + split_part - 1) / split_part;
                if (nparts > SPLIT_PARTS_MAX) nparts = SPLIT_PARTS_MAX;
            } else slot = 0xffff;                   // out of scratch slots: stays whole
        }
        const int base = atomicAdd(&ticket[SPLIT_ITEMS], nparts);
        for (int p = 0; p < nparts; ++p) { items[2 * (base + p)] = t; items[2 * (base + p) + 1] = p | (nparts << 8) | (slot << 16); }
        return;                                     // nothing is listed for the tile kernel
    }
    const int lane = threadIdx.x & 63;
    if (tile_mode == MODE_CELLS || tile_mode == MODE_CELLS_AGG) {
        // fuse_cells_kernel's list: fine load classes; a work item carries what the kernel would otherwise look up in a
        // chain of dependent loads: first entry, count, tile origin.  The block's tiles are ranked per class in LDS, so the
        // block issues ONE returning global atomic per class it holds (all in flight together).
        __shared__ int hist[CELL_CLASSES], hbase[CELL_CLASSES];
        if (threadIdx.x < CELL_CLASSES) hist[threadIdx.x] = 0;
        __syncthreads();
        const int fc = n > 0 ? cell_class(n) : -1;
        int rank = 0;
        if (fc >= 0) rank = atomicAdd(&hist[fc], 1);
        __syncthreads();
        if (threadIdx.x < CELL_CLASSES && hist[threadIdx.x] > 0) hbase[threadIdx.x] = atomicAdd(&ticket[CELL_COUNT + threadIdx.x], hist[threadIdx.x]);
        __syncthreads();
        if (fc >= 0) {
            const int tz = t % LP.nt2, ty = (t / LP.nt2) % LP.nt1, tx = t / (LP.nt2 * LP.nt1);
            M.items4[(size_t)fc * n_tiles + hbase[fc] + rank] =
                make_int4(t, cursor[t * G], n, (tx << LP.s0) | ((ty << LP.s1) << 10) | ((tz << LP.s2) << 20));
        }
        return;
    }
    const int cls = tile_class(n);
    for (int c = 0; c < TILE_CLASSES; ++c) {
        const bool mine = n > 0 && cls == c;
        const unsigned long long m = __ballot(mine);
        if (m == 0) continue;
        int base = 0;
        if (lane == 0) base = atomicAdd(&ticket[1 + c], __popcll(m));
        base = __shfl(base, 0, 64);
        if (mine) active[c * n_tiles + base + __popcll(m & ((1ull << lane) - 1ull))] = t;
    }
}

// ----------------------------------------------------------------------------
// tile kernel
// ----------------------------------------------------------------------------
constexpr int MAX_GROUPS = 256;

__device__ __forceinline__ unsigned div_magic(unsigned n, unsigned magic)   // n / C, magic = ceil(2^32/C), 0 for C == 1
{
    return magic ? __umulhi(n, magic) : n;
}


// Visit the corners of point record r that fall inside the tile whose origin
// is (o0, o1, o2): body(v, w) gets the tile-local voxel id and the corner weight.
// Sequential group (frame) of a record: eight spare bits of the record (scatter_kernel).
__device__ __forceinline__ int rec_group(const uint4 &r)
{
    return (int)((r.x >> 30) | ((r.y >> 30) << 2) | ((r.z >> 30) << 4) | ((r.w >> 30) << 6));
}

template <class F>
__device__ __forceinline__ void for_corners_idx(const TileParams &P, const uint4 &r, int o0, int o1, int o2, F body)
{
    const int k0 = r.x & 1023, k1 = (r.x >> 10) & 1023, k2 = (r.x >> 20) & 1023;
    // with several groups the top two bits of each ratio word carry the record's group (ratios of binned
    // pixels lie in [0, 1]: sign and top exponent bit are 0); ratios handed in by a caller are taken as they are
    const unsigned rm = P.G > 1 ? 0x3fffffffu : 0xffffffffu;
    const AxisFoot a0 = axis_foot(k0, __uint_as_float(r.y & rm), P.size0);
    const AxisFoot a1 = axis_foot(k1, __uint_as_float(r.z & rm), P.size1);
    const AxisFoot a2 = axis_foot(k2, __uint_as_float(r.w & rm), P.size2);
    // per axis: tile-local coordinate of the lower / upper corner, pre-shifted into its field of
    // the local voxel id, and whether it lies inside the tile
    const unsigned l0 = (unsigned)(a0.lo - o0), h0 = (unsigned)(a0.hi - o0);
    const unsigned l1 = (unsigned)(a1.lo - o1), h1 = (unsigned)(a1.hi - o1);
    const unsigned l2 = (unsigned)(a2.lo - o2), h2 = (unsigned)(a2.hi - o2);
    const bool in0[2] = {(l0 >> P.s0) == 0, (h0 >> P.s0) == 0};
    const bool in1[2] = {(l1 >> P.s1) == 0, (h1 >> P.s1) == 0};
    const bool in2[2] = {(l2 >> P.s2) == 0, (h2 >> P.s2) == 0};
    const unsigned p0[2] = {l0 << (P.s1 + P.s2), h0 << (P.s1 + P.s2)};
    const unsigned p1[2] = {l1 << P.s2, h1 << P.s2};
    const unsigned p2[2] = {l2, h2};
    // (w0 * w1) first, like the reference's product order (projection.py:319-323)
    const float w01[4] = {a0.wlo * a1.wlo, a0.wlo * a1.whi, a0.whi * a1.wlo, a0.whi * a1.whi};
    const float w2[2] = {a2.wlo, a2.whi};
#pragma unroll
    for (int cc = 0; cc < 8; ++cc) {
        const int ca = cc >> 2, cb = (cc >> 1) & 1, cd = cc & 1;
        if (in0[ca] && in1[cb] && in2[cd]) {
            const float pw = w01[ca * 2 + cb] * w2[cd];
            body(cc, (int)(p0[ca] | p1[cb] | p2[cd]), 1e-9f + pw);
        }
    }
}

template <class F>
__device__ __forceinline__ void for_corners(const TileParams &P, const uint4 &r, int o0, int o1, int o2, F body)
{
    for_corners_idx(P, r, o0, o1, o2, [&](int, int v, float w) { body(v, w); });
}

// Dev-only phase accounting (MF_STAMPS=1): cycles of workgroup-thread 0 between
// the barriers of the tile kernel, summed over all workgroups.
__device__ unsigned long long g_stamps[16];
#define MF_STAMP(i)                                                                   \
    if (STAMPS && tid == 0) {                                                         \
        const unsigned long long _t = __builtin_amdgcn_s_memtime();                   \
        stamp_acc[i] += _t - t_last;                                                  \
        t_last = _t;                                                                  \
    }

// LDS float add.  ds_add_f32 costs ~80 ns per wave instruction on gfx950 whatever the
// address pattern; one compare-and-swap round trip on the bit pattern costs ~11 ns when the
// lanes of a wave hit distinct words.  Try the swap once, and let only the lanes that lost a
// race (same word hit twice) take the hardware float atomic.
__device__ __forceinline__ void lds_add_f32(float *p, float x)
{
    unsigned *u = reinterpret_cast<unsigned *>(p);
    const unsigned seen = *u;
    const unsigned prev = atomicCAS(u, seen, __float_as_uint(__uint_as_float(seen) + x));
    if (prev != seen) atomicAdd(p, x);
}

// Workgroup barrier that leaves vector-memory operations (the LDS-DMA preload, prefetched
// records) in flight: __syncthreads() drains vmcnt when an LDS-DMA is outstanding.  LDS
// traffic of this wave is retired first; the compiler may not move memory ops across it.
__device__ __forceinline__ void barrier_keep_vm()
{
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

__device__ __forceinline__ float klow(const unsigned long long *W64, int i)
{
    return reinterpret_cast<const float *>(W64 + i)[0];
}

constexpr int EB = 2;                       // entries a thread keeps in flight / in registers per batch
constexpr int MAX_CHUNK = 16;          // frames whose W / S2 accumulators are live at once
constexpr float RESCALE_BELOW = 9.094947e-13f;   // 2^-40: fold the lazy decay into the deltas below this

// One workgroup per map tile, all frames of the call.
//
// For a voxel v the reference's sequential blend over frames f = 1..n is the
// affine recurrence  m_f = a_f * m_{f-1} + g_f * U_f  with
//   a_f = 1 - iw*S2_f/W_f,  g_f = iw/W_f,  U_f[c] = sum_k w_k^2 feat_k[c]
// (W_f, S2_f, U_f sums over the frame's corner contributions to v; frames that do
// not touch v have a_f = 1, U_f = 0).  Unrolled:
//   m_n = (prod_f a_f) * m_0 + s_n * sum_f (g_f / s_f) * U_f ,   s_f = prod_{f' <= f} a_f'
// so the C-wide decay of the reference becomes ONE scalar multiply per voxel and
// frame (s_f), every contribution is a single LDS float atomic scaled by
// k_f = g_f / s_f, contributions of all frames commute, and the old map value m_0
// is only needed in the final pass, where each touched voxel is read, combined and
// written exactly once.  s is kept away from underflow by folding it into the
// LDS deltas whenever it drops below 2^-40 (a_f can be 0 when iw = 1).
//
// LDS: D[TV][C] deltas, W/S2 accumulators for up to `gc` frames at a time,
// s and prod(a) per voxel.  Frames are processed in chunks of `gc` NON-EMPTY
// frames: pass 1 accumulates W, S2 (LDS float atomics), pass 2 walks the chunk's
// frames per voxel to turn them into k_f, pass 3 adds k_f * w^2 * feat.
//
// KIND: 0 = ones (C == 1), 1 = labels, 2 = dense fp32 features
template <int KIND, int MAXT, bool STAMPS = false>
__global__ __launch_bounds__(MAXT) void fuse_tiles_kernel(TileParams P)
{
    extern __shared__ float smem[];
    unsigned long long t_last = STAMPS ? __builtin_amdgcn_s_memtime() : 0ull;
    unsigned long long stamp_acc[8] = {0ull, 0ull, 0ull, 0ull, 0ull, 0ull, 0ull, 0ull};   // summed locally, written once at the end
    const int tid = threadIdx.x, NT = blockDim.x;
    const int C = P.C;
    const int sv = P.s0 + P.s1 + P.s2;
    const int TV = 1 << sv;
    const int GC = P.gc;
    // W_f and S2_f are accumulated as 64-bit FIXED-POINT integers: gfx950's LDS float atomic
    // (ds_add_f32) retires ~0.33 lanes/clk/CU, ds_add_u64 5.9 (tools/micro/lds_atomic_bench.hip),
    // and integer sums are exact and order independent.  After pass 2 the low word of a W slot
    // holds k_f as a float.
    unsigned long long *W64 = reinterpret_cast<unsigned long long *>(smem);   // [GC][TV]
    unsigned long long *S64 = W64 + (size_t)GC * TV;                           // [GC][TV]
    float *D = reinterpret_cast<float *>(S64 + (size_t)GC * TV);   // [TV][C] accumulated deltas (in units of s)
    // The float4 path (preloaded D, every row written) needs neither the factor of the old value nor the
    // touched flags, and the per-group arrays are sized by the call's groups: with 64 frames at C = 54 that is
    // the 4 KB that let a sixth frame into the chunk (tile_lds_fixed on the host mirrors this layout).
    const int G = P.G;
    const bool lean = P.vec4 != 0;
    const int OS = lean ? G + 1 : MAX_GROUPS + 1;  // stride of the two bucket-offset arrays
    float *sc = D + (size_t)TV * C;                // [TV] s: decay not yet folded into D
    float *osc = sc + TV;                          // [TV] prod a_f: factor of the old map value (not in the lean layout)
    int *offs2 = (int *)(lean ? osc : osc + TV);   // [2][OS] bucket starts: this tile / next tile
    int *cb = offs2 + 2 * OS;                      // [MAX_CHUNK + 1] entry offsets of the chunk's frames
    int *misc = cb + MAX_CHUNK + 1;                // [0] first tile, [1] non-empty frame count, [2] tile after next, [3] second tile, [4..] class sizes
    unsigned short *ne = (unsigned short *)(misc + 4 + TILE_CLASSES);   // [G | MAX_GROUPS] non-empty frames, ascending
    unsigned char *touched = (unsigned char *)(ne + (lean ? ((G + 1) & ~1) : MAX_GROUPS));   // [TV] (not in the lean layout)
    const int m1 = (1 << P.s1) - 1, m2 = (1 << P.s2) - 1;
    const unsigned n_el = (unsigned)TV * (unsigned)C;
    const int fx_c = 182 - P.fx_shift;
    const float fx_inv = __uint_as_float((unsigned)(127 - P.fx_shift) << 23);      // 2^-shift

    // The work list is walked with a ticket counter.  Everything the NEXT tile needs before
    // its first pass (ticket, tile id, bucket offsets: three dependent global round trips)
    // is fetched while the current tile is being processed.
    // Tickets are drawn three tiles ahead: while tile i is processed, the bucket offsets of tile
    // i+1 are fetched, the id of tile i+2 is looked up (ticket -> work list entry) and the ticket
    // of tile i+3 is drawn, so none of these dependent global round trips is ever waited for.
    auto resolve = [&](int idx) {          // ticket -> (class, position) -> tile id, -1 past the end
        int tile_id = -1;
#pragma unroll
        for (int c = 0; c < TILE_CLASSES; ++c) {
            const int cc = misc[4 + c];                 // tiles in class c (copied from P.ticket[1 + c] once)
            if (tile_id < 0 && idx >= 0 && idx < cc) tile_id = P.active[c * P.n_tiles + idx];
            idx -= cc;
        }
        return tile_id;
    };
    constexpr int OPT = (MAX_GROUPS + 1 + 63) / 64;     // offsets a thread may have to fetch (NT >= 64)
    auto load_offs = [&](int t, int (&o)[OPT]) {
        const int kb = t * G;
#pragma unroll
        for (int q = 0; q < OPT; ++q) {
            const int g = tid + q * NT;
            o[q] = (g <= G && kb + g > 0) ? P.cursor[kb + g - 1] : 0;
        }
    };
    auto store_offs = [&](int *dst, const int (&o)[OPT]) {
#pragma unroll
        for (int q = 0; q < OPT; ++q) {
            const int g = tid + q * NT;
            if (g <= G) dst[g] = o[q];
        }
    };
    // tid 0 advances this pipeline once per tile, right after the workgroup's only full vmcnt(0)
    // wait (the old-map preload): nothing else is outstanding there, so reading the results of the
    // loads issued one tile earlier costs no wait of its own (vmcnt retires in order: waiting for an
    // old load at any other point would also wait for every younger load and store of the wave).
    if (P.ticket[MODE_SLOT] != MODE_TILES) return;                               // another tile kernel takes the call (uniform)
    {
        int listed = 0;
#pragma unroll
        for (int c = 0; c < TILE_CLASSES; ++c) listed += P.ticket[1 + c];
        if (listed == 0) return;                                                 // nothing listed (uniform)
    }
    // The first four work items of a workgroup are dealt statically, list positions b, b + n, b + 2n,
    // b + 3n for workgroup b of n: the list is heaviest first, so every workgroup starts on one of the
    // n heaviest tiles (four tickets drawn in a row would hand the four heaviest to one workgroup).
    // tile_list_kernel starts the ticket counter at 4n.  A position past the end implies that all
    // later ones of this workgroup are past it too.
    int idx_pend = -1, act_pend = -1;                   // ticket drawn / work list entry being loaded
    if (tid == 0) {
#pragma unroll
        for (int c = 0; c < TILE_CLASSES; ++c) misc[4 + c] = P.ticket[1 + c];
        const int nb = gridDim.x, b = blockIdx.x;
        misc[0] = resolve(b);
        misc[3] = resolve(b + nb);
        act_pend = resolve(b + 2 * nb);
        idx_pend = b + 3 * nb;
    }
    __syncthreads();
    int tile = misc[0];
    int tile_next = misc[3];
    if (tile >= 0) {
        int o[OPT];
        load_offs(tile, o);
        store_offs(offs2, o);
    }
    __syncthreads();
    int buf = 0;
    // first EB entries per thread of the tile's first bucket range, fetched one tile ahead
    // They stay in registers for the whole tile: a tile with at most NT * EB records (most tiles of
    // a batch of unrelated frames) never goes back to global memory inside its chunk loop, whose
    // phases are then LDS latency plus barriers only.
    uint4 pre[EB];
    uint32_t prex[EB];
    auto prefetch_entries = [&](const int *o) {
        const int ta = o[0], tb = o[G];
#pragma unroll
        for (int j = 0; j < EB; ++j) {
            const int e = ta + tid + j * NT;
            pre[j].x = 0xffffffffu;
            prex[j] = 0;
            if (e < tb) { pre[j] = P.rec[e]; if (KIND == 1) prex[j] = P.aux[e]; }
        }
    };
    if (tile >= 0) prefetch_entries(offs2);

    while (tile >= 0) {
        int onext[OPT];
        if (tile_next >= 0) load_offs(tile_next, onext);      // in flight during the first chunk
        const int *offs = offs2 + buf * OS;
        const int t_a = offs[0];
        MF_STAMP(0)
        const int tz = tile % P.nt2, ty = (tile / P.nt2) % P.nt1, tx = tile / (P.nt2 * P.nt1);
        const int o0 = tx << P.s0, o1 = ty << P.s1, o2 = tz << P.s2;

        // compact the non-empty frames in order (wave 0), clear the tile state (everyone)
        if (tid < 64) {
            int count = 0;
            for (int b = 0; b < G; b += 64) {
                const int g = b + tid;
                const bool f = g < G && offs[g + 1] > offs[g];
                const unsigned long long m = __ballot(f);
                if (f) ne[count + __popcll(m & ((1ull << tid) - 1ull))] = (unsigned short)g;
                count += __popcll(m);
            }
            if (tid == 0) misc[1] = count;
        }
        if (P.vec4) {
            // Preload the tile's current map values straight into D with LDS-DMA (no VGPRs):
            // with D_0 = m_0 and s = 1 the invariant "true value = s * D" covers the old map
            // too, the final pass becomes write-only, and the 110 KB read overlaps pass 1.
            // One wave instruction moves 64 x 16 B = 1 KB; the LDS side is linear, the global
            // side is per lane (tile rows of T2*C floats are contiguous in the map).
            // A wave takes whole rows (row, wave and the row's address are scalars: the address arithmetic
            // runs on the scalar unit; measured before this, 6,912 float4s per tile each with its own divide
            // and 64-bit address made the preload and the final pass VALU-bound, not memory-bound).
            const unsigned row_len = (unsigned)C << P.s2, row4 = row_len >> 2;
            const int n_rows = TV >> P.s2, NW = NT >> 6;
            const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
            for (int r = wave; r < n_rows; r += NW) {
                const int l1 = r & m1, l0 = r >> P.s1;
                if (o0 + l0 < P.size0 && o1 + l1 < P.size1) {
                    const float *grow = P.map + (((size_t)(o0 + l0) * P.size1 + (o1 + l1)) * P.size2 + o2) * C;
                    float *drow = D + (size_t)r * row_len;
                    for (unsigned f0 = 0; f0 < row4; f0 += 64)
                        if (f0 + lane < row4)
                            __builtin_amdgcn_global_load_lds(
                                (const __attribute__((address_space(1))) void *)(grow + ((f0 + lane) << 2)),
                                (__attribute__((address_space(3))) void *)(drow + (f0 << 2)), 16, 0, 0);
                }
            }
        } else {
            for (unsigned i = tid; i < n_el; i += NT) D[i] = 0.0f;
        }
        for (int v = tid; v < TV; v += NT) { sc[v] = 1.0f; if (!lean) { osc[v] = 1.0f; touched[v] = 0; } }
        barrier_keep_vm();
        MF_STAMP(1)
        const int n_ne = misc[1];

        for (int c0 = 0; c0 < n_ne; c0 += GC) {
            const int nc = min(GC, n_ne - c0);
            for (int i = tid; i < nc * TV; i += NT) { W64[i] = 0ull; S64[i] = 0ull; }
            if (tid <= nc) cb[tid] = tid < nc ? offs[ne[c0 + tid]] : offs[ne[c0 + nc - 1] + 1];
            barrier_keep_vm();
            MF_STAMP(2)
            const int ea = cb[0], eb = cb[nc];
            // slot of entry e inside the chunk = number of frame starts cb[1..nc-1] that are <= e
            auto slot_of = [&](int e) { int j = 0; for (int q = 1; q < nc; ++q) j += e >= cb[q]; return j; };

            // pass 1: W_f, S2_f.  Entries are taken EB at a time per thread, all EB loads issued
            // before the first use (memory-level parallelism: one workgroup per CU).  Batches are
            // aligned to the tile's first record, so the first batch is always the register copy.
            const int bb0 = t_a + (ea - t_a) / (NT * EB) * (NT * EB);
            for (int bb = bb0; bb < eb; bb += NT * EB) {
                uint4 r[EB];
                if (bb == t_a) {
#pragma unroll
                    for (int j = 0; j < EB; ++j) r[j] = pre[j];
                } else {
#pragma unroll
                    for (int j = 0; j < EB; ++j) {
                        const int e = bb + tid + j * NT;
                        if (e >= ea && e < eb) r[j] = P.rec[e];
                    }
                }
#pragma unroll
                for (int j = 0; j < EB; ++j) {
                    const int e = bb + tid + j * NT;
                    if (e >= ea && e < eb) {
                        const int base = slot_of(e) * TV;
                        for_corners(P, r[j], o0, o1, o2, [&](int v, float w) {
                            atomicAdd(&W64[base + v], to_fixed(w, fx_c));
                            atomicAdd(&S64[base + v], to_fixed(w * w, fx_c));
                        });
                    }
                }
            }
            if (c0 == 0) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // this wave's DMA pieces (and everything older)
                if (tid == 0) {                                        // advance the ticket pipeline (see above)
                    misc[2] = act_pend;
                    act_pend = resolve(idx_pend);
                    idx_pend = atomicAdd(P.ctr, 1);
                }
            }
            barrier_keep_vm();
            MF_STAMP(3)
            // pass 2: per voxel, frames in order: s *= a_f, k_f = g_f / s
            for (int v = tid; v < TV; v += NT) {
                float s = sc[v], o = lean ? 1.0f : osc[v];
                bool any = false;
                for (int j = 0; j < nc; ++j) {
                    const unsigned long long wq = W64[j * TV + v];
                    if (wq != 0ull) {
                        const float Wv = (float)wq * fx_inv;
                        const float S2 = (float)S64[j * TV + v] * fx_inv;
                        const float rW = __builtin_amdgcn_rcpf(Wv);
                        const float a = 1.0f - P.iw * (S2 * rW);
                        o *= a; s *= a;
                        if (!(s >= RESCALE_BELOW)) {
                            // fold s into the deltas, and into the k of this chunk's earlier frames
                            // (their contributions are added in pass 3, in units of the old s)
                            for (int c = 0; c < C; ++c) D[v * C + c] *= s;
                            for (int t = 0; t < j; ++t) reinterpret_cast<float *>(&W64[t * TV + v])[0] *= s;
                            s = 1.0f;
                        }
                        reinterpret_cast<float *>(&W64[j * TV + v])[0] = P.iw * rW * __builtin_amdgcn_rcpf(s);
                        any = true;
                    }
                }
                if (any) { sc[v] = s; if (!lean) { osc[v] = o; touched[v] = 1; } }
            }
            __syncthreads();
            MF_STAMP(4)
            // pass 3: D += k_f * w^2 * feat
            if (KIND == 0 || KIND == 1) {
                // The adds of a record's corners go in rounds of four (k reads, reads of the current sums,
                // compare-and-swaps, float atomics of the lanes that lost a race) with NO branch between the LDS
                // operations of a round: the wait counters are only tracked exactly inside a basic block, so a
                // corner outside the tile goes through the motions on the word of one that is inside, with a
                // compare value that makes the swap a no-op.  (Dense scenes, where back-to-back swaps of a wave
                // collide, are taken by fuse_dense_kernel.)
                auto add = [&](int e, const uint4 &r, uint32_t label) {
                    if (KIND == 1 && label >= (uint32_t)C) return;
                    const int base = slot_of(e) * TV;
                    unsigned *Du = reinterpret_cast<unsigned *>(D);
                    int vi[8];
                    float qv[8];
#pragma unroll
                    for (int cc = 0; cc < 8; ++cc) vi[cc] = -1;
                    for_corners_idx(P, r, o0, o1, o2, [&](int cc, int v, float w) { vi[cc] = v; qv[cc] = w * w; });
                    int vf = 0;
#pragma unroll
                    for (int cc = 0; cc < 8; ++cc) vf = vi[cc] >= 0 ? vi[cc] : vf;
#pragma unroll
                    for (int h = 0; h < 8; h += 4) {
                        unsigned seen[4], prev[4];
                        int a[4];
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            a[i] = vi[h + i] >= 0 ? vi[h + i] : vf;
                            qv[h + i] *= klow(W64, base + a[i]);
                        }
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            a[i] = KIND == 0 ? a[i] : a[i] * C + (int)label;
                            seen[i] = Du[a[i]];
                        }
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            const bool on = vi[h + i] >= 0;
                            prev[i] = atomicCAS(&Du[a[i]], on ? seen[i] : 0xffffffffu,
                                                on ? __float_as_uint(__uint_as_float(seen[i]) + qv[h + i]) : 0xffffffffu);
                        }
#pragma unroll
                        for (int i = 0; i < 4; ++i)
                            if (vi[h + i] >= 0 && prev[i] != seen[i]) atomicAdd(&D[a[i]], qv[h + i]);
                    }
                };
                for (int bb = bb0; bb < eb; bb += NT * EB) {
                    uint4 r[EB];
                    uint32_t x[EB];
                    if (bb == t_a) {
#pragma unroll
                        for (int j = 0; j < EB; ++j) { r[j] = pre[j]; x[j] = prex[j]; }
                    } else {
#pragma unroll
                        for (int j = 0; j < EB; ++j) {
                            const int e = bb + tid + j * NT;
                            x[j] = 0;
                            if (e >= ea && e < eb) { r[j] = P.rec[e]; if (KIND == 1) x[j] = P.aux[e]; }
                        }
                    }
#pragma unroll
                    for (int j = 0; j < EB; ++j) {
                        const int e = bb + tid + j * NT;
                        if (e >= ea && e < eb) add(e, r[j], x[j]);
                    }
                }
            } else {
                // lanes-per-entry: the smallest power of two >= min(C, 64)
                int lpe = 1;
                while (lpe < C && lpe < 64) lpe <<= 1;
                const int sub = tid & (lpe - 1);
                const int per = NT / lpe;
                // neighbouring records hit the same voxels (neighbouring pixels): dealt in order, the
                // lanes of a wave would meet on a few LDS words and lose the compare-and-swap to each
                // other; a multiplicative permutation of the range spreads them
                const int nrec = eb - ea;
                int mulk = 1;
                if (nrec > 64) {
                    mulk = 61;
                    for (;;) {
                        int a_ = mulk, b_ = nrec;
                        while (b_) { const int t_ = a_ % b_; a_ = b_; b_ = t_; }
                        if (a_ == 1) break;
                        mulk += 2;
                    }
                }
                for (int e0 = tid / lpe; e0 < nrec; e0 += per) {
                    const int e = ea + (int)(((long long)e0 * mulk) % nrec);
                    const uint4 r = P.rec[e];
                    const float *f = (const float *)P.feat + (size_t)P.aux[e] * C;
                    const int base = slot_of(e) * TV;
                    for_corners(P, r, o0, o1, o2, [&](int v, float w) {
                        const float q = (w * w) * klow(W64, base + v);
                        for (int c = sub; c < C; c += lpe) lds_add_f32(&D[v * C + c], q * f[c]);
                    });
                }
            }
            if (c0 == 0 && tile_next >= 0) store_offs(offs2 + (buf ^ 1) * OS, onext);
            __syncthreads();
            MF_STAMP(5)
        }

        if (tile_next >= 0) prefetch_entries(offs2 + (buf ^ 1) * OS);   // lands during the final pass

        // final pass: every touched voxel is read, combined and written once, with all of a
        // thread's loads in flight before the first store (one workgroup per CU: the loop is
        // otherwise bound by HBM latency).  Rows of the tile (T2 voxels along z = T2*C floats)
        // are contiguous in the map; when they are 16-byte aligned the pass moves float4s, and a
        // float4 that straddles a touched and an untouched voxel rewrites the latter unchanged
        // (s = prod a = 1, D = 0), which is safe because the whole box belongs to this tile.
        if (P.vec4) {
            // D already contains the old values (preloaded above): true = s * D, store only; a wave takes
            // whole rows, as in the preload
            const unsigned row_len = (unsigned)C << P.s2, row4 = row_len >> 2;
            const int n_rows = TV >> P.s2, NW = NT >> 6, T2 = 1 << P.s2;
            const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
            for (int r = wave; r < n_rows; r += NW) {
                const int l1 = r & m1, l0 = r >> P.s1;
                if (o0 + l0 < P.size0 && o1 + l1 < P.size1) {
                    float4 *grow4 = reinterpret_cast<float4 *>(P.map + (((size_t)(o0 + l0) * P.size1 + (o1 + l1)) * P.size2 + o2) * C);
                    const float *drow = D + (size_t)r * row_len;
                    const float *srow = sc + r * T2;
                    for (unsigned f0 = 0; f0 < row4; f0 += 64) {
                        const unsigned f4 = f0 + lane;
                        if (f4 < row4) {
                            const unsigned i = f4 << 2;                    // float index inside the row
                            const unsigned v0 = div_magic(i, P.magicC), rem = i - v0 * C;
                            unsigned v1, v2, v3;                           // voxels of the other three floats
                            if (C >= 4) { v1 = v0 + (rem + 1 >= (unsigned)C); v2 = v0 + (rem + 2 >= (unsigned)C); v3 = v0 + (rem + 3 >= (unsigned)C); }
                            else { v1 = div_magic(i + 1, P.magicC); v2 = div_magic(i + 2, P.magicC); v3 = div_magic(i + 3, P.magicC); }
                            const float4 d = *reinterpret_cast<const float4 *>(drow + i);
                            float4 o;
                            o.x = srow[v0] * d.x; o.y = srow[v1] * d.y; o.z = srow[v2] * d.z; o.w = srow[v3] * d.w;
                            grow4[f4] = o;
                        }
                    }
                }
            }
        } else {
            constexpr int U = 4;
            for (unsigned b = 0; b < n_el; b += NT * U) {
                float old[U];
                size_t gi[U];
                unsigned li[U], vv[U];
#pragma unroll
                for (int j = 0; j < U; ++j) {
                    const unsigned i = b + j * NT + tid;
                    li[j] = 0xffffffffu;
                    if (i < n_el) {
                        const unsigned v = div_magic(i, P.magicC);
                        if (touched[v]) {
                            const unsigned c = i - v * C;
                            const int l2 = v & m2, l1 = (v >> P.s2) & m1, l0 = v >> (P.s1 + P.s2);
                            gi[j] = (((size_t)(o0 + l0) * P.size1 + (o1 + l1)) * P.size2 + (o2 + l2)) * C + c;
                            li[j] = i; vv[j] = v;
                            old[j] = P.map[gi[j]];
                        }
                    }
                }
#pragma unroll
                for (int j = 0; j < U; ++j)
                    if (li[j] != 0xffffffffu) P.map[gi[j]] = osc[vv[j]] * old[j] + sc[vv[j]] * D[li[j]];
            }
        }
        // the stores of the final pass drain while the next tile starts (its preload lands in D, whose
        // values this pass already holds in registers); a full barrier here waits for every write to be
        // acknowledged by memory
        if (P.vec4) barrier_keep_vm(); else __syncthreads();
        MF_STAMP(6)
        tile = tile_next;
        tile_next = misc[2];
        buf ^= 1;
    }
    if (STAMPS && tid == 0)
        for (int i = 0; i < 8; ++i) atomicAdd(&g_stamps[i], stamp_acc[i]);
}


// ----------------------------------------------------------------------------
// tile kernel for dense frames (real scenes)
// ----------------------------------------------------------------------------
// A real trajectory puts hundreds to thousands of records into a tile per frame, 30-140 of them on
// the same voxel with the same class.  fuse_tiles_kernel adds its fp32 deltas by compare-and-swap:
// lanes that meet on a word lose the swap and fall back to ds_add_f32 (29x slower than an integer
// atomic), which is where such batches spend their time.  This variant keeps EVERYTHING it
// accumulates as integers, on 4 x 4 x 8 tiles (four times the workgroups for a scene that sits on
// few tiles):
//   * W, S2 per (voxel, frame) cell as 64-bit fixed point, for a chunk of `gc` frames at a time;
//   * the frame of a record travels in the record (spare bits, see scatter_kernel): no per-tile
//     offset table, no search;
//   * the unrolled blend is written with SUFFIX products, m_n = (prod_f a_f) m_0 + sum_f t_f U_f with
//     t_f = g_f * prod_{f' > f} a_f' (per chunk; a later chunk multiplies what is there by its own
//     product): every added term t_f w^2 lies in [0, 1], so the deltas D are 64-bit fixed point too
//     (40 fraction bits; a positive term below one unit adds one unit, so that "non-zero" survives)
//     and pass 3 is one dependent LDS read and integer atomics that return nothing;
//   * the old map values never enter LDS: they are combined with prod a and D when the rows are stored
//     (a real scene's tile holds tens of thousands of records: moving its 55 KB is noise).
// Sums are exact and order independent: the result is run-to-run identical.
// tile_list_kernel decides per call which tile kernel runs (ticket[MODE_SLOT]); the host picks the
// tile shape from what the previous call on the same workspace counted (see tile_hint).
constexpr int DENSE_SV = 7;               // 4 x 4 x 8 tiles
constexpr int DENSE_FX = 40;              // fraction bits of the deltas
constexpr int DENSE_MAX_CHUNKS = 32;

// META: the records are in the tile-local meta format (sequential frames; class id and frame in the record, no aux words)
template <int KIND, int MAXT, bool META, bool STAMPS = false>
__global__ __launch_bounds__(MAXT, 4) void fuse_dense_kernel(TileParams P)      // <= 128 VGPRs: two workgroups of 512 fit a CU when gc is small
{
    extern __shared__ float smem[];
    unsigned long long t_last = STAMPS ? __builtin_amdgcn_s_memtime() : 0ull;
    unsigned long long stamp_acc[8] = {0ull, 0ull, 0ull, 0ull, 0ull, 0ull, 0ull, 0ull};
    const int tid = threadIdx.x, NT = blockDim.x;
    const int C = P.C;
    const int sv = P.s0 + P.s1 + P.s2;
    const int TV = 1 << sv, TVP = TV + 1;
    const int G = P.G, GC = P.gc;
    const int n_chunks = (G + GC - 1) / GC;
    const int m1 = (1 << P.s1) - 1;
    const unsigned n_el = (unsigned)TV * (unsigned)C;
    unsigned long long *A = reinterpret_cast<unsigned long long *>(smem);      // [GC][TV + 1][2] W, S2; low word of W = t_f after pass 2
    unsigned long long *Di = A + (size_t)GC * TVP * 2;                         // [TV][C] deltas, units of 2^-DENSE_FX
    float *atot = reinterpret_cast<float *>(Di + n_el);                        // [TV] prod a over the frames so far
    int *misc = reinterpret_cast<int *>(atot + TV);                            // [0] tile, [4..] class sizes, [NEXT] next tile
    constexpr int NEXT = 4 + TILE_CLASSES;
    int *offs = misc + 16;                                                     // [n_chunks + 1] record offsets of this tile's chunks
    int *offs_n = offs + DENSE_MAX_CHUNKS + 2;                                 // the next tile's
    const int fx_c = 182 - P.fx_shift;
    const float fx_inv = __uint_as_float((unsigned)(127 - P.fx_shift) << 23);  // 2^-shift
    const float di_inv = __uint_as_float((unsigned)(127 - DENSE_FX) << 23);    // 2^-DENSE_FX

    if (P.ticket[MODE_SLOT] != MODE_DENSE) return;                             // another tile kernel takes the call (uniform)
    {
        int listed = 0;
#pragma unroll
        for (int c = 0; c < TILE_CLASSES; ++c) listed += P.ticket[1 + c];
        if (listed == 0) return;
    }

    auto resolve = [&](int idx) {          // work list position -> tile id, -1 past the end
        int tile_id = -1;
#pragma unroll
        for (int c = 0; c < TILE_CLASSES; ++c) {
            const int cc = misc[4 + c];
            if (tile_id < 0 && idx >= 0 && idx < cc) tile_id = P.active[c * P.n_tiles + idx];
            idx -= cc;
        }
        return tile_id;
    };
    // Look-ups of the tiles ahead (ticket -> list entry -> chunk offsets: three dependent global round
    // trips), made by wave 0, each advanced once per tile; lane l holds chunk offset l.  The first four
    // list positions of a workgroup are dealt statically (b, b + n, b + 2n, b + 3n: the list is heaviest
    // first), tickets start at 4n.
    auto chunk_off = [&](int t) {          // lane tid: first record of chunk tid of tile t (or the tile's end)
        int o = 0;
        if (t >= 0 && tid <= n_chunks) {
            const int k = t * G + min(tid * GC, G);
            o = k > 0 ? P.cursor[k - 1] : 0;
        }
        return o;
    };
    int idx_pend = -1, act_pend = -1, rng_tile = -1, rng_off = 0, nx_tile = -1, nx_off = 0;
    if (tid == 0)
#pragma unroll
        for (int c = 0; c < TILE_CLASSES; ++c) misc[4 + c] = P.ticket[1 + c];
    __syncthreads();
    if (tid < 64) {
        const int nb = gridDim.x, b = blockIdx.x;
        const int t0 = resolve(b);
        const int o0 = chunk_off(t0);
        if (tid == 0) misc[0] = t0;
        if (tid <= n_chunks) offs[tid] = o0;
        nx_tile = resolve(b + nb);
        nx_off = chunk_off(nx_tile);
        rng_tile = resolve(b + 2 * nb);
        rng_off = chunk_off(rng_tile);
        act_pend = resolve(b + 3 * nb);
        idx_pend = tid == 0 ? atomicAdd(P.ctr, 1) : 0;
        idx_pend = __shfl(idx_pend, 0, 64);
    }
    __syncthreads();
    int tile = misc[0];
    if (tile < 0) return;

    uint4 pre[EB];
    uint32_t prex[EB];
    auto prefetch_entries = [&](int ta, int tb, uint4 (&q)[EB], uint32_t (&qx)[EB]) {
#pragma unroll
        for (int j = 0; j < EB; ++j) {                     // unconditional (clamped) loads: no branch, no wait in between
            const int e = min(ta + tid + j * NT, tb - 1);
            q[j] = P.rec[e];
            qx[j] = (KIND == 1 && !META) ? P.aux[e] : 0u;
        }
    };
    auto tile_origin = [&](int t, int &o0, int &o1, int &o2) {
        const int tz = t % P.nt2, ty = (t / P.nt2) % P.nt1, tx = t / (P.nt2 * P.nt1);
        o0 = tx << P.s0; o1 = ty << P.s1; o2 = tz << P.s2;
    };
    // Thread `tid` owns the float4s tid, tid + NT, ... of the tile image (OVM of them at most; the host checks).
    constexpr int OVM = 4;
    const unsigned n4 = n_el >> 2, row_len = (unsigned)C << P.s2;
    typedef float v4f __attribute__((ext_vector_type(4)));
    v4f *map4 = reinterpret_cast<v4f *>(P.map);
    auto elem_index = [&](int t, unsigned q, size_t &g4) {  // float4 q of tile t -> float4 index in the map, false outside the map
        int o0, o1, o2;
        tile_origin(t, o0, o1, o2);
        const unsigned i = q << 2;
        const unsigned r = div_magic(i, P.magicC) >> P.s2;
        const int l1 = r & m1, l0 = r >> P.s1;
        g4 = ((((size_t)(o0 + l0) * P.size1 + (o1 + l1)) * P.size2 + o2) * C + (i - r * row_len)) >> 2;
        return q < n4 && o0 + l0 < P.size0 && o1 + l1 < P.size1;
    };

    prefetch_entries(offs[0], offs[n_chunks], pre, prex);
    for (int i = tid; i < GC * TVP * 2 + (int)n_el; i += NT) A[i] = 0ull;        // cells and deltas
    // threads of a voxel in pass 2: PP consecutive lanes, each with FP consecutive frames of the chunk
    const int PP = NT >> sv > 0 ? NT >> sv : 1;

    while (true) {
        int o0, o1, o2;
        tile_origin(tile, o0, o1, o2);
        const int t_s = offs[0];
        const bool big = offs[n_chunks] - t_s > NT * EB;     // more records than the registers hold
        int tile_n = -1;
        MF_STAMP(0)
        for (int c = 0; c < n_chunks; ++c) {
            const int f_base = c * GC, nc = min(GC, G - f_base);
            const int ea = offs[c], eb = offs[c + 1];
            const int FP = (nc + PP - 1) / PP;
            barrier_keep_vm();                            // cells and masks are clear
            MF_STAMP(1)

            // ---- pass 1: W, S2 of every (voxel, frame) cell of the chunk
            auto p1_record = [&](const uint4 &r) {
                const int f = (META ? meta_frame(r) : rec_group(r)) - f_base;
                unsigned long long *cell = A + (size_t)f * TVP * 2;
                auto add = [&](int, int v, float w) {
                    atomicAdd(&cell[2 * v], to_fixed(w, fx_c));
                    atomicAdd(&cell[2 * v + 1], to_fixed(w * w, fx_c));
                };
                if (META) meta_corners_idx<2, 3>(r, add);
                else for_corners_idx(P, r, o0, o1, o2, add);
            };
            // A small tile (all its records fit the registers fetched a tile ahead) is taken from there, in
            // straight-line code.  A big tile is dealt in SEGMENTS: thread t walks records [t * S, (t + 1) * S) of
            // the chunk.  Neighbouring records come from neighbouring pixels and hit the same cells; dealt one
            // per lane, a wave would pile its 64 lanes onto a handful of LDS words per atomic.  With segments the
            // lanes of a wave are S records apart (other pixels, mostly other frames: other cells), and a lane's
            // own consecutive records queue up behind each other instead of inside one instruction.
            const int seg = (eb - ea + NT - 1) / NT;
            // segment of thread t: number 37 t mod NT (NT is a power of two), so that the lanes of a wave are
            // neither in the same frame nor at the same place of neighbouring frames (the same surface)
            const int my_seg = (tid * 37) & (NT - 1);
            const int my0 = min(eb, ea + my_seg * seg), my1 = min(eb, my0 + seg);
            if (!big) {
#pragma unroll
                for (int j = 0; j < EB; ++j) {
                    const int e = t_s + tid + j * NT;
                    if (e >= ea && e < eb) p1_record(pre[j]);
                }
            } else {
                for (int e = my0; e < my1; e += EB) {
                    uint4 r[EB];
#pragma unroll
                    for (int j = 0; j < EB; ++j) r[j] = P.rec[min(e + j, my1 - 1)];
#pragma unroll
                    for (int j = 0; j < EB; ++j)
                        if (e + j < my1) p1_record(r[j]);
                }
            }
            MF_STAMP(6)
            if (c == 0 && tid < 64) {                    // advance the look-ups
                if (tid == 0) misc[NEXT] = nx_tile;
                if (tid <= n_chunks) offs_n[tid] = nx_off;
                nx_tile = rng_tile; nx_off = rng_off;
                rng_tile = act_pend;
                rng_off = chunk_off(rng_tile);
                act_pend = resolve(idx_pend);
                idx_pend = tid == 0 ? atomicAdd(P.ctr, 1) : 0;
                idx_pend = __shfl(idx_pend, 0, 64);
            }
            barrier_keep_vm();
            MF_STAMP(2)

            if (c == 0) tile_n = misc[NEXT];
            MF_STAMP(5)

            // ---- pass 2: per voxel t_f = g_f * prod_{f' > f} a_f' into the cells; prod a over the chunk
            // multiplies what the earlier chunks left (deltas and the factor of the old value)
            {
                const int v = tid / PP, p = tid - v * PP;
                if (v < TV) {
                    const int f0 = p * FP, f1 = min(nc, f0 + FP);
                    float prod = 1.0f;
                    bool any = false;
                    for (int f = f0; f < f1; ++f) {
                        const unsigned long long *cl = A + ((size_t)f * TVP + v) * 2;
                        if (cl[0] != 0ull) {
                            const float rW = __builtin_amdgcn_rcpf((float)cl[0] * fx_inv);
                            prod *= 1.0f - P.iw * (((float)cl[1] * fx_inv) * rW);
                            any = true;
                        }
                    }
                    float x = prod;                      // -> product over this and all later parts of the voxel
                    for (int d = 1; d < PP; d <<= 1) {
                        const float y = __shfl_down(x, d, 64);
                        if (p + d < PP) x *= y;
                    }
                    float run = __shfl_down(x, 1, 64);   // product over the later parts
                    if (p + 1 >= PP) run = 1.0f;
                    const float total = __shfl(x, (tid & 63) - p, 64);
                    const bool touched = __any(any) && (__ballot(any) >> ((tid & 63) - p) & ((1ull << PP) - 1ull)) != 0ull;
                    for (int f = f1 - 1; f >= f0; --f) {       // latest frame first
                        unsigned long long *cl = A + ((size_t)f * TVP + v) * 2;
                        if (cl[0] != 0ull) {
                            const float rW = __builtin_amdgcn_rcpf((float)cl[0] * fx_inv);
                            const float a = 1.0f - P.iw * (((float)cl[1] * fx_inv) * rW);
                            reinterpret_cast<float *>(cl)[0] = P.iw * rW * run;
                            run *= a;
                        }
                    }
                    if (c == 0) {
                        if (p == 0) atot[v] = total;
                    } else if (touched) {
                        if (p == 0) atot[v] *= total;
                        for (int ch = p; ch < C; ch += PP) {
                            const unsigned long long d = Di[v * C + ch];
                            if (d != 0ull) {
                                unsigned long long nd = (unsigned long long)((double)d * (double)total);
                                if (nd == 0ull && total > 0.0f) nd = 1ull;      // "non-zero" survives
                                Di[v * C + ch] = nd;
                            }
                        }
                    }
                }
            }
            barrier_keep_vm();
            MF_STAMP(3)

            // ---- pass 3: D += t_f * w^2 (the class-id / ones feature is 1): the t_f of a record's corners are
            // read together (a corner outside the tile reads that of one inside), then integer atomics
            {
                auto p3_record = [&](const uint4 &r, uint32_t x) {
                    const unsigned long long *cell = A + (size_t)((META ? meta_frame(r) : rec_group(r)) - f_base) * TVP * 2;
                    int vi[8];
                    float qv[8];
#pragma unroll
                    for (int cc = 0; cc < 8; ++cc) vi[cc] = -1;
                    if (META) meta_corners_idx<2, 3>(r, [&](int cc, int v, float w) { vi[cc] = v; qv[cc] = w * w; });
                    else for_corners_idx(P, r, o0, o1, o2, [&](int cc, int v, float w) { vi[cc] = v; qv[cc] = w * w; });
                    int vf = 0;
#pragma unroll
                    for (int cc = 0; cc < 8; ++cc) vf = vi[cc] >= 0 ? vi[cc] : vf;
#pragma unroll
                    for (int cc = 0; cc < 8; ++cc) qv[cc] *= klow(cell, 2 * (vi[cc] >= 0 ? vi[cc] : vf));
#pragma unroll
                    for (int cc = 0; cc < 8; ++cc)
                        if (vi[cc] >= 0) {
                            const float term = qv[cc];
                            unsigned long long m = to_fixed(term, 182 - DENSE_FX);
                            if (m == 0ull && term > 0.0f) m = 1ull;
                            if (m != 0ull) atomicAdd(&Di[KIND == 0 ? vi[cc] : vi[cc] * C + (int)x], m);
                        }
                };
                if (!big) {
#pragma unroll
                    for (int j = 0; j < EB; ++j) {
                        const int e = t_s + tid + j * NT;
                        const uint32_t x = META ? meta_label(pre[j]) : prex[j];
                        if (e >= ea && e < eb && (KIND == 0 || x < (uint32_t)C)) p3_record(pre[j], x);
                    }
                } else {
                    for (int e = my0; e < my1; e += EB) {
                        uint4 r[EB];
                        uint32_t x[EB];
#pragma unroll
                        for (int j = 0; j < EB; ++j) {
                            const int q = min(e + j, my1 - 1);
                            r[j] = P.rec[q];
                            x[j] = KIND != 1 ? 0u : META ? meta_label(r[j]) : P.aux[q];
                        }
#pragma unroll
                        for (int j = 0; j < EB; ++j)
                            if (e + j < my1 && (KIND == 0 || x[j] < (uint32_t)C)) p3_record(r[j], x[j]);
                    }
                }
            }
            barrier_keep_vm();
            MF_STAMP(4)
            if (c == n_chunks - 1) {
                if (tile_n >= 0) prefetch_entries(offs_n[0], offs_n[n_chunks], pre, prex);   // the next tile's first records
                v4f oldv[OVM];
#pragma unroll
                for (int j = 0; j < OVM; ++j) {          // unconditional: an element outside the map reads element 0
                    size_t g4;
                    const bool in = elem_index(tile, tid + j * NT, g4);
                    oldv[j] = map4[in ? g4 : 0];
                }
                // ---- the tile's rows go out: old * prod a + D (every row whole: an untouched voxel is rewritten
                // with the value it had), and the deltas are cleared
#pragma unroll
                for (int j = 0; j < OVM; ++j) {
                    size_t g4;
                    const unsigned q = tid + j * NT;
                    if (elem_index(tile, q, g4)) {
                        const unsigned i = q << 2;
                        unsigned long long *d = Di + i;
                        v4f o;
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            const unsigned v = div_magic(i + k, P.magicC);
                            o[k] = oldv[j][k] * atot[v] + (float)d[k] * di_inv;
                            d[k] = 0ull;
                        }
                        map4[g4] = o;
                    }
                }
            }
            for (int i = tid; i < nc * TVP * 2; i += NT) A[i] = 0ull;     // the chunk's cells
        }
        if (tile_n < 0) break;
        tile = tile_n;
        if (tid <= n_chunks) offs[tid] = offs_n[tid];       // read by everyone after the next barrier ...
        barrier_keep_vm();                                   // ... which is this one (offs is read before the chunk loop)
    }
    if (STAMPS && tid == 0) {
        unsigned long long tot = 0;
        for (int i = 0; i < 7; ++i) { atomicAdd(&g_stamps[i], stamp_acc[i]); tot += stamp_acc[i]; }
        atomicMax(&g_stamps[7], tot);                   // the slowest workgroup
    }
}

// ----------------------------------------------------------------------------
// tile kernel for sparse frames (class ids / ones): compact (voxel, frame) cells over CONTRIBUTIONS
// ----------------------------------------------------------------------------
// A batch of unrelated frames (SURVEY distribution A) puts ~5 points of a frame into a 4 x 4 x 8 tile: of the
// 128 x 64 (voxel, frame) pairs of a tile only ~430 receive anything.  fuse_tiles_kernel walks such a tile in
// chunks of a few frames with dense per-frame accumulators; this kernel takes ALL frames of the tile at once:
//   mask    every contribution ORs its frame's bit into a 64-bit mask per voxel;
//   scan    cell index of (voxel v, frame f) = cellbase[v] + popcount(mask[v] & bits below f): the cells that
//           exist are numbered densely, voxel-major, frames ascending inside a voxel;
//   pass 1  W, S2 of every cell as 64-bit fixed-point integer atomics (exact, order independent);
//   pass 2  per cell g = iw / W and a = 1 - iw S2 / W (all threads), then one thread per voxel walks its cells
//           from the last frame to the first (SUFFIX form of the unrolled blend, like fuse_dense_kernel):
//           m_n = (prod_f a_f) m_0 + sum_f t_f U_f,  t_f = g_f prod_{f' > f} a_f';
//   pass 3  D[v][class] += t_f w^2 as 32-bit fixed point (31 fraction bits: every term and every sum of terms of
//           a (voxel, class) lies in [0, 1 + eps]; a positive term below one unit adds one unit so that
//           "non-zero" survives): integer atomics that return nothing;
//   final   rows out: old * prod a + D (two floats per instruction), the old rows loaded into registers; a float4
//           whose voxels received nothing is not written back.
// The passes read CONTRIBUTIONS (scatter_kernel, make_contribution): one 8-byte word per (point, corner inside the
// tile) with the tile-local voxel, class id, frame and the corner weight.  Rounds 2-3 stored one 16-byte record per
// (point, tile) and every pass decoded it and walked its eight corners (4.55 of them inside the tile on average):
// ~700 instructions per 64 records and pass, on SIMDs that are busy issuing ~90 % of the time
// (SQ_ACTIVE_INST_ANY, profiles/r03_sq_counters.txt).  Expanded once where the geometry is at hand anyway, a pass
// is ~15 instructions per 64 contributions with every lane busy.
// A tile's first 1,024 contributions are fetched one tile ahead and stay in registers for all passes.  The work
// list holds items {tile, first contribution, count, origin} (tile_list_kernel): ticket -> item is the only
// dependent look-up, drawn two tiles ahead by thread 0.
// A tile whose frames need more cells than fit (tiles next to the cameras) keeps its masks and takes its frames
// in several windows; a later window multiplies the deltas by its own prod a.  Calls of more than 64 frames take
// 64 at a time the same way.  All sums are integers: results are run-to-run identical.
// tile_list_kernel picks this kernel or fuse_dense_kernel from the call's density; both work on 4 x 4 x 8 tiles.
// Inclusive prefix sum over the 64 lanes of a wave with DPP row shifts and row broadcasts (six dependent VALU
// instructions; the __shfl_up ladder is six ds_bpermute round trips of ~100 cycles each, on the critical path of
// every tile).
__device__ __forceinline__ int wave_inclusive_scan(int x)
{
    x += __builtin_amdgcn_update_dpp(0, x, 0x111, 0xf, 0xf, true);      // row_shr:1
    x += __builtin_amdgcn_update_dpp(0, x, 0x112, 0xf, 0xf, true);      // row_shr:2
    x += __builtin_amdgcn_update_dpp(0, x, 0x114, 0xf, 0xf, true);      // row_shr:4
    x += __builtin_amdgcn_update_dpp(0, x, 0x118, 0xf, 0xf, true);      // row_shr:8: inclusive inside each row of 16
    x += __builtin_amdgcn_update_dpp(0, x, 0x142, 0xa, 0xf, false);     // row_bcast:15 into rows 1 and 3
    x += __builtin_amdgcn_update_dpp(0, x, 0x143, 0xc, 0xf, false);     // row_bcast:31 into rows 2 and 3
    return x;
}

constexpr int CELLS_FX = 31;              // fraction bits of the deltas

// A 64-bit fixed-point sum as a float: two 32-bit conversions and one multiply-add (the compiler's correctly rounded
// u64 -> f32 is a dozen instructions; the double rounding here is far below the 1e-4 the map is held to).
__device__ __forceinline__ float u64_to_float(unsigned long long x)
{
    return __builtin_fmaf((float)(unsigned)(x >> 32), 4294967296.0f, (float)(unsigned)x);
}

#ifndef CELLS_CR_COMMIT
#define CELLS_CR_COMMIT 8
#endif
template <int KIND, int F4, bool STAMPS = false, bool AGG = false, int CELLS_CR = 4>   // F4: float4s per thread and tile (ceil(32 C / 256)); AGG: aggregated 16-byte entries (bucket_agg_kernel); CELLS_CR: entries per thread kept in registers
__global__ __launch_bounds__(256, CELLS_CR > 4 ? 2 : 3) void fuse_cells_kernel(TileParams P)
{
    typedef typename std::conditional<AGG, uint4, uint2>::type Ent;
    extern __shared__ float smem[];
    unsigned long long t_last = STAMPS ? __builtin_amdgcn_s_memtime() : 0ull;
    unsigned long long stamp_acc[8] = {0ull, 0ull, 0ull, 0ull, 0ull, 0ull, 0ull, 0ull};
    typedef float v4f __attribute__((ext_vector_type(4)));
    typedef float v2f __attribute__((ext_vector_type(2)));
    constexpr int TV = 128, NT = 256;
    const int tid = threadIdx.x, lane = tid & 63;
    const int C = P.C, G = P.G, CAP = P.cells_cap;
    const unsigned n_el = (unsigned)TV * (unsigned)C, n4 = n_el >> 2;
    // LDS: per voxel frame masks and cell bases (arrays of their own: a lane's word is its voxel's bank), prod a;
    // cells W / S2; deltas; look-up words
    unsigned *vm = reinterpret_cast<unsigned *>(smem);                          // [4][128] mask of frames 0..31, cell base, mask 32..63, cell base of frame 32
    float *atot = reinterpret_cast<float *>(vm + 4 * TV);                       // [128]
    int *misc = reinterpret_cast<int *>(atot + TV);                             // [128]
    unsigned long long *Wc = reinterpret_cast<unsigned long long *>(misc + 128);   // [CAP + 1] W, then (g, a), then t_f
    unsigned long long *Sc = Wc + CAP + 1;                                      // [CAP + 1] S2
    unsigned *Du = reinterpret_cast<unsigned *>(Sc + CAP + 1);                  // [128][C] deltas, units of 2^-CELLS_FX
    constexpr int M_TOT = 0, M_FIT = 1, M_EA = 2 /* [2] */, M_CS = 8 /* [CELL_CLASSES + 1] list position of the classes' first tiles */,
                  M_CNT = 64 /* [64] inclusive per-frame cell counts */;
    static_assert(M_CS + CELL_CLASSES + 1 <= M_CNT, "look-up words");
    const Ent *rec2 = reinterpret_cast<const Ent *>(P.rec);
    const int fx_c = 182 - P.fx_shift;
    const float fx_inv = __uint_as_float((unsigned)(127 - P.fx_shift) << 23);  // 2^-shift
    const float du_inv = __uint_as_float((unsigned)(127 - CELLS_FX) << 23);    // 2^-CELLS_FX
    const float du_scale = __uint_as_float((unsigned)(127 + CELLS_FX) << 23);  // 2^CELLS_FX
    const float iw = P.iw;
    const float as_inv = __uint_as_float((unsigned)(127 - AGG_FS) << 23);      // 2^-AGG_FS

    if (P.ticket[MODE_SLOT] != (AGG ? MODE_CELLS_AGG : MODE_CELLS)) return;    // another tile kernel takes the call (uniform)
#ifdef CELLS_PRIO_DEF
    __builtin_amdgcn_s_setprio(CELLS_PRIO_DEF);
#endif
    // Work list position -> item; tile -1 past the end.  misc[M_CS + c] = list position of class c's first tile; a
    // workgroup's positions only grow, so the class is found by walking on from the last one.
    if (tid == 0) {
        int run = 0;
        for (int c = 0; c < CELL_CLASSES; ++c) { misc[M_CS + c] = run; run += P.ticket[CELL_COUNT + c]; }
        misc[M_CS + CELL_CLASSES] = run;
    }
    __syncthreads();
    int walk_c = 0;
    auto fetch_item = [&](int idx) {
        while (walk_c < CELL_CLASSES && idx >= misc[M_CS + walk_c + 1]) ++walk_c;      // (uniform)
        int4 it = make_int4(-1, 0, 0, 0);
        if (walk_c < CELL_CLASSES) it = P.light[(size_t)walk_c * P.n_tiles + (idx - misc[M_CS + walk_c])];   // (uniform; made scalar by `uniform` where it is first used)
        return it;
    };
    auto uniform = [](int4 it) {
        it.x = __builtin_amdgcn_readfirstlane(it.x); it.y = __builtin_amdgcn_readfirstlane(it.y);
        it.z = __builtin_amdgcn_readfirstlane(it.z); it.w = __builtin_amdgcn_readfirstlane(it.w);
        return it;
    };
    // first contribution of bucket k (after scatter_kernel cursor[k] is the END of bucket k)
    auto bucket_start = [&](int k) { return k > 0 ? P.cursor[k - 1] : 0; };
    for (int i = tid; i < 4 * TV; i += NT) vm[i] = 0u;
    for (int i = tid; i < 2 * (CAP + 1); i += NT) Wc[i] = 0ull;
    for (unsigned i = tid; i < n_el; i += NT) Du[i] = 0u;
    // The list is heaviest first (eight load classes); workgroup b of n takes positions b, b + n, b + 2 n, ...: every
    // workgroup gets its share of every class, in the same order, and no position depends on a returning atomic (a
    // ticket per tile was a memory round trip at every tile's start: the compiler waits for a value that comes out of a
    // divergent branch where the branches join).
    const int nb = gridDim.x;
    int lp = blockIdx.x;                                 // list position of the current item
    int4 item = uniform(fetch_item(lp)), item_n = uniform(fetch_item(lp + nb));
    if (item.x < 0) return;
    __syncthreads();

    // final pass: float4 j = tid + 256 i of the tile image [16 rows][8 voxels][C]: byte offset from the tile's first
    // voxel, first voxel and how many of its four floats belong to that voxel (tile-invariant)
    const unsigned rsb = (unsigned)P.size2 * (unsigned)C * 4u;                  // bytes between map rows (x + 1)
    unsigned pk[F4];
#pragma unroll
    for (int i = 0; i < F4; ++i) {
        const unsigned j = tid + 256u * i, c2 = 2u * C;
        const unsigned row = j / c2, col4 = j - row * c2;                  // (once per kernel)
        const unsigned e = 4u * col4, svr = div_magic(e, P.magicC), rem = e - svr * C;
        const unsigned cross = min(4u, (unsigned)C - rem);
        pk[i] = ((row & 15u) * 8u + svr) | (cross << 7) | ((row & 15u) << 10) | ((j < n4 ? 1u : 0u) << 14) | (col4 << 15);
    }
    auto row_offset = [&](unsigned p) {                 // bytes from the tile's first voxel (one register per slot is kept, not two)
        const unsigned row = (p >> 10) & 15u;
        return ((row >> 2) * (unsigned)P.size1 + (row & 3u)) * rsb + (p >> 15) * 16u;
    };

    // The first CELLS_CR x 256 contributions of a tile stay in registers for all passes; they are requested at the start
    // of the tile BEFORE, together with that tile's rows, and taken over at its end: no load whose result is still
    // awaited crosses the loop's back edge (where the compiler, which tracks the counter exactly only in straight-line
    // code, would wait for everything in flight - the rows just requested included).
    auto hide = [](Ent &c) {
        if constexpr (AGG) asm volatile("" : "+v"(c.x), "+v"(c.y), "+v"(c.z), "+v"(c.w));
        else asm volatile("" : "+v"(c.x), "+v"(c.y));
    };
    Ent cache[CELLS_CR], cache_n[CELLS_CR];
    auto prefetch = [&](int first, int n, Ent (&q)[CELLS_CR]) {       // unconditional (clamped) loads
#pragma unroll
        for (int j = 0; j < CELLS_CR; ++j) q[j] = rec2[first + min(tid + NT * j, max(n - 1, 0))];
    };
    prefetch(item.y, item.z, cache);
    auto base_of = [&](unsigned org) {                   // address of the tile's first voxel
        const int o0 = org & 1023u, o1 = (org >> 10) & 1023u, o2 = org >> 20;
        return reinterpret_cast<const char *>(P.map + ((size_t)o0 * P.size1 + o1) * ((size_t)P.size2 * C) + (size_t)o2 * C);
    };
    auto rows_inside = [&](unsigned org) {               // bit i: float4 slot i lies inside the map (a map whose extents are no multiples of four)
        const int o0 = org & 1023u, o1 = (org >> 10) & 1023u;
        unsigned in = 0xffffffffu;
        if (o0 + 4 > P.size0 || o1 + 4 > P.size1) {     // (uniform, rare)
            in = 0u;
#pragma unroll
            for (int i = 0; i < F4; ++i) {
                const unsigned row = (pk[i] >> 10) & 15u;
                if (o0 + (int)(row >> 2) < P.size0 && o1 + (int)(row & 3u) < P.size1) in |= 1u << i;
            }
        }
        return in;
    };
    auto load_rows = [&](const char *base, unsigned in, bool any, v4f (&q)[F4]) {
#pragma unroll
        for (int i = 0; i < F4; ++i) {
            q[i] = (v4f){0.0f, 0.0f, 0.0f, 0.0f};
            if (any && (unsigned)(i * NT) < n4) {           // (uniform)
                unsigned pki = pk[i];
                asm volatile("" : "+v"(pki));               // (see the final pass)
                if (((pki >> 14) & 1u) && ((in >> i) & 1u)) q[i] = *reinterpret_cast<const v4f *>(base + (size_t)row_offset(pki));
            }
        }
    };

    while (true) {
        const int tile = item.x, t_a = item.y, t_b = item.y + item.z;
        const unsigned org = (unsigned)item.w;
        const int tile_n = item_n.x;
        // ---- all global loads of the tile: the item after the next (scalar), the next tile's first contributions, this
        // tile's old rows (used in the final pass: the passes in between neither wait for them nor need the registers)
        MF_STAMP(7)
        const int4 item_n2 = fetch_item(lp + 2 * nb);
        prefetch(item_n.y, item_n.z, cache_n);
        // (the NEXT tile's rows instead, held in 28 more registers for a whole tile, were measured too: 1.93 against 1.88 ms -
        // the wait moves from the final pass to the point where the loads are issued: the memory pipeline, not the round
        // trip, is what the rows wait for)
        const char *tile_base = base_of(org);
        v4f oldv[F4];
        const unsigned rows_in = rows_inside(org);
        load_rows(tile_base, rows_in, true, oldv);
        // the contributions [ea, eb) of this tile: the cached ones from registers, the rest of a heavy tile from
        // memory (one trip ahead, so that the round trip runs beside the body)
        int cell_kept[CELLS_CR];                                                // pass 1's cells of the cached contributions, for pass 3
        auto for_contribs = [&](int ea, int eb, auto body) {
#pragma unroll
            for (int j = 0; j < CELLS_CR; ++j) {
                const int k = t_a + tid + NT * j;
                if (t_a + NT * j < eb) {                                        // (uniform)
                    // (the empty asm hides from the compiler that the word is the same in every pass: it would decode all
                    // four once and hold the fields in registers across the tile)
                    Ent c = cache[j];
                    hide(c);
                    if (k >= ea && k < eb) body(c, cell_kept[j]);
                }
            }
            // (four loads per thread in flight, one trip ahead: with one, every trip of 256 contributions waited a whole
            // memory round trip - ~1.3 M trips per launch, the bulk of the kernel's time in round 4's first version)
            int k = max(ea, t_a + NT * CELLS_CR);
            k += ((tid - k) & (NT - 1));                                        // the first k' >= k with k' = tid (mod 256): coalesced trips
            if (k < eb) {
                Ent nx[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) nx[q] = rec2[min(k + NT * q, eb - 1)];
#pragma unroll 1
                for (; k < eb; k += 4 * NT) {
                    Ent c[4];
#pragma unroll
                    for (int q = 0; q < 4; ++q) c[q] = nx[q];
                    if (k + 4 * NT < eb) {                                      // (per thread; clamped: no branch per load)
#pragma unroll
                        for (int q = 0; q < 4; ++q) nx[q] = rec2[min(k + NT * (4 + q), eb - 1)];
                    }
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        int none = -1;                                          // (a streamed contribution's cell is worked out in both passes)
                        if (k + NT * q < eb) body(c[q], none);
                    }
                }
            }
        };

        MF_STAMP(0)
        bool first = true;
        for (int F = 0; F < G; F += 64) {                   // at most 64 frames share the voxels' masks
            const int nf = min(64, G - F);
            int sa = t_a, sb = t_b;
            if (G > 64) {                                   // part of the tile's frames: look its range up
                if (tid <= 1) misc[M_EA + tid] = bucket_start(tile * G + (tid == 0 ? F : F + nf));
                __syncthreads();
                sa = misc[M_EA]; sb = misc[M_EA + 1];
                __syncthreads();
            }
            if (sa >= sb) continue;                         // (uniform)
            // ---- mask pass: which of these frames touch which voxel
            for_contribs(sa, sb, [&](const Ent &c, int &) {
                const unsigned f = ((c.x >> 15) & 255u) - (unsigned)F;
                atomicOr(&vm[(f >> 5) * 256u + (c.x & 127u)], 1u << (f & 31u));
            });
            __syncthreads();
            MF_STAMP(1)
            // exclusive scan of the voxels' cell counts inside the frame window wm (wave 0, two voxels per lane)
            auto scan_cells = [&](unsigned long long wm) {
                if (tid >= 64) return;
                const unsigned wl = (unsigned)wm, wh = (unsigned)(wm >> 32);
                const int v = 2 * lane;
                const int nl0 = __popc(vm[v] & wl), nl1 = __popc(vm[v + 1] & wl);
                const int c0n = nl0 + __popc(vm[256 + v] & wh), c1n = nl1 + __popc(vm[256 + v + 1] & wh);
                const int inc = wave_inclusive_scan(c0n + c1n);
                const int ex = inc - (c0n + c1n);
                vm[128 + v] = (unsigned)ex; vm[384 + v] = (unsigned)(ex + nl0);
                vm[128 + v + 1] = (unsigned)(ex + c0n); vm[384 + v + 1] = (unsigned)(ex + c0n + nl1);
                if (lane == 63) misc[M_TOT] = inc;
            };
            scan_cells(~0ull);
            __syncthreads();
            const bool split = misc[M_TOT] > CAP;           // (uniform) the cells of these frames do not fit at once
            if (split) {
                // inclusive per-frame cell counts: a window is a run of frames whose cells fit
                if (tid < 64) {
                    int cnt = 0;
                    const unsigned *mw = vm + (tid >> 5) * 256;
                    for (int v = 0; v < TV; ++v) cnt += (int)((mw[v] >> (tid & 31)) & 1u);
                    misc[M_CNT + tid] = wave_inclusive_scan(cnt);
                }
                __syncthreads();
            }
            for (int f0 = 0; f0 < nf;) {                    // windows of frames [F + f0, F + f1)
                int f1 = nf, ea = sa, eb = sb;
                unsigned long long wm = ~0ull;
                if (split) {
                    if (tid < 64) {
                        const int base = f0 > 0 ? misc[M_CNT + f0 - 1] : 0;
                        const int fit = __popcll(__ballot(tid >= f0 && tid < nf && misc[M_CNT + tid] - base <= CAP));
                        if (tid == 0) misc[M_FIT] = fit < 1 ? 1 : fit;       // a single frame has at most TV <= CAP cells
                    }
                    __syncthreads();
                    f1 = f0 + misc[M_FIT];
                    wm = (f1 >= 64 ? ~0ull : (1ull << f1) - 1ull) & ~((1ull << f0) - 1ull);
                    if (tid <= 1) misc[M_EA + tid] = bucket_start(tile * G + F + (tid == 0 ? f0 : f1));
                    scan_cells(wm);
                    __syncthreads();
                    ea = misc[M_EA]; eb = misc[M_EA + 1];
                }
                MF_STAMP(2)
                if (ea < eb) {
                    const unsigned wl = (unsigned)wm, wh = (unsigned)(wm >> 32);
                    // cell of a contribution: its voxel's base + the frames of the window below its own
                    auto cell_of = [&](const Ent &c) {
                        const unsigned f = ((c.x >> 15) & 255u) - (unsigned)F, v = c.x & 127u;
                        const unsigned *mw = vm + (f >> 5) * 256u;
                        return (int)mw[128 + v] + __popc(mw[v] & (f >> 5 ? wh : wl) & ((1u << (f & 31u)) - 1u));
                    };
                    // ---- pass 1: W, S2 of every cell
                    for_contribs(ea, eb, [&](const Ent &c, int &kept) {
                        const int ci = cell_of(c);
                        kept = ci;
                        if constexpr (AGG) {
                            atomicAdd(&Wc[ci], agg_W(c) << (P.fx_shift - AGG_FW));       // (fx_shift >= 33: make_layout bounds the points)
                            atomicAdd(&Sc[ci], agg_S(c) >> (AGG_FS - P.fx_shift));       // (fx_shift <= 50)
                        } else {
                            const float w = __uint_as_float(c.y);
                            atomicAdd(&Wc[ci], to_fixed(w, fx_c));
                            atomicAdd(&Sc[ci], to_fixed(w * w, fx_c));
                        }
                    });
                    __syncthreads();
                    MF_STAMP(3)
                    // ---- pass 2a: per cell g = iw / W, a = 1 - iw S2 / W (every thread takes cells)
                    const int used = misc[M_TOT];
                    for (int i = tid; i < used; i += NT) {
                        const float rW = __builtin_amdgcn_rcpf(u64_to_float(Wc[i]) * fx_inv);
                        float2 ga; ga.x = iw * rW; ga.y = 1.0f - iw * ((u64_to_float(Sc[i]) * fx_inv) * rW);
                        *reinterpret_cast<float2 *>(Wc + i) = ga;
                        Sc[i] = 0ull;
                    }
                    __syncthreads();
                    // ---- pass 2b: per voxel, cells from the last frame to the first: t_f = g_f prod_{f' > f} a_f';
                    // prod a over the window multiplies what the earlier windows left
                    if (tid < TV) {
                        const int v = tid, cs = (int)vm[128 + v], cnt = __popc(vm[v] & wl) + __popc(vm[256 + v] & wh);
                        float run = 1.0f;
                        for (int j = cs + cnt - 1; j >= cs; j -= 4) {       // four cells per trip: their reads do not wait for each other
                            float2 ga[4];
#pragma unroll
                            for (int q = 0; q < 4; ++q) ga[q] = *reinterpret_cast<const float2 *>(Wc + max(j - q, cs));
#pragma unroll
                            for (int q = 0; q < 4; ++q)
                                if (j - q >= cs) {
                                    reinterpret_cast<float *>(Wc + j - q)[0] = ga[q].x * run;
                                    run *= ga[q].y;
                                }
                        }
                        if (first) atot[v] = run;
                        else if (cnt > 0) {
                            atot[v] *= run;
                            for (int ch = 0; ch < C; ++ch) {
                                const unsigned d = Du[v * C + ch];
                                if (d != 0u) {
                                    unsigned nd = (unsigned)((double)d * (double)run);
                                    if (nd == 0u && run > 0.0f) nd = 1u;            // "non-zero" survives
                                    Du[v * C + ch] = nd;
                                }
                            }
                        }
                    }
                    __syncthreads();
                    MF_STAMP(4)
                    // ---- pass 3: D += t_f * w^2 (the class-id / ones feature is 1; an invalid class id adds nothing)
                    for_contribs(ea, eb, [&](const Ent &c, int &kept) {
                        const int ci = kept >= 0 ? kept : cell_of(c);
                        const float tf = klow(Wc, ci);
                        float term;
                        if constexpr (AGG) term = (u64_to_float(agg_S(c)) * as_inv) * tf;      // the group's sum of w^2 (every corner has w >= 1e-9)
                        else { const float w = __uint_as_float(c.y); term = (w * w) * tf; }
                        unsigned q = (unsigned)(term * du_scale);
                        if (q == 0u && (AGG ? tf > 0.0f : term > 0.0f)) q = 1u;
                        const unsigned x = (c.x >> 7) & 255u, v = c.x & 127u;
                        const bool xok = KIND == 0 || x < (unsigned)C;
                        atomicAdd(&Du[v * C + (KIND == 0 || !xok ? 0u : x)], xok ? q : 0u);
                    });
                    __syncthreads();
                    MF_STAMP(5)
                    for (int i = tid; i < used; i += NT) Wc[i] = 0ull;          // the window's cells
                    first = false;
                }
                f0 = f1;
                if (f0 < nf) __syncthreads();               // the next window's scan writes the cell bases
            }
            if (tid < TV) { vm[tid] = 0u; vm[256 + tid] = 0u; }
            if (F + 64 < G) __syncthreads();                // the next frames' mask pass ORs into cleared masks
        }
        // ---- final pass
        // What was requested at the tile's start is taken in HERE, before this tile's stores are issued: the vector-memory
        // counter retires in order, so behind the stores the wait for these loads would be a wait for the stores' round trip too.
#pragma unroll
        for (int j = 0; j < CELLS_CR; ++j) {
            if constexpr (AGG) asm volatile("" :: "v"(cache_n[j].x), "v"(cache_n[j].y), "v"(cache_n[j].z), "v"(cache_n[j].w));
            else asm volatile("" :: "v"(cache_n[j].x), "v"(cache_n[j].y));
        }
#pragma unroll
        for (int i = 0; i < F4; ++i) {
            if ((unsigned)(i * NT) < n4) {                  // (uniform)
                // (the empty asm hides that the slot's fields are the same for every tile: worked out before the loop, they
                // would be held - and spilled - across it)
                unsigned pki = pk[i];
                asm volatile("" : "+v"(pki));
                if ((pki >> 14) & 1u) {
                    unsigned j = tid + 256u * i;
                    asm volatile("" : "+v"(j));
                    const uint4 dq = *reinterpret_cast<const uint4 *>(Du + 4u * j);
                    *reinterpret_cast<uint4 *>(Du + 4u * j) = make_uint4(0u, 0u, 0u, 0u);
                    const unsigned sv0 = pki & 127u, cross = (pki >> 7) & 7u;
                    float ak[4];
                    if (F4 > 1 || C >= 4) {                 // (F4 > 1: C > 8, known when compiled) a float4 spans two voxels at most
                        const float a0 = atot[sv0], a1 = atot[(sv0 + 1u) & 127u];
                        ak[0] = a0; ak[1] = cross > 1u ? a0 : a1; ak[2] = cross > 2u ? a0 : a1; ak[3] = cross > 3u ? a0 : a1;
                    } else {
#pragma unroll
                        for (int k = 0; k < 4; ++k) ak[k] = atot[div_magic(4u * j + k, P.magicC) & 127u];
                    }
                    const bool changed = (dq.x | dq.y | dq.z | dq.w) != 0u || ak[0] != 1.0f || ak[3] != 1.0f || (F4 == 1 && C < 4 && (ak[1] != 1.0f || ak[2] != 1.0f));
                    if (((rows_in >> i) & 1u) && changed) {
                        const v2f d01 = (v2f){(float)dq.x, (float)dq.y} * (v2f){du_inv, du_inv};
                        const v2f d23 = (v2f){(float)dq.z, (float)dq.w} * (v2f){du_inv, du_inv};
                        const v2f o01 = __builtin_elementwise_fma((v2f){oldv[i][0], oldv[i][1]}, (v2f){ak[0], ak[1]}, d01);
                        const v2f o23 = __builtin_elementwise_fma((v2f){oldv[i][2], oldv[i][3]}, (v2f){ak[2], ak[3]}, d23);
                        *reinterpret_cast<v4f *>(const_cast<char *>(tile_base) + (size_t)row_offset(pki)) = (v4f){o01[0], o01[1], o23[0], o23[1]};
                    }
                }
            }
            if (i & 1) __builtin_amdgcn_sched_barrier(0);   // two slots at a time (registers: the rows of all seven are held)
        }
        MF_STAMP(6)
        if (tile_n < 0) break;
#pragma unroll
        for (int j = 0; j < CELLS_CR; ++j) cache[j] = cache_n[j];
        item = item_n; item_n = uniform(item_n2); lp += nb;
        __syncthreads();                                     // rows combined, deltas / cells / masks clear
    }
    if (STAMPS && tid == 0) {
        unsigned long long tot = 0;
        for (int i = 0; i < 8; ++i) { atomicAdd(&g_stamps[i], stamp_acc[i]); tot += stamp_acc[i]; }
        atomicMax(&g_stamps[15], tot);                  // the slowest workgroup
        atomicAdd(&g_stamps[14], tot);
    }
}

// ----------------------------------------------------------------------------
// single-group calls with class-id / ones features: one pass, integer sums
// ----------------------------------------------------------------------------
// With one group (a frame, or a merged batch) the update of a voxel is
//     new = a * old + g * U,   a = 1 - iw*S2/W,  g = iw/W,  U[c] = sum of w^2 over the corners with class c,
// and W, S2, U are plain sums over the records: no frame order, no scale to carry, nothing to read
// before the end.  So the tile's records are expanded ONCE and W, S2 and U are all accumulated as
// 64-bit fixed-point LDS integer atomics.  That is what a real scene needs: its neighbouring
// pixels land on the same voxel with the same class (28-136 points per voxel), and LDS float
// adds by compare-and-swap collapse there (lost races fall back to ds_add_f32, 29x slower than
// the integer atomic), while integer atomics on one word just queue up; the sums are exact and
// run-to-run identical as well.  A work item is a tile or, for a tile with many records, one of
// `nparts` record ranges: parts add what they touched to the tile's scratch slot in global memory
// (integer atomics: performed at the memory side, coherent across XCDs) and the part that arrives
// last reads the totals back and applies them.  Items are dealt round-robin (no ticket, no
// dependent look-ups): parts are bounded in size, so static dealing balances.
//   ticket[SPLIT_ITEMS] = items listed, ticket[SPLIT_TILES] = scratch slots handed out;
//   item = {tile, part | nparts << 8 | slot << 16}; the scratch is zeroed by the call's memset.
struct SingleParams {
    int size0, size1, size2, C;
    float *map;
    float iw;
    int s0, s1, s2;
    int nt1, nt2;
    unsigned magicC;
    int fx_shift;
    const int *cursor;
    const int *ticket;
    const int *items;
    const uint4 *rec;
    const uint32_t *aux;
    const float *feat;                 // dense features (fuse_single_dense_kernel)
    const unsigned *absmax;            // bits of their max |x| (fuse_single_dense_kernel)
    int *slot_count;                   // [slots] parts arrived
    unsigned long long *slot_ws;       // [slots][TV][2] W, S2
    unsigned long long *slot_u;        // [slots][TV * C] U (labels only; ones: U = S2)
    unsigned *slot_bits;               // [slots][ceil(TV * C / 32)] (voxel, class) pairs that met a sub-unit corner
};

template <int KIND, int NT, bool STAMPS = false>
__global__ __launch_bounds__(NT) void fuse_single_kernel(SingleParams P)
{
    extern __shared__ float smem[];
    const int tid = threadIdx.x;
    unsigned long long t_last = STAMPS ? __builtin_amdgcn_s_memtime() : 0ull;
    unsigned long long stamp_acc[8] = {0ull, 0ull, 0ull, 0ull, 0ull, 0ull, 0ull, 0ull};   // summed locally, written once at the end
    const int C = P.C;
    const int sv = P.s0 + P.s1 + P.s2;
    const int TV = 1 << sv;
    const unsigned n_el = (unsigned)TV * (unsigned)C;
    unsigned long long *U64 = reinterpret_cast<unsigned long long *>(smem);     // [TV][C] (labels)
    unsigned long long *W64 = U64 + (KIND == 1 ? n_el : 0);                      // [TV]
    unsigned long long *S64 = W64 + TV;                                          // [TV]
    unsigned long long *T64 = S64 + TV;                                          // [TV] sum of the sub-unit w^2, 34 more fraction bits
    float *sa = reinterpret_cast<float *>(T64 + TV);                             // [TV] a
    float *sg = sa + TV;                                                         // [TV] g (times 2^-shift), 0 = untouched
    float *st = sg + TV;                                                         // [TV] value added to a class that only met sub-unit corners
    int *misc = reinterpret_cast<int *>(st + TV);
    unsigned *bits = reinterpret_cast<unsigned *>(misc + 4);                     // [ceil(n_el / 32)] (labels)
    const int n_bits = KIND == 1 ? (int)((n_el + 31) >> 5) : 0;
    const int m1 = (1 << P.s1) - 1, m2 = (1 << P.s2) - 1;
    const int fx_c = 182 - P.fx_shift;
    const float fx_inv = __uint_as_float((unsigned)(127 - P.fx_shift) << 23);
    // Fixed point has an absolute resolution (one unit = 2^-shift): a corner weight below
    // sqrt(unit) ~ 3e-7 (a point on a voxel boundary has 1e-9) squares to less than a unit.  Such
    // squares go to a second per-voxel sum with 34 more fraction bits (they are below 2^-shift each,
    // so the sum of all of a call's points still fits), which completes S2, and the (voxel, class)
    // pairs they belong to are remembered in a bitmap: a class that received nothing else gets the
    // voxel's sub-unit sum, split evenly if several classes share it (exact for the usual single
    // corner; the shares are below 2e-7 in any case).
    const int fx_c2 = fx_c - 34;
    const float fx_inv2 = fx_inv * 5.8207661e-11f;                               // 2^-34
    // for_corners wants the tile kernel's parameter block: only the geometry fields are read
    TileParams T;
    T.size0 = P.size0; T.size1 = P.size1; T.size2 = P.size2; T.C = C; T.s0 = P.s0; T.s1 = P.s1; T.s2 = P.s2; T.G = 1;
    const int n_items = P.ticket[SPLIT_ITEMS];

    // the next item's descriptor and record range are looked up while the current item is processed
    // (two dependent global round trips per item otherwise, which dominate when items are small)
    int it = blockIdx.x, tile = -1, meta = 0, s = 0, e = 0;
    if (it < n_items) {
        tile = P.items[2 * it]; meta = P.items[2 * it + 1];
        s = tile > 0 ? P.cursor[tile - 1] : 0; e = P.cursor[tile];               // one group: bucket key = tile
    }
    for (; it < n_items; it += gridDim.x) {
        const int itn = it + gridDim.x;
        int tile_n = -1, meta_n = 0, s_n = 0, e_n = 0;
        if (itn < n_items) { tile_n = P.items[2 * itn]; meta_n = P.items[2 * itn + 1]; }
        const int part = meta & 255, nparts = (meta >> 8) & 255, slot = (meta >> 16) & 0xffff;
        const int tz = tile % P.nt2, ty = (tile / P.nt2) % P.nt1, tx = tile / (P.nt2 * P.nt1);
        const int o0 = tx << P.s0, o1 = ty << P.s1, o2 = tz << P.s2;
        const int len = (e - s + nparts - 1) / nparts;
        const int ea = s + part * len, eb = min(e, ea + len);
        MF_STAMP(0)
        if (KIND == 1) {
            uint4 *z = reinterpret_cast<uint4 *>(U64);
            for (unsigned i = tid; i < (n_el >> 1); i += NT) z[i] = make_uint4(0u, 0u, 0u, 0u);
            if ((n_el & 1) && tid == 0) U64[n_el - 1] = 0ull;
        }
        for (int v = tid; v < TV; v += NT) { W64[v] = 0ull; S64[v] = 0ull; T64[v] = 0ull; }
        for (int k = tid; k < n_bits; k += NT) bits[k] = 0u;
        __syncthreads();
        MF_STAMP(1)
        // Neighbouring records come from neighbouring pixels and hit the same voxels: a wave taking 64
        // consecutive records would pile its lanes onto a handful of LDS words per atomic.  The records
        // are therefore dealt to the lanes with a stride (a multiplicative permutation of the range).
        const int nrec = eb - ea;
        int mulk = 1;
        if (nrec > 64) {
            mulk = 61;
            while (true) {                                    // smallest of a few odd multipliers coprime with nrec
                int a_ = mulk, b_ = nrec;
                while (b_) { const int t_ = a_ % b_; a_ = b_; b_ = t_; }
                if (a_ == 1) break;
                mulk += 2;
            }
        }
        for (int q0 = tid; q0 < nrec; q0 += NT) {
            const int q = ea + (int)(((long long)q0 * mulk) % nrec);
            const uint4 r = P.rec[q];
            const uint32_t label = KIND == 1 ? P.aux[q] : 0u;
            for_corners(T, r, o0, o1, o2, [&](int v, float w) {
                const unsigned long long w2 = to_fixed(w * w, fx_c);
                atomicAdd(&W64[v], to_fixed(w, fx_c));
                if (w2 != 0ull) atomicAdd(&S64[v], w2);
                else atomicAdd(&T64[v], to_fixed(w * w, fx_c2));
                if (KIND == 1 && label < (uint32_t)C) {
                    const unsigned i = (unsigned)v * C + label;
                    if (w2 != 0ull) atomicAdd(&U64[i], w2);
                    else atomicOr(&bits[i >> 5], 1u << (i & 31));
                }
            });
        }
        __syncthreads();
        MF_STAMP(2)
        if (tile_n >= 0) { s_n = tile_n > 0 ? P.cursor[tile_n - 1] : 0; e_n = P.cursor[tile_n]; }
        bool finish = true;
        unsigned long long *ws = nullptr, *us = nullptr;
        if (nparts > 1) {
            // this part's sums -> the tile's slot; the last part to arrive takes the totals back
            ws = P.slot_ws + ((size_t)slot * TV) * 3;
            us = P.slot_u + (size_t)slot * n_el;
            for (int v = tid; v < TV; v += NT)
                if (W64[v] != 0ull) {
                    atomicAdd(&ws[3 * v], W64[v]);
                    if (S64[v] != 0ull) atomicAdd(&ws[3 * v + 1], S64[v]);
                    if (T64[v] != 0ull) atomicAdd(&ws[3 * v + 2], T64[v]);
                }
            unsigned *bs = P.slot_bits + (size_t)slot * n_bits;
            if (KIND == 1) {
                for (unsigned i = tid; i < n_el; i += NT)
                    if (U64[i] != 0ull) atomicAdd(&us[i], U64[i]);
                for (int k = tid; k < n_bits; k += NT)
                    if (bits[k] != 0u) atomicOr(&bs[k], bits[k]);
            }
            __threadfence();                               // the adds above are performed before the arrival below
            __syncthreads();
            if (tid == 0) misc[1] = atomicAdd(&P.slot_count[slot], 1) == nparts - 1;
            __syncthreads();
            finish = misc[1] != 0;
            if (finish) {
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                for (int v = tid; v < TV; v += NT) {       // totals through the atomic path
                    W64[v] = atomicAdd(&ws[3 * v], 0ull);
                    S64[v] = atomicAdd(&ws[3 * v + 1], 0ull);
                    T64[v] = atomicAdd(&ws[3 * v + 2], 0ull);
                }
                for (int k = tid; k < n_bits; k += NT) bits[k] = atomicOr(&bs[k], 0u);
                __syncthreads();
            }
        }
        MF_STAMP(3)
        if (finish) {
            // per voxel: a, the factor that turns the integer U[c] into the added value g * U[c], and
            // what a class that only met sub-unit corners gets
            for (int v = tid; v < TV; v += NT) {
                const unsigned long long wq = W64[v];
                float a = 1.0f, g = 0.0f, t = 0.0f;
                if (wq != 0ull) {
                    const float Wv = (float)wq * fx_inv;
                    const unsigned long long sq = S64[v], tq = T64[v];
                    const float tiny = (float)tq * fx_inv2;
                    const float S2 = (float)sq * fx_inv + tiny;
                    const float rW = __builtin_amdgcn_rcpf(Wv);
                    a = 1.0f - P.iw * (S2 * rW);
                    g = P.iw * rW * fx_inv;                                    // times U in units
                    if (KIND == 0) g = P.iw * rW * S2;                         // features = ones: U = S2
                    else if (tq != 0ull) {
                        int flagged = 0;
                        for (int c = 0; c < C; ++c) {
                            const unsigned i = (unsigned)v * C + c;
                            flagged += (bits[i >> 5] >> (i & 31)) & 1u;
                        }
                        t = flagged ? P.iw * rW * tiny / (float)flagged : 0.0f;
                    }
                }
                sa[v] = a; sg[v] = g; st[v] = t;
            }
            __syncthreads();
            MF_STAMP(4)
            // read-modify-write of the touched voxels, FB elements per thread at a time with all their
            // loads in flight before the first use (the loop is otherwise one HBM latency per element)
            constexpr int FB = 8;
            for (unsigned b0 = 0; b0 < n_el; b0 += NT * FB) {
                float old[FB], add[FB], av[FB];
                size_t gi[FB];
                bool on[FB];
#pragma unroll
                for (int j = 0; j < FB; ++j) {
                    const unsigned i = b0 + j * NT + tid;
                    on[j] = false;
                    if (i < n_el) {
                        const unsigned v = KIND == 0 ? i : div_magic(i, P.magicC);
                        const float g = sg[v];
                        if (g != 0.0f) {
                            const unsigned c = i - v * C;
                            const int l2 = v & m2, l1 = (v >> P.s2) & m1, l0 = v >> (P.s1 + P.s2);
                            gi[j] = (((size_t)(o0 + l0) * P.size1 + (o1 + l1)) * P.size2 + (o2 + l2)) * C + c;
                            old[j] = P.map[gi[j]];
                            add[j] = g;
                            if (KIND == 1) {
                                const unsigned long long u = nparts > 1 ? atomicAdd(&us[i], 0ull) : U64[i];
                                add[j] = g * (float)u + ((bits[i >> 5] >> (i & 31)) & 1u ? st[v] : 0.0f);
                            }
                            av[j] = sa[v];
                            on[j] = true;
                        }
                    }
                }
#pragma unroll
                for (int j = 0; j < FB; ++j)
                    if (on[j]) P.map[gi[j]] = av[j] * old[j] + add[j];
            }
        }
        __syncthreads();
        MF_STAMP(5)
        tile = tile_n; meta = meta_n; s = s_n; e = e_n;
    }
    if (STAMPS && tid == 0)
        for (int i = 0; i < 8; ++i) atomicAdd(&g_stamps[i], stamp_acc[i]);
}

// max |x| over a dense feature image, as float bits in *out (non-negative floats order like their bits)
__global__ __launch_bounds__(256) void feat_absmax_kernel(const float *__restrict__ f, long long n, int *out)
{
    unsigned m = 0u;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256)
        m = max(m, __float_as_uint(f[i]) & 0x7fffffffu);
    for (int o = 32; o > 0; o >>= 1) m = max(m, (unsigned)__shfl_down((int)m, o, 64));
    if ((threadIdx.x & 63) == 0 && m) atomicMax(reinterpret_cast<unsigned *>(out), m);
}

// The same single pass for dense fp32 features of few channels (an RGB map): U[c] = sum w^2 feat[c]
// as SIGNED 64-bit fixed point.  The scale comes from the call's largest |feature| (found by
// feat_absmax_kernel), sub-unit terms go to a second sum with 34 more fraction bits, like the
// squares of the class-id path.  Tiles are taken whole (no parts).
template <int NT>
__global__ __launch_bounds__(NT) void fuse_single_dense_kernel(SingleParams P)
{
    extern __shared__ float smem[];
    const int tid = threadIdx.x;
    const int C = P.C;
    const int sv = P.s0 + P.s1 + P.s2;
    const int TV = 1 << sv;
    const unsigned n_el = (unsigned)TV * (unsigned)C;
    unsigned long long *U64 = reinterpret_cast<unsigned long long *>(smem);     // [TV][C] signed
    unsigned long long *V64 = U64 + n_el;                                        // [TV][C] signed, sub-unit terms
    unsigned long long *W64 = V64 + n_el;                                        // [TV]
    unsigned long long *S64 = W64 + TV;                                          // [TV]
    unsigned long long *T64 = S64 + TV;                                          // [TV]
    float *sa = reinterpret_cast<float *>(T64 + TV);                             // [TV]
    float *sg = sa + TV;                                                         // [TV]
    const int m1 = (1 << P.s1) - 1, m2 = (1 << P.s2) - 1;
    const int fx_c = 182 - P.fx_shift, fx_c2 = fx_c - 34;
    const float fx_inv = __uint_as_float((unsigned)(127 - P.fx_shift) << 23);
    const float fx_inv2 = fx_inv * 5.8207661e-11f;                               // 2^-34
    // |w^2 * feat| < 2^E with E from the largest |feature|: U gets fx_shift - E fraction bits
    const unsigned mx = *P.absmax;
    int E = mx ? (int)(mx >> 23) - 126 : 0;
    int shift_u = P.fx_shift - E;
    shift_u = shift_u > 56 ? 56 : (shift_u < 4 ? 4 : shift_u);
    const int ux_c = 182 - shift_u, ux_c2 = ux_c - 34;
    const float ux_inv = __uint_as_float((unsigned)(127 - shift_u) << 23);
    const float ux_inv2 = ux_inv * 5.8207661e-11f;
    TileParams T;
    T.size0 = P.size0; T.size1 = P.size1; T.size2 = P.size2; T.C = C; T.s0 = P.s0; T.s1 = P.s1; T.s2 = P.s2; T.G = 1;
    const int n_items = P.ticket[SPLIT_ITEMS];

    int it = blockIdx.x, tile = -1, s = 0, e = 0;
    if (it < n_items) { tile = P.items[2 * it]; s = tile > 0 ? P.cursor[tile - 1] : 0; e = P.cursor[tile]; }
    for (; it < n_items; it += gridDim.x) {
        const int itn = it + gridDim.x;
        int tile_n = -1, s_n = 0, e_n = 0;
        if (itn < n_items) tile_n = P.items[2 * itn];
        const int tz = tile % P.nt2, ty = (tile / P.nt2) % P.nt1, tx = tile / (P.nt2 * P.nt1);
        const int o0 = tx << P.s0, o1 = ty << P.s1, o2 = tz << P.s2;
        {
            uint4 *z = reinterpret_cast<uint4 *>(U64);                           // U64, V64 are adjacent
            for (unsigned i = tid; i < n_el; i += NT) z[i] = make_uint4(0u, 0u, 0u, 0u);
        }
        for (int v = tid; v < TV; v += NT) { W64[v] = 0ull; S64[v] = 0ull; T64[v] = 0ull; }
        __syncthreads();
        const int nrec = e - s;
        int mulk = 1;
        if (nrec > 64) {                                       // strided dealing, as in fuse_single_kernel
            mulk = 61;
            for (;;) {
                int a_ = mulk, b_ = nrec;
                while (b_) { const int t_ = a_ % b_; a_ = b_; b_ = t_; }
                if (a_ == 1) break;
                mulk += 2;
            }
        }
        for (int q0 = tid; q0 < nrec; q0 += NT) {
            const int q = s + (int)(((long long)q0 * mulk) % nrec);
            const uint4 r = P.rec[q];
            const float *f = P.feat + (size_t)P.aux[q] * C;
            for_corners(T, r, o0, o1, o2, [&](int v, float w) {
                const float ww = w * w;
                const unsigned long long w2 = to_fixed(ww, fx_c);
                atomicAdd(&W64[v], to_fixed(w, fx_c));
                if (w2 != 0ull) atomicAdd(&S64[v], w2);
                else atomicAdd(&T64[v], to_fixed(ww, fx_c2));
                for (int c = 0; c < C; ++c) {
                    const float x = ww * f[c];
                    const float ax = fabsf(x);
                    if (!(ax < 3.0e38f)) continue;                  // inf / NaN features are not representable here
                    unsigned long long m = to_fixed(ax, ux_c);
                    unsigned long long *dst = &U64[v * C + c];
                    if (m == 0ull) { m = to_fixed(ax, ux_c2); dst = &V64[v * C + c]; }
                    if (m != 0ull) atomicAdd(dst, x < 0.0f ? 0ull - m : m);
                }
            });
        }
        __syncthreads();
        if (tile_n >= 0) { s_n = tile_n > 0 ? P.cursor[tile_n - 1] : 0; e_n = P.cursor[tile_n]; }
        for (int v = tid; v < TV; v += NT) {
            const unsigned long long wq = W64[v];
            float a = 1.0f, g = 0.0f;
            if (wq != 0ull) {
                const float Wv = (float)wq * fx_inv;
                const float S2 = (float)S64[v] * fx_inv + (float)T64[v] * fx_inv2;
                const float rW = __builtin_amdgcn_rcpf(Wv);
                a = 1.0f - P.iw * (S2 * rW);
                g = P.iw * rW;
            }
            sa[v] = a; sg[v] = g;
        }
        __syncthreads();
        constexpr int FB = 4;
        for (unsigned b0 = 0; b0 < n_el; b0 += NT * FB) {
            float old[FB], add[FB], av[FB];
            size_t gi[FB];
            bool on[FB];
#pragma unroll
            for (int j = 0; j < FB; ++j) {
                const unsigned i = b0 + j * NT + tid;
                on[j] = false;
                if (i < n_el) {
                    const unsigned v = div_magic(i, P.magicC);
                    const float g = sg[v];
                    if (g != 0.0f) {
                        const unsigned c = i - v * C;
                        const int l2 = v & m2, l1 = (v >> P.s2) & m1, l0 = v >> (P.s1 + P.s2);
                        gi[j] = (((size_t)(o0 + l0) * P.size1 + (o1 + l1)) * P.size2 + (o2 + l2)) * C + c;
                        old[j] = P.map[gi[j]];
                        add[j] = g * ((float)(long long)U64[i] * ux_inv + (float)(long long)V64[i] * ux_inv2);
                        av[j] = sa[v];
                        on[j] = true;
                    }
                }
            }
#pragma unroll
            for (int j = 0; j < FB; ++j)
                if (on[j]) P.map[gi[j]] = av[j] * old[j] + add[j];
        }
        __syncthreads();
        tile = tile_n; s = s_n; e = e_n;
    }
}

// ----------------------------------------------------------------------------
// parity kernels (a3, a4, fused a3+a4)
// ----------------------------------------------------------------------------
__global__ void transform_rays_kernel(const float *__restrict__ cam, long long n_pix,
                                      const float *__restrict__ poses, int n_frames, float *out)
{
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n_pix * n_frames) return;
    const int f = (int)(idx / n_pix);
    const long long pix = idx - (long long)f * n_pix;
    float q0, q1, q2;
    rotate_ray(poses + f * 12 + 3, cam[pix * 3], cam[pix * 3 + 1], cam[pix * 3 + 2], q0, q1, q2);
    out[idx * 3] = q0; out[idx * 3 + 1] = q1; out[idx * 3 + 2] = q2;
}

struct BinOut {
    int64_t *i0, *i1, *i2;
    float *r0, *r1, *r2;
    uint8_t *valid;
};

__device__ __forceinline__ void store_bin(const BinOut &o, long long idx, bool ok, int k0, int k1, int k2,
                                          float r0, float r1, float r2)
{
    if (o.i0) o.i0[idx] = k0;
    if (o.i1) o.i1[idx] = k1;
    if (o.i2) o.i2[idx] = k2;
    if (o.r0) o.r0[idx] = r0;
    if (o.r1) o.r1[idx] = r1;
    if (o.r2) o.r2[idx] = r2;
    if (o.valid) o.valid[idx] = ok ? 1 : 0;
}

__global__ void bin_rays_kernel(Bins B, const float *__restrict__ origin, const float *__restrict__ rays,
                                int rays_per_frame, const float *__restrict__ depth, int n_frames,
                                long long n_pix, float min_d, float max_d, BinOut o)
{
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n_pix * n_frames) return;
    const int f = (int)(idx / n_pix);
    const long long ri = rays_per_frame ? idx : idx - (long long)f * n_pix;
    const float d = depth[idx];
    const float m0 = rays[ri * 3] * d, m1 = rays[ri * 3 + 1] * d, m2 = rays[ri * 3 + 2] * d;
    const float p0 = origin[f * 3] + m0, p1 = origin[f * 3 + 1] + m1, p2 = origin[f * 3 + 2] + m2;
    int k0, k1, k2; float r0, r1, r2;
    const bool ok = bin_point(B, p0, p1, p2, d, min_d, max_d, k0, k1, k2, r0, r1, r2);
    store_bin(o, idx, ok, k0, k1, k2, r0, r1, r2);
}

__global__ void unproject_bin_kernel(FuseParams P, BinOut o)
{
    const long long idx = point_index<0>(P, 256);
    if (idx < 0) return;
    Point pt; uint32_t aux = 0;
    const bool ok = get_point<0>(P, idx, pt, aux);
    // outputs in the reference's (x, y, z) order of bin_rays
    store_bin(o, idx, ok, pt.k1, pt.k0, pt.k2, pt.r1, pt.r0, pt.r2);
}

// ----------------------------------------------------------------------------
// host side
// ----------------------------------------------------------------------------
static int ilog2_floor(unsigned x) { int l = 0; while ((2u << l) <= x) ++l; return l; }

// Development knobs come from the environment.  A value outside [lo, hi] (or not a number) is ignored with one
// line on stderr: nothing unchecked reaches a shift count, a thread count or an LDS size.
static int env_int(const char *name, int lo, int hi, int dflt)
{
    const char *e = getenv(name);
    if (!e || !*e) return dflt;
    char *end = nullptr;
    const long v = strtol(e, &end, 10);
    if (end == e || *end != '\0' || v < lo || v > hi) {
        fprintf(stderr, "[massfuse] %s=%s ignored (expected an integer in [%d, %d])\n", name, e, lo, hi);
        return dflt;
    }
    return (int)v;
}

// Tile extents: the largest power-of-two voxel count whose LDS image (C floats of
// deltas + scales + flag + four frames of 64-bit W/S2 accumulators per voxel) fits the
// CU's LDS, capped at 512 voxels (8 x 8 x 8); z gets up to 8 so that HBM runs stay long.
// Tuning override for experiments: MF_TILE="s0 s1 s2 threads" (log2 extents).
// fuse_dense_kernel: gc rows of (TV + 1) 16-byte cells, 8-byte deltas, prod a per voxel, look-up words
static size_t dense_lds_bytes(int C, int gc)
{
    const size_t TV = (size_t)1 << DENSE_SV;
    return (size_t)gc * (TV + 1) * 16 + TV * C * 8 + TV * 4 + (16 + 2 * (DENSE_MAX_CHUNKS + 2)) * 4 + 16;
}

// frames per chunk of fuse_dense_kernel: up to 32 within 128 KB of LDS (measured on the room batch: 8 frames
// and two workgroups per CU 2.15 ms, 16: 2.10, 32: 2.04; the rest of the LDS is left to the bucketing
// kernels of the next batch).  MF_DENSE_GC overrides.
static int dense_chunk_frames(int C, int G)
{
    static const int forced_raw = env_int("MF_DENSE_GC", 1, 64, 0);
    static const int forced = forced_raw > 0 ? 1 << ilog2_floor((unsigned)forced_raw) : 0;     // a power of two (halved / doubled below)
    int gc = forced > 0 ? forced : 32;
    if (gc > 64) gc = 64;
    if (gc > G) gc = G;
    while (gc > 1 && dense_lds_bytes(C, gc) > (forced > 0 ? 160 : 128) * 1024) gc /= 2;
    while ((G + gc - 1) / gc > DENSE_MAX_CHUNKS && gc < 64) gc *= 2;
    return gc;
}

// May a call of G sequential frames onto this grid be bucketed on the 4 x 4 x 8 tiles of fuse_dense_kernel?
// (its LDS image fits, the tile image fits the registers of 512 threads; the feature kind and the alignment
// of the rows are checked when the kernel is launched: fuse_tiles_kernel works on any tile shape)
static bool dense_shape_ok(const mf_grid *g, int G)
{
    static const bool on = env_int("MF_DENSE", 0, 1, 1) != 0;
    if (!on || G < 2) return false;
    const int gc = dense_chunk_frames(g->channels, G);
    return dense_lds_bytes(g->channels, gc) <= 160 * 1024 && (G + gc - 1) / gc <= DENSE_MAX_CHUNKS &&
           ((size_t)g->channels << DENSE_SV) / 4 <= 4 * 512;
}

// fuse_cells_kernel: per voxel 16 bytes of masks and cell bases, prod a, look-up words, cap + 1 cells of 16 bytes, 4-byte deltas
constexpr int CELLS_SV = 7, CELLS_NT = 256;          // 4 x 4 x 8 tiles, 256 threads (the instantiation the host launches)
static size_t cells_lds_bytes(int C, int cap)
{
    const size_t TV = (size_t)1 << CELLS_SV;
    return TV * 16 + TV * 4 + 128 * 4 + (size_t)(cap + 1) * 16 + TV * C * 4;
}

// Workgroups of fuse_cells_kernel per CU and the cells each of them holds: as many workgroups as leave each at
// least 1,100 cells (a tile of a batch of unrelated frames needs ~600), eight at most (256 threads each).
static bool cells_config(int C, int lds_per_cu, int reserve, int most, int &cap, int &per_cu)
{
    static const int forced = env_int("MF_CELLS_PER_CU", 1, 8, 0);            // dev
    if (C > 64) return false;                                                  // the tile image in eight float4s per thread
    const size_t fixed = cells_lds_bytes(C, 0);
    for (per_cu = forced > 0 ? forced : most; per_cu >= 1; --per_cu) {
        const size_t budget = ((size_t)(lds_per_cu - reserve) / per_cu) & ~(size_t)2047;   // LDS is handed out in blocks (three requests of 53 KB did not share a CU, three of 52 KB do)
        const size_t want = forced > 0 || per_cu == 1 ? (size_t)((1 << CELLS_SV) + 1) * 16 : (size_t)800 * 16;
        if (budget < fixed + want) { if (forced > 0) return false; continue; }
        cap = (int)((budget - fixed) / 16) - 1;
        return cap >= (1 << CELLS_SV);          // a single frame's cells (one per voxel at most) always fit
    }
    return false;
}

// May a call of G sequential frames of class ids / ones onto this grid be bucketed on 4 x 4 x 8 tiles, for
// fuse_dense_kernel / fuse_cells_kernel (tile_list_kernel picks one of them from the call's density)?  Decided
// from the call's arguments alone, so a stage / commit pair and a repeated run agree.
static bool int_tiles_ok(const mf_grid *g, int G, int feat_kind, float iw)
{
    // (the all-integer kernels bound every term by 1, which needs 0 <= iw <= 1: any other blend weight, which the
    // reference accepts, takes the float tile kernel)
    if (feat_kind == MF_FEAT_DENSE_F32 || !(iw >= 0.0f && iw <= 1.0f) || !dense_shape_ok(g, G)) return false;
    return (uintptr_t)g->map % 16 == 0 && g->size2 % 8 == 0 &&
           (size_t)g->size0 * g->size1 * g->size2 * g->channels < ((size_t)1 << 34);
}

static int g_gc_override = -1;
// MF_TILE="s0 s1 s2 threads [gc]": log2 tile extents (each 0..4, at most 512 voxels in all), workgroup size (a
// multiple of 64 up to 1024) and frames per chunk (1..MAX_CHUNK).  Anything else is ignored as a whole.
static bool tile_override(int &s0, int &s1, int &s2, int &nt)
{
    static int v[5] = {-1, -1, -1, -1, -1};
    static bool parsed = false, have = false;
    if (!parsed) {
        parsed = true;
        const char *e = getenv("MF_TILE");
        const int n = e ? sscanf(e, "%d %d %d %d %d", &v[0], &v[1], &v[2], &v[3], &v[4]) : 0;
        have = n >= 4 && v[0] >= 0 && v[0] <= 4 && v[1] >= 0 && v[1] <= 4 && v[2] >= 0 && v[2] <= 4 &&
               v[0] + v[1] + v[2] <= 9 && v[3] >= 64 && v[3] <= 1024 && v[3] % 64 == 0 &&
               (n < 5 || (v[4] >= 1 && v[4] <= MAX_CHUNK));
        if (e && *e && !have)
            fprintf(stderr, "[massfuse] MF_TILE=\"%s\" ignored (expected \"s0 s1 s2 threads [gc]\": log2 extents 0..4 each, "
                            "sum <= 9, threads a multiple of 64 in [64, 1024], gc in [1, %d])\n", e, MAX_CHUNK);
        g_gc_override = have && n >= 5 ? v[4] : -1;
    }
    if (have) { s0 = v[0]; s1 = v[1]; s2 = v[2]; nt = v[3]; }
    return have;
}

static void choose_tile(const mf_grid *g, int n_groups, bool dense, int &s0, int &s1, int &s2)
{
    int nt_unused;
    if (tile_override(s0, s1, s2, nt_unused)) return;
    if (dense && dense_shape_ok(g, n_groups)) { s0 = 2; s1 = 2; s2 = 3; return; }
    const size_t budget = 158 * 1024 - 8 * (MAX_GROUPS + 32) - 2 * MAX_GROUPS;
    size_t per_voxel = (size_t)g->channels * 4 + 8 + 1 + 4 * 16;      // deltas, scales, flag, >= 4 frames of W/S2
    unsigned tv = (unsigned)(budget / per_voxel);
    if (tv < 1) tv = 1;
    if (tv > 512) tv = 512;       // larger tiles starve the chip when a scene concentrates on few tiles
    // a single frame of a real scene lands on few tiles (a wall near the camera); 4 x 4 x 8
    // tiles give four times the workgroups for the same points, and with one or two frames
    // there is little temporal reuse of a tile's LDS image to lose
    if (n_groups <= 2 && tv > 128) tv = 128;
    int sv = ilog2_floor(tv);
    s2 = 3;
    if (s2 > sv) s2 = sv;
    const int rem = sv - s2;
    s1 = (rem + 1) / 2;
    s0 = rem - s1;
}

static size_t tile_lds_fixed(int C, int sv, int G, bool lean)
{
    const size_t TV = (size_t)1 << sv;
    if (lean)       // float4 path: no factor of the old value, no touched flags, group arrays sized by G
        return TV * C * 4 + TV * 4 + 2 * (size_t)(G + 1) * 4 + (MAX_CHUNK + 1) * 4 + (4 + TILE_CLASSES) * 4 + (size_t)((G + 1) & ~1) * 2 + 16;
    return TV * C * 4 + TV * 8 + 2 * (MAX_GROUPS + 1) * 4 + (MAX_CHUNK + 1) * 4 + (4 + TILE_CLASSES) * 4 + MAX_GROUPS * 2 + TV + 16;
}

// frames per chunk: what fits next to the tile's deltas, at most 64 KB of accumulators
static int chunk_frames(int C, int sv, int G, bool lean)
{
    const size_t per_slot = ((size_t)1 << sv) * 16;
    const size_t fixed = tile_lds_fixed(C, sv, G, lean);
    size_t avail = fixed + per_slot <= 160 * 1024 ? 160 * 1024 - fixed : per_slot;
    if (avail > 64 * 1024) avail = 64 * 1024;
    // small tiles: four workgroups per CU (their memory and compute phases overlap) matter more than long chunks
    if (sv <= 7 && fixed + 4 * per_slot <= 40 * 1024) avail = 40 * 1024 - fixed;
    int gc = (int)(avail / per_slot);
    if (g_gc_override > 0) gc = g_gc_override;
    if (gc > MAX_CHUNK) gc = MAX_CHUNK;
    if (gc > G) gc = G;
    if (gc < 1) gc = 1;
    return gc;
}

static size_t tile_lds_bytes(int C, int sv, int gc, int G, bool lean)
{
    return tile_lds_fixed(C, sv, G, lean) + (size_t)gc * ((size_t)1 << sv) * 16;
}

struct Layout {
    size_t cursor, block_sums, ticket, slot_count, slot_ws, slot_u, slot_bits, active, light, items, rec, aux, pts, total;
    int n_keys, n_scan_blocks;
    int split_slots, split_items;      // 0: no split tiles (sequential groups, or MF_SPLIT=0)
    long long cap;
};

// a tile with more than split_min() records (single group) is cut into parts of about split_part()
// records; MF_SPLIT_MIN / MF_SPLIT_PART override them for experiments
static int split_min()
{
    static const int v = env_int("MF_SPLIT_MIN", 64, 1 << 28, 8192);
    return v;
}
static int split_part()
{
    static const int v = env_int("MF_SPLIT_PART", 64, 1 << 28, 4096);
    return v;
}

static bool split_enabled()
{
    static const bool on = env_int("MF_SPLIT", 0, 1, 1) != 0;
    return on;
}

static bool make_layout(const mf_grid *g, long long n_points, int G, int s0, int s1, int s2, Layout &L,
                        int &nt0, int &nt1, int &nt2)
{
    nt0 = (g->size0 + (1 << s0) - 1) >> s0;
    nt1 = (g->size1 + (1 << s1) - 1) >> s1;
    nt2 = (g->size2 + (1 << s2) - 1) >> s2;
    const long long n_keys = (long long)nt0 * nt1 * nt2 * G;
    const long long cap = n_points * 8;
    if (n_keys + 1 > 0x7fffff00LL || cap > 0x7fffff00LL) return false;
    L.n_keys = (int)n_keys;
    L.cap = cap;
    L.n_scan_blocks = (int)((n_keys + 1 + SCAN_TILE - 1) / SCAN_TILE);
    size_t off = 0;
    L.cursor = off; off = align_up(off + (size_t)(n_keys + 1) * 4, 256);
    L.block_sums = off; off = align_up(off + (size_t)L.n_scan_blocks * 4, 256);
    L.ticket = off; off = align_up(off + 512, 256);
    // split-tile scratch (zeroed with the counters): a tile qualifies with > SPLIT_MIN of the <= cap records
    L.split_slots = L.split_items = 0;
    // the single-pass path is for frame-sized calls; a large merged batch is the tile kernel's throughput regime
    if (G == 1 && split_enabled() && cap > split_min() && n_points <= SINGLE_MAX_POINTS) {
        long long slots = cap / split_min() + 1;
        if (slots > 128) slots = 128;
        const size_t TV = (size_t)1 << (s0 + s1 + s2);
        L.split_slots = (int)slots;
        L.split_items = (int)(cap / split_part() + (n_keys / G) + slots + 1);    // parts + whole tiles
        L.slot_count = off; off = align_up(off + (size_t)slots * 4, 256);
        L.slot_ws = off; off = align_up(off + (size_t)slots * TV * 24, 256);
        L.slot_u = off; off = align_up(off + (size_t)slots * TV * g->channels * 8, 256);
        L.slot_bits = off; off = align_up(off + (size_t)slots * ((TV * g->channels + 31) / 32) * 4, 256);
    } else {
        L.slot_count = L.slot_ws = L.slot_u = L.slot_bits = off;
    }
    L.active = off; off = align_up(off + (size_t)(n_keys / G) * TILE_CLASSES * 4, 256);
    L.light = off; off = align_up(off + (G >= 2 ? (size_t)(n_keys / G) * CELL_CLASSES * 16 : 0), 256);   // work items of fuse_cells_kernel (sequential calls)
    L.items = off; off = align_up(off + (size_t)L.split_items * 8, 256);
    L.rec = off; off = align_up(off + (size_t)cap * 16, 256);
    L.aux = off; off = align_up(off + (size_t)cap * 4, 256);
    L.pts = off; off = align_up(off + (size_t)n_points * 16, 256);      // binned pixels, count -> scatter
    L.total = off;
    return true;
}

static int check_grid(const mf_grid *g, bool need_bins)
{
    if (!g) return fail(MF_ERR_INVALID, "grid is NULL");
    if (g->struct_size != sizeof(mf_grid))
        return fail(MF_ERR_INVALID, "mf_grid.struct_size is %u, this library (ABI %d) expects %zu: the binding was "
                    "written against another version of include/massfuse.h", g->struct_size, MF_ABI_VERSION, sizeof(mf_grid));
    if (g->size0 < 1 || g->size1 < 1 || g->size2 < 1 || g->size0 > 1024 || g->size1 > 1024 || g->size2 > 1024)
        return fail(MF_ERR_INVALID, "map dims must be in [1, 1024], got %d x %d x %d", g->size0, g->size1, g->size2);
    if (g->channels < 1 || g->channels > 8192)
        return fail(MF_ERR_INVALID, "channels must be in [1, 8192], got %d", g->channels);
    if (!g->map) return fail(MF_ERR_INVALID, "grid->map is NULL");
    if (need_bins) {
        if (!g->bins_x || !g->bins_y || !g->bins_z) return fail(MF_ERR_INVALID, "bin edge pointer is NULL");
        if (g->n_edges_x != g->size1 + 1 || g->n_edges_y != g->size0 + 1 || g->n_edges_z != g->size2 + 1)
            return fail(MF_ERR_INVALID, "edge counts (%d, %d, %d) must be (size1+1, size0+1, size2+1) = (%d, %d, %d)",
                        g->n_edges_x, g->n_edges_y, g->n_edges_z, g->size1 + 1, g->size0 + 1, g->size2 + 1);
    }
    return MF_OK;
}

static int check_frames(const mf_frames *f, int C)
{
    if (!f) return fail(MF_ERR_INVALID, "frames is NULL");
    if (f->struct_size != sizeof(mf_frames))
        return fail(MF_ERR_INVALID, "mf_frames.struct_size is %u, this library (ABI %d) expects %zu: the binding was "
                    "written against another version of include/massfuse.h", f->struct_size, MF_ABI_VERSION, sizeof(mf_frames));
    if (f->n_frames < 1 || f->height < 1 || f->width < 1)
        return fail(MF_ERR_INVALID, "n_frames/height/width must be positive");
    if (!f->cam_rays || !f->poses || !f->depth) return fail(MF_ERR_INVALID, "cam_rays/poses/depth pointer is NULL");
    if (f->poses_on_host != 0 && f->poses_on_host != 1) return fail(MF_ERR_INVALID, "poses_on_host must be 0 or 1");
    if (f->poses_on_host && f->n_frames != 1)
        return fail(MF_ERR_INVALID, "poses in host memory are for single-frame calls (n_frames = %d)", f->n_frames);
    if (f->feat_kind < MF_FEAT_ONES || f->feat_kind > MF_FEAT_DENSE_F32)
        return fail(MF_ERR_INVALID, "unknown feat_kind %d", f->feat_kind);
    if (f->feat_kind == MF_FEAT_ONES) {
        if (C != 1) return fail(MF_ERR_INVALID, "MF_FEAT_ONES needs channels == 1, got %d", C);
    } else {
        if (!f->feat) return fail(MF_ERR_INVALID, "feat pointer is NULL");
        if (f->feat_height < 1 || f->feat_width < 1 || f->height % f->feat_height || f->width % f->feat_width)
            return fail(MF_ERR_INVALID, "feature resolution %dx%d must divide the camera resolution %dx%d",
                        f->feat_height, f->feat_width, f->height, f->width);
    }
    return MF_OK;
}

static void fill_grid(FuseParams &P, const mf_grid *g)
{
    P.size0 = g->size0; P.size1 = g->size1; P.size2 = g->size2; P.C = g->channels;
    P.bins.b0 = g->bins_x; P.bins.b1 = g->bins_y; P.bins.b2 = g->bins_z;
    P.bins.n0 = g->n_edges_x; P.bins.n1 = g->n_edges_y; P.bins.n2 = g->n_edges_z;
    P.map = g->map;
    P.magicC = g->channels == 1 ? 0u : (unsigned)(((1ull << 32) + g->channels - 1) / g->channels);
}

static void fill_frames(FuseParams &P, const mf_frames *f)
{
    P.n_frames = f->n_frames; P.H = f->height; P.W = f->width;
    P.cam = f->cam_rays; P.poses = f->poses; P.depth = f->depth; P.feat = f->feat;
    P.pose_inline = 0;
    if (f->poses_on_host) {                 // (one frame: check_frames)
        P.pose_inline = 1;
        for (int k = 0; k < 12; ++k) P.pose0[k] = f->poses[k];
        P.poses = nullptr;
    }
    P.feat_kind = f->feat_kind;
    P.fh = f->feat_kind == MF_FEAT_ONES ? f->height : f->feat_height;
    P.fw = f->feat_kind == MF_FEAT_ONES ? f->width : f->feat_width;
    P.rep_y = f->height / P.fh; P.rep_x = f->width / P.fw;
    P.min_d = f->min_depth; P.max_d = f->max_depth;
    P.label_status = f->label_status;
    P.n_points = (long long)f->n_frames * f->height * f->width;
}

// Optional per-stage timing (bench.py's roofline leg): HIP events recorded on
// the caller's stream between the pipeline stages of the most recent call.
constexpr int PROF_CALLS = 256;
static bool g_profile = false;
static hipEvent_t g_ev[PROF_CALLS][6];
static bool g_ev_ready = false;
static int g_prof_calls = 0;       // profiled calls (commits) since mf_profile_enable(1)
static int g_prof_stages = 0;      // stagings since then: the k-th staging and the k-th commit are one call

static void prof_mark(int i, hipStream_t st)
{
    const int slot = i <= 3 ? g_prof_stages : g_prof_calls;
    if (g_profile && g_ev_ready && slot < PROF_CALLS) (void)hipEventRecord(g_ev[slot][i], st);
}

// The single-pass kernels (one group; class ids, ones, or dense features of few channels): LDS of a tile, and whether a call takes them
static size_t single_lds_bytes(int feat_kind, int C, int sv)
{
    const bool dense = feat_kind == MF_FEAT_DENSE_F32;
    return (feat_kind == MF_FEAT_ONES ? 0 : ((size_t)C << sv) * (dense ? 16 : 8) + ((((size_t)C << sv) + 31) / 32) * 4) + ((size_t)36 << sv) + 64;
}
static bool takes_single(const Layout &L, int feat_kind, int C, int sv)
{
    return L.split_slots > 0 && (feat_kind != MF_FEAT_DENSE_F32 || C <= SINGLE_DENSE_MAX_C) && single_lds_bytes(feat_kind, C, sv) <= 80 * 1024;
}

// Several maps updated from the same frames (mf_fuse_frame_maps): the first map's call buckets the points for all
// of them (role 0: it also fills their per-record words and checks their class ids), a further map (role 1) takes
// cursor and records from the first one's workspace, lists its tiles in its own and runs its tile kernels, on a
// stream of its own, behind the first map's scatter (`scattered`).
struct MultiCtx {
    int role;                  // 0 first map, 1 further map, 2 further map, planning only: its ListMap is wanted, nothing is issued
    hipEvent_t scattered;      // recorded behind the first map's scatter_kernel
    int n_lists;               // role 0: the further maps' list parameters (its tile_list_kernel makes all lists)
    ListMap lists[MAX_EXTRA_MAPS];
    ListMap *plan;             // role 2: filled in
    // role 1
    int *cursor;
    uint4 *rec;
    const int *nonempty;
    const int *abort;
    const unsigned *absmax;
    int s0, s1, s2;            // the tile shape the records were bucketed on
};

// The probe's verdict on the host.  A call in a tile-local format reads the probe's two counts back (8 bytes, a wait of
// ~30 us for the memset and the probe on the call's stream) and launches ONLY the kernels of the format they ask for: the
// kernels of the other formats need their LDS to be placed even when all they do is return (bucket_agg_kernel's 27 KB do
// not fit beside the headline's commit at all: its no-op launch waited for the tile kernel to finish, measured), and the
// tile kernels that are not chosen no longer start and exit.  The verdict of a staging call is kept for the commit that
// follows on the same workspace (the pair's own state, not history: a commit that finds none launches every variant
// and lets the mode word pick, as before).  MF_PROBE_SYNC=0: no wait - every variant of contributions / records is launched
// and the probe picks on the device; aggregated entries are then not offered.
static std::mutex g_verdict_mu;
static std::unordered_map<const void *, int> g_verdict;       // workspace -> FMT_* of the staging call in flight
static int *probe_words()
{
    static thread_local int *w = nullptr;
    if (!w && hipHostMalloc((void **)&w, 16, hipHostMallocDefault) != hipSuccess) w = nullptr;
    return w;
}

template <int FRONT>
// phase: 1 = stage (bucket the points: memset, count, scan, tile list, scatter; the map is not touched),
//        2 = commit (the tile kernels, on a workspace staged with the same arguments), 3 = both
static int run_pipeline(FuseParams &P, const mf_grid *grid, void *workspace, size_t workspace_bytes, hipStream_t st,
                        int phase = 3, const MultiCtx *mc = nullptr)
{
    if (P.G < 1 || P.G > MAX_GROUPS)
        return fail(MF_ERR_INVALID, "at most %d sequential frames per call, got %d", MAX_GROUPS, P.G);
    if (P.n_points == 0) return MF_OK;
    // tile shape: 4 x 4 x 8 for sequential frames of class ids / ones (the all-integer tile kernels), else by
    // the LDS budget of fuse_tiles_kernel; a function of the arguments only (a commit on its own agrees with its
    // staging call, a repeated run with itself)
    const bool dense_tiles = FRONT == 0 && P.G >= 2 && int_tiles_ok(grid, P.G, P.feat_kind, P.iw);
    choose_tile(grid, P.G, dense_tiles, P.s0, P.s1, P.s2);
    // tile-local entries (contributions): only the all-integer kernels read them, and they take every call bucketed on
    // their 4 x 4 x 8 tiles (an MF_TILE override of the shape keeps the point records that fuse_tiles_kernel reads)
    P.meta = dense_tiles && P.s0 == 2 && P.s1 == 2 && P.s2 == 3 ? 1 : 0;
    Layout L;
    if (!make_layout(grid, P.n_points, P.G, P.s0, P.s1, P.s2, L, P.nt0, P.nt1, P.nt2))
        return fail(MF_ERR_INVALID, "problem too large for 32-bit bucket offsets (points %lld, groups %d)",
                    P.n_points, P.G);
    if (!workspace || workspace_bytes < L.total)
        return fail(MF_ERR_WORKSPACE, "workspace of %zu bytes given, %zu needed", workspace_bytes, L.total);
    if (((uintptr_t)workspace & 255) != 0) return fail(MF_ERR_INVALID, "workspace must be 256-byte aligned");
    char *ws = (char *)workspace;
    P.cursor = (int *)(ws + L.cursor);
    P.block_sums = (int *)(ws + L.block_sums);
    P.ticket = (int *)(ws + L.ticket);
    P.active = (int *)(ws + L.active);
    P.rec = (uint4 *)(ws + L.rec);
    P.pts = (uint4 *)(ws + L.pts);
    P.aux = (uint32_t *)(ws + L.aux);
    P.n_tiles = P.nt0 * P.nt1 * P.nt2;
    P.n_keys = L.n_keys;
    const bool follower = mc && mc->role >= 1;
    if (follower) {
        if (P.s0 != mc->s0 || P.s1 != mc->s1 || P.s2 != mc->s2) return fail(MF_ERR_INVALID, "maps of one call need one tile shape");
        P.cursor = mc->cursor; P.rec = mc->rec;
    }

    // single-pass path: one group, class ids or ones (dense features keep the tile kernel)
    const bool dense = P.feat_kind == MF_FEAT_DENSE_F32;
    const size_t single_lds = single_lds_bytes(P.feat_kind, P.C, P.s0 + P.s1 + P.s2);
    const bool single = takes_single(L, P.feat_kind, P.C, P.s0 + P.s1 + P.s2);
    const dim3 bin_blocks = FRONT == 0 ? dim3((unsigned)(((P.H + PATCH - 1) / PATCH) * ((P.W + PATCH - 1) / PATCH)), (unsigned)P.n_frames)
                                       : dim3((unsigned)((P.n_points + BIN_THREADS - 1) / BIN_THREADS));
    // ---- configuration of the tile kernel (needed by both halves: its grid size seeds the ticket counter) ----
    const int sv = P.s0 + P.s1 + P.s2;
    P.vec4 = ((uintptr_t)P.map % 16 == 0) && ((P.C << P.s2) % 4 == 0) && (P.size2 % (1 << P.s2) == 0) &&
             ((size_t)P.size0 * P.size1 * P.size2 * P.C < ((size_t)1 << 34));
    P.gc = chunk_frames(P.C, sv, P.G, P.vec4 != 0);
    // a commit on its own is meant to run beside the staging kernels of the next batch: one frame less per
    // chunk leaves 8 KB of LDS per CU for their workgroups (with all of it taken they cannot start at all)
    if (phase != 3 && P.gc > 2) P.gc -= 1;
    const size_t lds = tile_lds_bytes(P.C, sv, P.gc, P.G, P.vec4 != 0);
    const DeviceInfo &dev = device_info();
    if (lds > (size_t)dev.lds_per_cu)
        return fail(MF_ERR_INVALID, "tile needs %zu bytes of LDS, device has %d", lds, dev.lds_per_cu);
    int nt = sv >= 9 ? 1024 : sv >= 7 ? 512 : 256;     // heavy tiles are bound by threads per tile
    // a commit on its own runs beside the bucketing kernels of the next batch, which need wave slots on every
    // CU (a CU holds 32 waves): with 12 waves instead of 16 the tile kernel is 10 % slower, the step 13 % faster
    if (phase != 3 && nt == 1024) nt = 768;
    { int a, b, c; tile_override(a, b, c, nt); }
    int per_cu = (int)((size_t)dev.lds_per_cu / lds);
    if (per_cu > 16) per_cu = 16;
    const int by_threads = 2048 / nt;
    if (per_cu > by_threads) per_cu = by_threads;
    if (per_cu < 1) per_cu = 1;
    int blocks = dev.cus * per_cu;
    static const int blocks_cap = env_int("MF_BLOCKS", 1, 1 << 20, 0);          // dev: fewer workgroups
    if (blocks_cap > 0 && blocks > blocks_cap) blocks = blocks_cap;
    if (blocks > P.n_tiles) blocks = P.n_tiles;
    // the all-integer tile kernel (class ids / ones, float4 rows, 4 x 4 x 8 tiles): launched next to the tile
    // kernel when the call was bucketed on its tiles; tile_list_kernel decides on the device which of the two runs
    static const int dense_nt = env_int("MF_DENSE_NT", 512, 1024, 512);      // 512 or 1024 (anything between counts as 512)
    static const bool dense_forced = getenv("MF_DENSE_FORCE") != nullptr;      // dev / tests: the dense kernel whatever the density
    const int dgc = dense_chunk_frames(P.C, P.G);
    const size_t dlds = dense_lds_bytes(P.C, dgc);
    // (a merged batch of several frames is one group on 4 x 4 x 8 tiles anyway: the kernel is offered to it too)
    static const int dense_min_frames = env_int("MF_DENSE_MIN_FRAMES", 1, 1 << 20, 2);   // dev
    const bool merged_batch = FRONT == 0 && P.G == 1 && P.n_frames >= dense_min_frames;
    const bool use_dense = (dense_tiles || merged_batch) && sv == DENSE_SV && P.s2 == 3 && P.feat_kind != MF_FEAT_DENSE_F32 && P.vec4 &&
                       dlds <= (size_t)dev.lds_per_cu && (P.G + dgc - 1) / dgc <= DENSE_MAX_CHUNKS;
    if (P.meta && !use_dense) return fail(MF_ERR_INVALID, "internal: a tile-local entry format was chosen for a call no integer tile kernel takes");
    const int dnt = dense_nt >= 1024 ? 1024 : 512;
    int dper = (int)((size_t)dev.lds_per_cu / dlds);
    if (dper > 2048 / dnt) dper = 2048 / dnt;
    if (dper < 1) dper = 1;
    int blocks_dense = dev.cus * dper;
    if (blocks_cap > 0 && blocks_dense > blocks_cap) blocks_dense = blocks_cap;
    if (blocks_dense > P.n_tiles) blocks_dense = P.n_tiles;
    // the compact-cell tile kernel for sparse frames (same tiles, same feature kinds, sequential frames only);
    // MF_CELLS=0 keeps it out, MF_CELLS_FORCE gives it every call it is offered (dev / tests)
    static const bool cells_on = env_int("MF_CELLS", 0, 1, 1) != 0;
    static const bool cells_forced = getenv("MF_CELLS_FORCE") != nullptr;
    int cells_cap = 0, cells_per_cu = 1;
    static const int cells_reserve = env_int("MF_CELLS_RESERVE", 0, 128 * 1024, 8192);          // dev: LDS per CU left to the bucketing kernels beside a commit
    static const int cells_most = env_int("MF_CELLS_MOST", 1, 3, 2);                            // dev: workgroups per CU beside them
    const bool use_cells = cells_on && dense_tiles && use_dense && sv == CELLS_SV && P.s2 == 3 && P.s0 == 2 && P.G >= 2 &&
                           // (a commit on its own runs beside the bucketing kernels of the next batch: they need a few KB of LDS per CU)
                           // ... and wave slots / registers: two workgroups per CU beside them (headline, pipelined: 24.5 k frames/s
                           // against 20.3 k with three), three when the call has the chip to itself (equal alone: 1.84 / 1.89 ms)
                           cells_config(P.C, dev.lds_per_cu, phase == 3 ? 0 : cells_reserve, phase == 3 ? 3 : cells_most, cells_cap, cells_per_cu);
    int blocks_cells = dev.cus * cells_per_cu;
    if (blocks_cap > 0 && blocks_cells > blocks_cap) blocks_cells = blocks_cap;
    if (blocks_cells > P.n_tiles) blocks_cells = P.n_tiles;

    const int list_dense_tv = (use_dense || use_cells ? 1 << sv : 0) | (use_dense ? 1 << 20 : 0) | (use_cells ? 1 << 21 : 0) |
                              (use_dense && dense_forced ? 1 << 22 : 0);
    // entry format of a call in a tile-local format: the probe's choice, unless only fuse_dense_kernel is offered or a
    // kernel is forced (MF_DENSE_FORCE / MF_CELLS_FORCE: dev / tests)
    // MF_FORMAT=contributions / records overrides the probe per call (read at every call: tests set it per case)
    // (MF_FORMAT=aggregated: the aggregated entries of bucket_agg_kernel for every call; MF_AGG=0: real scenes keep records +
    // fuse_dense_kernel instead of aggregated entries + fuse_cells_kernel)
    const char *fmt_env = getenv("MF_FORMAT");
    static const bool agg_on = env_int("MF_AGG", 0, 1, AGG_DEFAULT) != 0;
    static const bool probe_sync = env_int("MF_PROBE_SYNC", 0, 1, 1) != 0;
    const int by_probe = agg_on && probe_sync ? 4 : 0;
    const int fmt_env_force = !fmt_env ? by_probe : fmt_env[0] == 'c' ? 1 : fmt_env[0] == 'r' ? 2 : fmt_env[0] == 'a' ? 3 : by_probe;
    P.fmt_force = !P.meta ? 0 : (!use_cells || dense_forced) ? 2 : cells_forced ? 1 : fmt_env_force;
    // (bucket_agg_kernel's keys hold the tile in 17 bits: larger maps keep records for real scenes)
    if (P.n_tiles > (1 << 17) && P.fmt_force >= 3) P.fmt_force = P.fmt_force == 3 ? 2 : 0;
    // the format when the host knows it (forced, or read back from the probe below): FMT_*, else -1
    int fmt_known = !P.meta ? -1 : P.fmt_force == 1 ? FMT_CONTRIB : P.fmt_force == 2 ? FMT_RECORDS : P.fmt_force == 3 ? FMT_AGG : -1;
    if (P.meta && fmt_known < 0 && phase == 2 && probe_sync) {        // a commit on its own: what its staging call found
        std::lock_guard<std::mutex> lock(g_verdict_mu);
        const auto it = g_verdict.find(workspace);
        if (it != g_verdict.end()) fmt_known = it->second;
    }
    bool agg_offered = P.meta && (fmt_known >= 0 ? fmt_known == FMT_AGG : P.fmt_force == 4);
    const dim3 agg_blocks((unsigned)(((long long)bin_blocks.x * bin_blocks.y + AGG_ITEMS - 1) / AGG_ITEMS));      // (AGG_ITEMS patches per workgroup)
    // fixed-point fraction bits of the W / S2 sums: the per-voxel, per-frame sum of weights is
    // below (points per group) * (1 + 1e-9), and must stay below 2^63
    int fx_shift;
    {
        long long per_group = P.G > 1 ? (P.n_points + P.G - 1) / P.G : P.n_points;
        int bits = 1; while ((1ll << bits) <= per_group) ++bits;
        fx_shift = 62 - bits; if (fx_shift > 50) fx_shift = 50;
    }

    ListMap LM;
    LM.ticket = P.ticket; LM.active = P.active; LM.items = (int *)(ws + L.items);
    LM.split_min = single ? (dense ? 0x7fffffff : split_min()) : 0; LM.split_slots = L.split_slots;
    static const int single_min_mean = env_int("MF_SINGLE_MIN_MEAN", 0, 1 << 20, SINGLE_MIN_MEAN);      // dev
    // (the maps of a shared call always take the single-pass kernels when they can: a tile kernel launched only to find
    // that it has nothing to do waits for a whole CU's LDS beside the other maps' kernels, 40 us in their way)
    LM.min_mean = P.feat_kind == MF_FEAT_ONES || mc ? 0 : single_min_mean;
    LM.first_ticket = 4 * blocks; LM.dense_tv = list_dense_tv; LM.first_ticket_dense = 4 * blocks_dense; LM.first_ticket_cells = blocks_cells;
    LM.abort = follower ? mc->abort : P.ticket + ABORT_SLOT;
    LM.items4 = (int4 *)(ws + L.light);
    if (mc && mc->role == 2) { *mc->plan = LM; return MF_OK; }
    if (follower) {
        // the first map's kernels have zeroed this map's counters and split scratch, found its feature range and
        // made its tile list: its tile kernels start behind the scatter
        MF_HIP_CHECK(hipStreamWaitEvent(st, mc->scattered, 0));
    } else if (phase & 1) {
    prof_mark(0, st);
    // cursor .. ticket (.. split scratch) are contiguous: one memset
    MF_HIP_CHECK(hipMemsetAsync(ws + L.cursor, 0, (single ? L.active : L.slot_count) - L.cursor, st));
    if (FRONT == 0 && P.meta && (P.fmt_force == 0 || P.fmt_force == 4)) {
        // which entry format the call's data asks for: eight patches per frame are looked at (use_contributions)
        const int n_probe = 8;
        hipLaunchKernelGGL(probe_kernel, dim3(n_probe, (unsigned)P.n_frames), dim3(BIN_THREADS), 0, st, P, n_probe);
        MF_LAUNCH_CHECK("probe_kernel");
        int *pw = probe_sync ? probe_words() : nullptr;
        if (pw) {
            static_assert(PROBE_TILES == PROBE_POINTS + 1, "the probe's two counts are read back together");
            MF_HIP_CHECK(hipMemcpyAsync(pw, P.ticket + PROBE_POINTS, 2 * sizeof(int), hipMemcpyDeviceToHost, st));
            MF_HIP_CHECK(hipStreamSynchronize(st));
            const bool sparse = pw[0] < PROBE_DENSE_RATIO * pw[1];                  // (entry_format)
            fmt_known = sparse ? FMT_CONTRIB : P.fmt_force == 4 ? FMT_AGG : FMT_RECORDS;
            P.fmt_force = fmt_known == FMT_CONTRIB ? 1 : fmt_known == FMT_AGG ? 3 : 2;    // the kernels are told, they do not look again
            agg_offered = fmt_known == FMT_AGG;
        }
    }
    if (P.meta && phase == 1 && probe_sync) {
        std::lock_guard<std::mutex> lock(g_verdict_mu);
        if (fmt_known >= 0) g_verdict[workspace] = fmt_known; else g_verdict.erase(workspace);
    }
    P.absmax = FRONT == 0 && single && dense ? (unsigned *)(P.ticket + FEAT_ABSMAX) : nullptr;     // found by count_kernel itself
    if (fmt_known != FMT_AGG) {
        hipLaunchKernelGGL(count_kernel<FRONT>, bin_blocks, dim3(BIN_THREADS), 0, st, P);
        MF_LAUNCH_CHECK("count_kernel");
    }
    if (FRONT == 0 && agg_offered) {
        hipLaunchKernelGGL(bucket_agg_kernel<false>, agg_blocks, dim3(BIN_THREADS), 0, st, P);    // returns at once unless the call's format is FMT_AGG (count_kernel returns then)
        MF_LAUNCH_CHECK("bucket_agg_kernel<count>");
    }
    if (FRONT != 0 && single && dense) {
        const long long nf = FRONT == 0 ? (long long)P.n_frames * P.fh * P.fw * P.C : P.n_points * P.C;
        hipLaunchKernelGGL(feat_absmax_kernel, dim3((unsigned)((nf + 256 * 16 - 1) / (256 * 16) > 1024 ? 1024 : (nf + 256 * 16 - 1) / (256 * 16))),
                           dim3(256), 0, st, (const float *)P.feat, nf, P.ticket + FEAT_ABSMAX);
        MF_LAUNCH_CHECK("feat_absmax_kernel");
    }
    prof_mark(1, st);
    hipLaunchKernelGGL(scan_sums_kernel, dim3(L.n_scan_blocks), dim3(SCAN_THREADS), 0, st,
                       (const int *)P.cursor, P.n_keys + 1, P.block_sums);
    MF_LAUNCH_CHECK("scan_sums_kernel");
    hipLaunchKernelGGL(scan_apply_kernel, dim3(L.n_scan_blocks), dim3(SCAN_THREADS), 0, st,
                       P.cursor, P.n_keys + 1, (const int *)P.block_sums, P.ticket + SPLIT_NONEMPTY);
    MF_LAUNCH_CHECK("scan_apply_kernel");
    {
        ListParams LP;
        LP.cursor = P.cursor; LP.n_tiles = P.n_tiles; LP.G = P.G; LP.split_part = split_part();
        LP.nt1 = P.nt1; LP.nt2 = P.nt2; LP.s0 = P.s0; LP.s1 = P.s1; LP.s2 = P.s2; LP.meta = P.meta; LP.fmt_force = P.fmt_force;
        LP.nonempty = P.ticket + SPLIT_NONEMPTY;
        LP.map[0] = LM;
        const int n_lists = mc && mc->role == 0 ? mc->n_lists : 0;
        for (int m = 0; m < n_lists; ++m) LP.map[1 + m] = mc->lists[m];
        hipLaunchKernelGGL(tile_list_kernel, dim3((P.n_tiles + 255) / 256, 1 + n_lists), dim3(256), 0, st, LP);
        MF_LAUNCH_CHECK("tile_list_kernel");
    }
    prof_mark(2, st);
    if (fmt_known != FMT_AGG) {
        hipLaunchKernelGGL(scatter_kernel<FRONT>, bin_blocks, dim3(BIN_THREADS), 0, st, P);
        MF_LAUNCH_CHECK("scatter_kernel");
    }
    if (FRONT == 0 && agg_offered) {
        hipLaunchKernelGGL(bucket_agg_kernel<true>, agg_blocks, dim3(BIN_THREADS), 0, st, P);
        MF_LAUNCH_CHECK("bucket_agg_kernel<scatter>");
    }
    if (mc) MF_HIP_CHECK(hipEventRecord(mc->scattered, st));
    prof_mark(3, st);
    if (g_profile && g_ev_ready && g_prof_stages < PROF_CALLS) ++g_prof_stages;
    }
    if (!(phase & 2)) return MF_OK;
    prof_mark(4, st);

    const int kind = P.feat_kind == MF_FEAT_ONES ? 0 : (P.feat_kind == MF_FEAT_DENSE_F32 ? 2 : 1);
    void (*kern)(TileParams);
    if (nt <= 64) kern = kind == 0 ? fuse_tiles_kernel<0, 64> : kind == 1 ? fuse_tiles_kernel<1, 64> : fuse_tiles_kernel<2, 64>;
    else if (nt <= 256) kern = kind == 0 ? fuse_tiles_kernel<0, 256> : kind == 1 ? fuse_tiles_kernel<1, 256> : fuse_tiles_kernel<2, 256>;
    else kern = kind == 0 ? fuse_tiles_kernel<0, 1024> : kind == 1 ? fuse_tiles_kernel<1, 1024> : fuse_tiles_kernel<2, 1024>;
    static const bool stamps = getenv("MF_STAMPS") != nullptr;
    if (stamps && nt > 256) kern = kind == 0 ? fuse_tiles_kernel<0, 1024, true> : kind == 1 ? fuse_tiles_kernel<1, 1024, true> : fuse_tiles_kernel<2, 1024, true>;
    if (stamps && nt <= 256 && nt > 64) kern = kind == 0 ? fuse_tiles_kernel<0, 256, true> : kind == 1 ? fuse_tiles_kernel<1, 256, true> : fuse_tiles_kernel<2, 256, true>;
    if (stamps) { unsigned long long z[16] = {}; MF_HIP_CHECK(hipMemcpyToSymbol(HIP_SYMBOL(g_stamps), z, sizeof(z))); }
    {
        // the dynamic-LDS limit of a kernel is raised once (per size): not a per-call cost
        static std::mutex mu;
        static std::unordered_map<const void *, size_t> granted;
        std::lock_guard<std::mutex> lock(mu);
        size_t &have = granted[(const void *)kern];
        if (have < lds) {
            MF_HIP_CHECK(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            have = lds;
        }
    }
    TileParams T;
    T.size0 = P.size0; T.size1 = P.size1; T.size2 = P.size2; T.C = P.C; T.map = P.map; T.feat = P.feat;
    T.G = P.G; T.iw = P.iw; T.s0 = P.s0; T.s1 = P.s1; T.s2 = P.s2; T.nt1 = P.nt1; T.nt2 = P.nt2;
    T.n_tiles = P.n_tiles; T.magicC = P.magicC; T.gc = P.gc; T.vec4 = P.vec4; T.cursor = P.cursor;
    T.fx_shift = fx_shift;
    T.ticket = P.ticket; T.active = P.active; T.rec = P.rec; T.aux = P.aux;
    T.ctr = P.ticket;
    T.cells_cap = cells_cap;
    T.meta = P.meta;
    T.light = nullptr;
    // with ones features every tile of a single-group call goes to the single-pass kernel (tile_list_kernel,
    // min_mean = 0): nothing is listed for the tile kernel, whose launch is skipped
    // (a call in a tile-local format goes to one of the integer kernels; which one the host knows: fmt_known)
    const bool pick = P.meta && fmt_known >= 0;
    if (!(single && LM.min_mean == 0) && !pick) {
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(nt), lds, st, T);  // returns at once when the call went to another kernel
        MF_LAUNCH_CHECK("fuse_tiles_kernel");
    }
    if (use_dense && (!pick || fmt_known == FMT_RECORDS)) {
        void (*dk)(TileParams);
        if (P.meta) {
            if (dnt >= 1024) dk = stamps ? (kind == 0 ? fuse_dense_kernel<0, 1024, true, true> : fuse_dense_kernel<1, 1024, true, true>)
                                         : (kind == 0 ? fuse_dense_kernel<0, 1024, true> : fuse_dense_kernel<1, 1024, true>);
            else dk = stamps ? (kind == 0 ? fuse_dense_kernel<0, 512, true, true> : fuse_dense_kernel<1, 512, true, true>)
                             : (kind == 0 ? fuse_dense_kernel<0, 512, true> : fuse_dense_kernel<1, 512, true>);
        } else {
            if (dnt >= 1024) dk = stamps ? (kind == 0 ? fuse_dense_kernel<0, 1024, false, true> : fuse_dense_kernel<1, 1024, false, true>)
                                         : (kind == 0 ? fuse_dense_kernel<0, 1024, false> : fuse_dense_kernel<1, 1024, false>);
            else dk = stamps ? (kind == 0 ? fuse_dense_kernel<0, 512, false, true> : fuse_dense_kernel<1, 512, false, true>)
                             : (kind == 0 ? fuse_dense_kernel<0, 512, false> : fuse_dense_kernel<1, 512, false>);
        }
        {
            static std::mutex mu3;
            static std::unordered_map<const void *, size_t> granted3;
            std::lock_guard<std::mutex> lock(mu3);
            size_t &have = granted3[(const void *)dk];
            if (have < dlds) {
                MF_HIP_CHECK(hipFuncSetAttribute((const void *)dk, hipFuncAttributeMaxDynamicSharedMemorySize, (int)dlds));
                have = dlds;
            }
        }
        TileParams S = T;
        S.gc = dgc;
        S.ctr = P.ticket + TICKET_DENSE;
        hipLaunchKernelGGL(dk, dim3(blocks_dense), dim3(dnt), dlds, st, S);   // returns at once unless tile_list_kernel chose it
        MF_LAUNCH_CHECK("fuse_dense_kernel");
    }
    if (use_cells) {
        const int f4 = (32 * P.C + CELLS_NT - 1) / CELLS_NT;                 // float4s per thread and tile
        void (*ck)(TileParams);
        if (kind == 0) ck = fuse_cells_kernel<0, 1>;
        else ck = f4 <= 1 ? fuse_cells_kernel<1, 1> : f4 <= 2 ? fuse_cells_kernel<1, 2> : f4 <= 4 ? fuse_cells_kernel<1, 4> : f4 <= 7 ? fuse_cells_kernel<1, 7> : fuse_cells_kernel<1, 8>;
        if (stamps && kind == 1 && f4 == 7) ck = fuse_cells_kernel<1, 7, true>;      // (dev: the headline shape)
        // A commit on its own runs two workgroups per CU (beside the next batch's bucketing kernels): registers for 2,048
        // instead of 1,024 of a tile's contributions, i.e. 80 % instead of 60 % of the headline's tiles never stream an entry a
        // second and third time.  Five alternating runs of 60 steps: tile kernel 2.08-2.15 -> 2.00-2.04 ms as timed, step
        // 2.54 -> 2.49 ms; with three workgroups per CU (the call that has the chip to itself) 161 registers are too many
        // (1.80-1.87 -> 1.88 ms): that call keeps 1,024.  MF_CELLS_CR8=0: off (dev).
        static const bool cr8 = env_int("MF_CELLS_CR8", 0, 1, 1) != 0;
        if (cr8 && phase != 3 && kind == 1 && f4 == 7 && !stamps) ck = fuse_cells_kernel<1, 7, false, false, CELLS_CR_COMMIT>;
        const size_t clds = cells_lds_bytes(P.C, cells_cap);
        static std::mutex mu4;
        static std::unordered_map<const void *, size_t> granted4;
        {
            std::lock_guard<std::mutex> lock(mu4);
            size_t &have = granted4[(const void *)ck];
            if (have < clds) {
                MF_HIP_CHECK(hipFuncSetAttribute((const void *)ck, hipFuncAttributeMaxDynamicSharedMemorySize, (int)clds));
                have = clds;
            }
        }
        TileParams S = T;
        S.ctr = P.ticket + TICKET_CELLS;
        S.light = (const int4 *)(ws + L.light);
        // (Only when the host does not know the format are both variants launched.  The one that is NOT chosen returns at
        // once, but its workgroups have to be placed like any others: launched behind the chosen kernel it waited for that
        // kernel's last workgroups and then for 77 KB of LDS per workgroup beside the next batch's bucketing kernels, +0.3 ms
        // between the commit's events on the headline, measured.  The variant for real scenes therefore goes first.)
        if (agg_offered) {
            void (*ak)(TileParams);
            if (kind == 0) ak = fuse_cells_kernel<0, 1, false, true>;
            else ak = f4 <= 1 ? fuse_cells_kernel<1, 1, false, true> : f4 <= 2 ? fuse_cells_kernel<1, 2, false, true> : f4 <= 4 ? fuse_cells_kernel<1, 4, false, true>
                    : f4 <= 7 ? fuse_cells_kernel<1, 7, false, true> : fuse_cells_kernel<1, 8, false, true>;
            if (stamps && kind == 1 && f4 == 7) ak = fuse_cells_kernel<1, 7, true, true>;
            {
                std::lock_guard<std::mutex> lock(mu4);
                size_t &have = granted4[(const void *)ak];
                if (have < clds) {
                    MF_HIP_CHECK(hipFuncSetAttribute((const void *)ak, hipFuncAttributeMaxDynamicSharedMemorySize, (int)clds));
                    have = clds;
                }
            }
            hipLaunchKernelGGL(ak, dim3(blocks_cells), dim3(CELLS_NT), clds, st, S);   // aggregated entries: returns at once unless the call's format is FMT_AGG
            MF_LAUNCH_CHECK("fuse_cells_kernel<AGG>");
        }
        if (!pick || fmt_known == FMT_CONTRIB) {
            hipLaunchKernelGGL(ck, dim3(blocks_cells), dim3(CELLS_NT), clds, st, S);   // returns at once unless tile_list_kernel chose it
            MF_LAUNCH_CHECK("fuse_cells_kernel");
        }
    }
    if (single) {
        SingleParams S;
        S.size0 = P.size0; S.size1 = P.size1; S.size2 = P.size2; S.C = P.C; S.map = P.map; S.iw = P.iw;
        S.s0 = P.s0; S.s1 = P.s1; S.s2 = P.s2; S.nt1 = P.nt1; S.nt2 = P.nt2; S.magicC = P.magicC; S.fx_shift = T.fx_shift;
        S.cursor = P.cursor; S.ticket = P.ticket; S.items = (const int *)(ws + L.items); S.rec = P.rec; S.aux = P.aux;
        S.feat = (const float *)P.feat;
        S.absmax = follower ? mc->absmax : (const unsigned *)(P.ticket + FEAT_ABSMAX);
        S.slot_count = (int *)(ws + L.slot_count); S.slot_ws = (unsigned long long *)(ws + L.slot_ws);
        S.slot_u = (unsigned long long *)(ws + L.slot_u);
        S.slot_bits = (unsigned *)(ws + L.slot_bits);
        const size_t slds = single_lds;
        void (*sk)(SingleParams) = dense ? fuse_single_dense_kernel<512>
                                   : stamps ? (kind == 0 ? fuse_single_kernel<0, 512, true> : fuse_single_kernel<1, 512, true>)
                                            : (kind == 0 ? fuse_single_kernel<0, 512> : fuse_single_kernel<1, 512>);
        if (slds > (size_t)dev.lds_per_cu) return fail(MF_ERR_INVALID, "single-pass tile needs %zu bytes of LDS", slds);
        {
            static std::mutex mu2;
            static std::unordered_map<const void *, size_t> granted2;
            std::lock_guard<std::mutex> lock(mu2);
            size_t &have = granted2[(const void *)sk];
            if (have < slds) {
                MF_HIP_CHECK(hipFuncSetAttribute((const void *)sk, hipFuncAttributeMaxDynamicSharedMemorySize, (int)slds));
                have = slds;
            }
        }
        int sper = (int)((size_t)dev.lds_per_cu / slds);
        if (sper > 4) sper = 4;
        if (sper < 1) sper = 1;
        hipLaunchKernelGGL(sk, dim3(dev.cus * sper), dim3(512), slds, st, S);
        MF_LAUNCH_CHECK("fuse_single_kernel");
    }
    prof_mark(5, st);
    if (g_profile && g_ev_ready && g_prof_calls < PROF_CALLS) ++g_prof_calls;
    if (stamps) {
        unsigned long long z[16];
        MF_HIP_CHECK(hipStreamSynchronize(st));
        MF_HIP_CHECK(hipMemcpyFromSymbol(z, HIP_SYMBOL(g_stamps), sizeof(z)));
        double tot = 0; for (int i = 0; i < 7; ++i) tot += (double)z[i];
        if (single) fprintf(stderr, "[MF_STAMPS] single-pass kernel: item fetch / zero / accumulate / merge+arrive / per-voxel / read-modify-write =\n");
        fprintf(stderr, "[MF_STAMPS] blocks=%d nt=%d lds=%zu gc=%d | ticket+offs %.1f%% setup %.1f%% chunk-zero %.1f%% P1 %.1f%% P2 %.1f%% P3 %.1f%% final %.1f%% | total %.3g ticks/block\n",
                blocks, nt, lds, P.gc, 100 * z[0] / tot, 100 * z[1] / tot, 100 * z[2] / tot, 100 * z[3] / tot,
                100 * z[4] / tot, 100 * z[5] / tot, 100 * z[6] / tot, tot / blocks);
        if (z[15]) fprintf(stderr, "[MF_STAMPS] cells kernel: %d workgroups, cap %d, mean %.3g ticks, slowest %.3g | tile start %.1f%% mask %.1f%% scan %.1f%% pass1 %.1f%% pass2 %.1f%% pass3 %.1f%% final %.1f%% tile end %.1f%%\n",
                           blocks_cells, cells_cap, (double)z[14] / blocks_cells, (double)z[15], 100.0 * z[0] / z[14], 100.0 * z[1] / z[14], 100.0 * z[2] / z[14], 100.0 * z[3] / z[14],
                           100.0 * z[4] / z[14], 100.0 * z[5] / z[14], 100.0 * z[6] / z[14], 100.0 * z[7] / z[14]);
        else if (z[7]) fprintf(stderr, "[MF_STAMPS] dense kernel: %d workgroups, gc %d, mean %.3g ticks, slowest %.3g | tile start %.1f%% barrier %.1f%% pass1 %.1f%% look-ups+barrier %.1f%% fetch issue %.1f%% pass2 %.1f%% pass3 %.1f%%\n", blocks_dense, dgc,
                          tot / blocks_dense, (double)z[7], 100 * z[0] / tot, 100 * z[1] / tot, 100 * z[6] / tot, 100 * z[2] / tot, 100 * z[5] / tot, 100 * z[3] / tot, 100 * z[4] / tot);
    }
    return MF_OK;
}

}  // namespace mf

using namespace mf;

extern "C" {

int mf_profile_enable(int32_t on)
{
    if (on && !g_ev_ready) {
        for (int c = 0; c < PROF_CALLS; ++c)
            for (int i = 0; i < 6; ++i) MF_HIP_CHECK(hipEventCreate(&g_ev[c][i]));
        g_ev_ready = true;
    }
    g_profile = on != 0;
    if (on) g_prof_calls = g_prof_stages = 0;
    return MF_OK;
}

int mf_profile_read(int32_t call, float *ms)
{
    if (!ms) return fail(MF_ERR_INVALID, "ms is NULL");
    if (call < 0 || call >= g_prof_calls)
        return fail(MF_ERR_INVALID, "call %d not recorded (%d profiled calls since mf_profile_enable(1))", call,
                    g_prof_calls);
    // events 0..3 bracket the staging kernels, 4..5 the tile kernels (possibly on another stream, later)
    MF_HIP_CHECK(hipEventSynchronize(g_ev[call][5]));
    for (int i = 0; i < 3; ++i) MF_HIP_CHECK(hipEventElapsedTime(&ms[i], g_ev[call][i], g_ev[call][i + 1]));
    MF_HIP_CHECK(hipEventElapsedTime(&ms[3], g_ev[call][4], g_ev[call][5]));
    ms[4] = ms[0] + ms[1] + ms[2] + ms[3];
    return g_prof_calls;
}

int mf_fuse_last_mode(const mf_grid *grid, int64_t n_points, int32_t n_groups, const void *workspace, void *stream)
{
    if (check_grid(grid, false) != MF_OK) return MF_ERR_INVALID;
    if (!workspace || n_points < 1 || n_groups < 1 || n_groups > MAX_GROUPS) return fail(MF_ERR_INVALID, "bad argument");
    int s0, s1, s2, a, b, c;
    // the layout of a call of frames (front end 0) with these arguments; class-id / ones kinds (the only ones
    // with a choice of tile kernel)
    choose_tile(grid, n_groups, n_groups >= 2 && int_tiles_ok(grid, n_groups, MF_FEAT_ONES, 0.5f), s0, s1, s2);
    Layout L;
    if (!make_layout(grid, n_points, n_groups, s0, s1, s2, L, a, b, c)) return fail(MF_ERR_INVALID, "problem too large");
    int mode = -1;
    MF_HIP_CHECK(hipMemcpyAsync(&mode, (const char *)workspace + L.ticket + MODE_SLOT * sizeof(int), sizeof(int),
                                hipMemcpyDeviceToHost, (hipStream_t)stream));
    MF_HIP_CHECK(hipStreamSynchronize((hipStream_t)stream));
    return mode;
}

size_t mf_fuse_workspace_bytes(const mf_grid *grid, int64_t n_points, int32_t n_groups)
{
    if (check_grid(grid, false) != MF_OK) return 0;
    if (n_points < 0 || n_groups < 1 || n_groups > MAX_GROUPS) {
        fail(MF_ERR_INVALID, "n_points must be >= 0 and n_groups in [1, %d]", MAX_GROUPS);
        return 0;
    }
    // whichever tile shape a call takes (it depends on what the previous call measured): the larger layout
    size_t need = 0;
    for (int dense = 0; dense < 2; ++dense) {
        int s0, s1, s2, a, b, c;
        choose_tile(grid, n_groups, dense != 0, s0, s1, s2);
        Layout L;
        if (!make_layout(grid, n_points, n_groups, s0, s1, s2, L, a, b, c)) {
            fail(MF_ERR_INVALID, "problem too large for 32-bit bucket offsets");
            return 0;
        }
        if (L.total > need) need = L.total;
    }
    return need;
}

int mf_fuse_frames(const mf_grid *grid, const mf_frames *frames, float interpolation_weight, int32_t mode,
                   void *workspace, size_t workspace_bytes, void *stream)
{
    int rc = check_grid(grid, true);
    if (rc != MF_OK) return rc;
    rc = check_frames(frames, grid->channels);
    if (rc != MF_OK) return rc;
    if (mode != MF_MODE_SEQUENTIAL && mode != MF_MODE_MERGED) return fail(MF_ERR_INVALID, "unknown mode %d", mode);
    FuseParams P = {};
    fill_grid(P, grid);
    fill_frames(P, frames);
    P.G = mode == MF_MODE_SEQUENTIAL ? frames->n_frames : 1;
    P.iw = interpolation_weight;
    return run_pipeline<0>(P, grid, workspace, workspace_bytes, (hipStream_t)stream);
}

// ---- several maps from the same frames ----
// Side streams and events of mf_fuse_frame_maps, per host thread and device (created on first use, kept).
struct MapStreams {
    int device = -1;
    hipStream_t side[MAX_EXTRA_MAPS] = {};
    hipEvent_t scattered = nullptr, done[MAX_EXTRA_MAPS] = {};
};

static int map_streams(MapStreams *&out)
{
    static thread_local std::vector<MapStreams> pool;
    int dev = 0;
    MF_HIP_CHECK(hipGetDevice(&dev));
    for (MapStreams &m : pool)
        if (m.device == dev) { out = &m; return MF_OK; }
    MapStreams m;
    m.device = dev;
    for (int i = 0; i < MAX_EXTRA_MAPS; ++i) {
        MF_HIP_CHECK(hipStreamCreateWithFlags(&m.side[i], hipStreamNonBlocking));
        MF_HIP_CHECK(hipEventCreateWithFlags(&m.done[i], hipEventDisableTiming));
    }
    MF_HIP_CHECK(hipEventCreateWithFlags(&m.scattered, hipEventDisableTiming));
    pool.push_back(m);
    out = &pool.back();
    return MF_OK;
}

int mf_fuse_frame_maps(const mf_grid *grids, const mf_frames *frames, const float *interpolation_weights, int32_t n_maps,
                       int32_t mode, void *const *workspaces, const size_t *workspace_bytes, void *stream)
{
    if (!grids || !frames || !interpolation_weights || !workspaces || !workspace_bytes) return fail(MF_ERR_INVALID, "argument array is NULL");
    if (n_maps < 1 || n_maps > 1 + MAX_EXTRA_MAPS) return fail(MF_ERR_INVALID, "1 to %d maps per call, got %d", 1 + MAX_EXTRA_MAPS, n_maps);
    if (mode != MF_MODE_SEQUENTIAL && mode != MF_MODE_MERGED) return fail(MF_ERR_INVALID, "unknown mode %d", mode);
    for (int m = 0; m < n_maps; ++m) {
        int rc = check_grid(&grids[m], true);
        if (rc != MF_OK) return rc;
        rc = check_frames(&frames[m], grids[m].channels);
        if (rc != MF_OK) return rc;
        const mf_grid &g = grids[m], &g0 = grids[0];
        const mf_frames &f = frames[m], &f0 = frames[0];
        if (g.size0 != g0.size0 || g.size1 != g0.size1 || g.size2 != g0.size2)
            return fail(MF_ERR_INVALID, "map %d is %d x %d x %d, map 0 is %d x %d x %d: the maps of one call share their voxel grid",
                        m, g.size0, g.size1, g.size2, g0.size0, g0.size1, g0.size2);
        if (f.n_frames != f0.n_frames || f.height != f0.height || f.width != f0.width || f.cam_rays != f0.cam_rays ||
            f.poses != f0.poses || f.poses_on_host != f0.poses_on_host || f.depth != f0.depth || f.min_depth != f0.min_depth ||
            f.max_depth != f0.max_depth)
            return fail(MF_ERR_INVALID, "frames[%d] differs from frames[0] in more than its features: the maps of one call "
                        "are updated from the same rays, poses and depth", m);
        for (int k = 0; k < m; ++k) {
            if (grids[k].map == g.map) return fail(MF_ERR_INVALID, "maps %d and %d are the same buffer", k, m);
            // (the lead map's kernels zero the other maps' counters while their own run on the lead's: a shared workspace is
            // silent corruption)
            const char *wa = (const char *)workspaces[k], *wb = (const char *)workspaces[m];
            if (wa && wb && wa < wb + workspace_bytes[m] && wb < wa + workspace_bytes[k])
                return fail(MF_ERR_INVALID, "the workspaces of maps %d and %d overlap: every map of a call needs one of its own", k, m);
        }
    }
    hipStream_t st = (hipStream_t)stream;
    const int G = mode == MF_MODE_SEQUENTIAL ? frames[0].n_frames : 1;
    // The map whose tile kernels take longest leads (they start right behind the scatter on the caller's stream; the
    // others cross to a side stream and back): dense features before class ids before ones, more channels first.
    mf_grid grids_o[1 + MAX_EXTRA_MAPS];
    mf_frames frames_o[1 + MAX_EXTRA_MAPS];
    float weights_o[1 + MAX_EXTRA_MAPS];
    void *workspaces_o[1 + MAX_EXTRA_MAPS];
    size_t workspace_bytes_o[1 + MAX_EXTRA_MAPS];
    {
        int order[1 + MAX_EXTRA_MAPS];
        auto load = [&](int m) {
            return (frames[m].feat_kind == MF_FEAT_DENSE_F32 ? 1 << 20 : frames[m].feat_kind == MF_FEAT_ONES ? 0 : 1 << 16) + grids[m].channels;
        };
        for (int m = 0; m < n_maps; ++m) order[m] = m;
        for (int i = 1; i < n_maps; ++i)              // (stable insertion sort, n_maps <= 4)
            for (int j = i; j > 0 && load(order[j]) > load(order[j - 1]); --j) { const int t = order[j]; order[j] = order[j - 1]; order[j - 1] = t; }
        for (int m = 0; m < n_maps; ++m) {
            grids_o[m] = grids[order[m]]; frames_o[m] = frames[order[m]]; weights_o[m] = interpolation_weights[order[m]];
            workspaces_o[m] = workspaces[order[m]]; workspace_bytes_o[m] = workspace_bytes[order[m]];
        }
        grids = grids_o; frames = frames_o; interpolation_weights = weights_o; workspaces = workspaces_o; workspace_bytes = workspace_bytes_o;
    }
    // what is shared is the bucketing of a single group; anything else (and maps whose tiles differ) is the plain loop
    bool share = n_maps > 1 && G == 1;
    int s0 = 0, s1 = 0, s2 = 0;
    Layout L[1 + MAX_EXTRA_MAPS];
    for (int m = 0; m < n_maps && share; ++m) {
        int a, b, c, n0, n1, n2;
        choose_tile(&grids[m], G, false, a, b, c);
        if (m == 0) { s0 = a; s1 = b; s2 = c; }
        else if (a != s0 || b != s1 || c != s2) share = false;
        const long long n_points = (long long)frames[m].n_frames * frames[m].height * frames[m].width;
        if (share && !make_layout(&grids[m], n_points, G, s0, s1, s2, L[m], n0, n1, n2))
            return fail(MF_ERR_INVALID, "problem too large for 32-bit bucket offsets");
        if (share && (!workspaces[m] || workspace_bytes[m] < L[m].total))
            return fail(MF_ERR_WORKSPACE, "workspace %d of %zu bytes given, %zu needed", m, workspace_bytes[m], L[m].total);
        if (share && ((uintptr_t)workspaces[m] & 255) != 0) return fail(MF_ERR_INVALID, "workspace must be 256-byte aligned");
    }
    if (!share) {
        for (int m = 0; m < n_maps; ++m) {
            const int rc = mf_fuse_frames(&grids[m], &frames[m], interpolation_weights[m], mode, workspaces[m], workspace_bytes[m], stream);
            if (rc != MF_OK) return rc;
        }
        return MF_OK;
    }
    MapStreams *ms = nullptr;
    int rc = map_streams(ms);
    if (rc != MF_OK) return rc;
    // the first map's call buckets for all
    FuseParams P = {};
    fill_grid(P, &grids[0]);
    fill_frames(P, &frames[0]);
    P.G = G;
    P.iw = interpolation_weights[0];
    char *ws0 = (char *)workspaces[0];
    int *ticket0 = (int *)(ws0 + L[0].ticket);
    P.n_extra = n_maps - 1;
    for (int m = 1; m < n_maps; ++m) {
        FuseParams::ExtraMap &E = P.extra[m - 1];
        const mf_frames &f = frames[m];
        E.feat = f.feat; E.feat_kind = f.feat_kind; E.C = grids[m].channels;
        E.fh = f.feat_kind == MF_FEAT_ONES ? f.height : f.feat_height;
        E.fw = f.feat_kind == MF_FEAT_ONES ? f.width : f.feat_width;
        E.rep_y = f.height / E.fh; E.rep_x = f.width / E.fw;
        E.aux = f.feat_kind == MF_FEAT_ONES ? nullptr : (uint32_t *)((char *)workspaces[m] + L[m].aux);
        E.label_status = f.label_status;
        E.abort = ticket0 + ABORT_MAPS + (m - 1);
        const int sv = s0 + s1 + s2;
        const bool single_m = takes_single(L[m], f.feat_kind, E.C, sv);
        E.absmax = single_m && f.feat_kind == MF_FEAT_DENSE_F32 ? (unsigned *)(ticket0 + ABSMAX_MAPS + (m - 1)) : nullptr;
        E.zero = (uint4 *)((char *)workspaces[m] + L[m].ticket);
        E.zero16 = (unsigned)(((single_m ? L[m].active : L[m].slot_count) - L[m].ticket) / 16);
    }
    MultiCtx lead = {};
    lead.role = 0; lead.scattered = ms->scattered;
    // the further maps' list parameters come from their own configuration (workspace layout, tile kernel grids)
    lead.n_lists = n_maps - 1;
    for (int m = 1; m < n_maps; ++m) {
        FuseParams Q = {};
        fill_grid(Q, &grids[m]);
        fill_frames(Q, &frames[m]);
        Q.G = G;
        Q.iw = interpolation_weights[m];
        MultiCtx plan = {};
        plan.role = 2; plan.abort = ticket0 + ABORT_MAPS + (m - 1);
        plan.s0 = s0; plan.s1 = s1; plan.s2 = s2;
        plan.plan = &lead.lists[m - 1];
        rc = run_pipeline<0>(Q, &grids[m], workspaces[m], workspace_bytes[m], st, 3, &plan);
        if (rc != MF_OK) return rc;
    }
    // (a side stream starts behind the first map's scatter_kernel: what the caller's stream holds at the call - the
    // frames, earlier updates of the maps - is ordered before it)
    rc = run_pipeline<0>(P, &grids[0], workspaces[0], workspace_bytes[0], st, 3, &lead);
    if (rc != MF_OK) return rc;             // nothing was issued on a side stream
    int first_error = MF_OK, joined = 0;
    for (int m = 1; m < n_maps; ++m) {
        hipStream_t side = ms->side[m - 1];
        FuseParams Q = {};
        fill_grid(Q, &grids[m]);
        fill_frames(Q, &frames[m]);
        Q.G = G;
        Q.iw = interpolation_weights[m];
        MultiCtx follow = {};
        follow.role = 1; follow.scattered = ms->scattered;
        follow.cursor = (int *)(ws0 + L[0].cursor);
        follow.rec = (uint4 *)(ws0 + L[0].rec);
        follow.nonempty = ticket0 + SPLIT_NONEMPTY;
        follow.abort = ticket0 + ABORT_MAPS + (m - 1);
        follow.absmax = (const unsigned *)(ticket0 + ABSMAX_MAPS + (m - 1));
        follow.s0 = s0; follow.s1 = s1; follow.s2 = s2;
        rc = run_pipeline<0>(Q, &grids[m], workspaces[m], workspace_bytes[m], side, 3, &follow);
        if (rc != MF_OK && first_error == MF_OK) first_error = rc;
        // joined whatever happened: the caller's stream continues behind everything a side stream was given
        if (hipEventRecord(ms->done[m - 1], side) == hipSuccess && hipStreamWaitEvent(st, ms->done[m - 1], 0) == hipSuccess) ++joined;
        else if (first_error == MF_OK) first_error = fail(MF_ERR_HIP, "joining a side stream failed");
        if (rc != MF_OK) break;
    }
    (void)joined;
    return first_error;
}

static int fuse_frames_phase(const mf_grid *grid, const mf_frames *frames, float interpolation_weight, int32_t mode,
                             void *workspace, size_t workspace_bytes, void *stream, int phase)
{
    int rc = check_grid(grid, true);
    if (rc != MF_OK) return rc;
    rc = check_frames(frames, grid->channels);
    if (rc != MF_OK) return rc;
    if (mode != MF_MODE_SEQUENTIAL && mode != MF_MODE_MERGED) return fail(MF_ERR_INVALID, "unknown mode %d", mode);
    FuseParams P = {};
    fill_grid(P, grid);
    fill_frames(P, frames);
    P.G = mode == MF_MODE_SEQUENTIAL ? frames->n_frames : 1;
    P.iw = interpolation_weight;
    return run_pipeline<0>(P, grid, workspace, workspace_bytes, (hipStream_t)stream, phase);
}

int mf_fuse_frames_stage(const mf_grid *grid, const mf_frames *frames, float interpolation_weight, int32_t mode,
                         void *workspace, size_t workspace_bytes, void *stream)
{
    return fuse_frames_phase(grid, frames, interpolation_weight, mode, workspace, workspace_bytes, stream, 1);
}

int mf_fuse_frames_commit(const mf_grid *grid, const mf_frames *frames, float interpolation_weight, int32_t mode,
                          void *workspace, size_t workspace_bytes, void *stream)
{
    return fuse_frames_phase(grid, frames, interpolation_weight, mode, workspace, workspace_bytes, stream, 2);
}

int mf_update_feature_map(const mf_grid *grid, int64_t n, const int64_t *ind0, const int64_t *ind1,
                          const int64_t *ind2, const float *ratio0, const float *ratio1, const float *ratio2,
                          const void *feat, int32_t feat_kind, float interpolation_weight, void *workspace,
                          size_t workspace_bytes, void *stream)
{
    int rc = check_grid(grid, false);
    if (rc != MF_OK) return rc;
    if (n < 0) return fail(MF_ERR_INVALID, "n must be >= 0");
    if (n == 0) return MF_OK;
    if (!ind0 || !ind1 || !ind2 || !ratio0 || !ratio1 || !ratio2) return fail(MF_ERR_INVALID, "index/ratio pointer is NULL");
    if (feat_kind < MF_FEAT_ONES || feat_kind > MF_FEAT_DENSE_F32) return fail(MF_ERR_INVALID, "unknown feat_kind %d", feat_kind);
    if (feat_kind == MF_FEAT_ONES && grid->channels != 1) return fail(MF_ERR_INVALID, "MF_FEAT_ONES needs channels == 1");
    if (feat_kind != MF_FEAT_ONES && !feat) return fail(MF_ERR_INVALID, "feat pointer is NULL");
    FuseParams P = {};
    fill_grid(P, grid);
    P.i0 = ind0; P.i1 = ind1; P.i2 = ind2; P.q0 = ratio0; P.q1 = ratio1; P.q2 = ratio2;
    P.feat = feat; P.feat_kind = feat_kind;
    P.n_points = n; P.G = 1; P.iw = interpolation_weight;
    return run_pipeline<1>(P, grid, workspace, workspace_bytes, (hipStream_t)stream);
}

int mf_transform_rays(const float *cam_rays, int64_t n_pixels, const float *poses, int32_t n_frames, float *out,
                      void *stream)
{
    if (!cam_rays || !poses || !out || n_pixels < 0 || n_frames < 0) return fail(MF_ERR_INVALID, "bad argument");
    const long long n = n_pixels * n_frames;
    if (n == 0) return MF_OK;
    hipLaunchKernelGGL(transform_rays_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       cam_rays, (long long)n_pixels, poses, n_frames, out);
    MF_LAUNCH_CHECK("transform_rays_kernel");
    return MF_OK;
}

int mf_bin_rays(const float *bins0, int32_t n0, const float *bins1, int32_t n1, const float *bins2, int32_t n2,
                const float *origin, const float *rays, int32_t rays_per_frame, const float *depth,
                int32_t n_frames, int64_t n_pixels, float min_depth, float max_depth, int64_t *ind0, int64_t *ind1,
                int64_t *ind2, float *ratio0, float *ratio1, float *ratio2, uint8_t *valid, void *stream)
{
    if (!bins0 || !bins1 || !bins2 || n0 < 2 || n1 < 2 || n2 < 2) return fail(MF_ERR_INVALID, "each axis needs >= 2 bin edges");
    if (!origin || !rays || !depth || n_frames < 0 || n_pixels < 0) return fail(MF_ERR_INVALID, "bad argument");
    const long long n = (long long)n_pixels * n_frames;
    if (n == 0) return MF_OK;
    Bins B = {bins0, bins1, bins2, n0, n1, n2};
    BinOut o = {ind0, ind1, ind2, ratio0, ratio1, ratio2, valid};
    hipLaunchKernelGGL(bin_rays_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, B,
                       origin, rays, rays_per_frame, depth, n_frames, (long long)n_pixels, min_depth, max_depth, o);
    MF_LAUNCH_CHECK("bin_rays_kernel");
    return MF_OK;
}

int mf_unproject_bin(const mf_grid *grid, const mf_frames *frames, int64_t *ind_x, int64_t *ind_y, int64_t *ind_z,
                     float *ratio_x, float *ratio_y, float *ratio_z, uint8_t *valid, void *stream)
{
    int rc = check_grid(grid, true);
    if (rc != MF_OK) return rc;
    if (!frames || frames->struct_size != sizeof(mf_frames) || !frames->cam_rays || !frames->poses || !frames->depth ||
        frames->n_frames < 1 || (frames->poses_on_host && frames->n_frames != 1))
        return fail(MF_ERR_INVALID, "frames incomplete (or mf_frames.struct_size != %zu)", sizeof(mf_frames));
    FuseParams P = {};
    fill_grid(P, grid);
    mf_frames f = *frames;
    f.feat_kind = MF_FEAT_ONES;
    fill_frames(P, &f);
    P.G = 1;
    BinOut o = {ind_x, ind_y, ind_z, ratio_x, ratio_y, ratio_z, valid};
    hipLaunchKernelGGL(unproject_bin_kernel, dim3((unsigned)(((P.H + PATCH - 1) / PATCH) * ((P.W + PATCH - 1) / PATCH)), (unsigned)P.n_frames),
                       dim3(256), 0, (hipStream_t)stream, P, o);
    MF_LAUNCH_CHECK("unproject_bin_kernel");
    return MF_OK;
}

}  // extern "C"
