/*
 * massfuse.h — C ABI of libmassfuse.so: the MI355X (gfx950) implementation of
 * the MaSS per-frame voxel-map update and inter-map matching hot path.
 *
 * The reference (brandontrabucco/mass, /root/reference) is pure Python and has
 * no FFI of its own; every entry point below replaces a *sequence of torch ops*
 * inside one reference function, cited as file:line relative to the reference
 * root.  The host side that keeps the reference's Python call surface lives in
 * mass_amd/ and binds these symbols with ctypes.
 *
 * Conventions
 *   - plain C types only; every pointer marked "device" is a HIP device
 *     pointer owned by the caller (torch tensors in the Python host); the
 *     library allocates no persistent device memory.
 *   - `stream` is a hipStream_t passed as void* (0 = default stream); all work
 *     is enqueued on it and nothing synchronises.  Calls that touch the same
 *     map / workspace must be stream-ordered by the caller.  The device of
 *     `stream` is the calling thread's current device (hipSetDevice), as for a
 *     kernel launch: one process per GPU is the model.
 *     mf_fuse_frame_maps is the one call that uses streams of its own beside
 *     `stream` (kept per host thread and device, forked from and joined into
 *     `stream` inside the call).
 *   - return value: MF_OK (0) or a negative MF_ERR_* code; the message is
 *     available from mf_last_error() (thread local).  No C++ exception crosses
 *     the ABI.
 *   - axis convention (reference base_projection_layer.py:334-341): world axis
 *     0 = x <-> bins_x <-> map dim 1 (map_width); world axis 1 = y <-> bins_y
 *     <-> map dim 0 (map_height, index flipped); world axis 2 = z <-> bins_z
 *     <-> map dim 2 (map_depth).  The map is fp32 [size0][size1][size2][C],
 *     C fastest, updated in place.
 */
#ifndef MASSFUSE_H
#define MASSFUSE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MF_ABI_VERSION 5

#if defined(__GNUC__)
#define MF_API __attribute__((visibility("default")))
#else
#define MF_API
#endif

#define MF_OK             0
#define MF_ERR_INVALID   -1   /* bad argument (shape, null pointer, unsupported size) */
#define MF_ERR_WORKSPACE -2   /* workspace too small: see mf_fuse_workspace_bytes     */
#define MF_ERR_HIP       -3   /* a HIP runtime call or kernel launch failed           */

/* feature kinds (what the per-pixel "features" image holds) */
#define MF_FEAT_ONES       0  /* features = ones_like(depth), C must be 1 (occupancy_projection_layer.py:159-161) */
#define MF_FEAT_LABEL_U8   1  /* class id per pixel, uint8;  stands for one_hot(label, C).float()                 */
#define MF_FEAT_LABEL_I32  2  /* class id per pixel, int32   (semantic_projection_layer.py:203-214)                */
#define MF_FEAT_LABEL_I64  3  /* class id per pixel, int64                                                         */
#define MF_FEAT_DENSE_F32  4  /* fp32 [.., C] per pixel      (base_projection_layer.py:320-325)                    */

/* batch semantics */
#define MF_MODE_SEQUENTIAL 0  /* frame t+1 blends against the result of frame t: n calls of layer.update()         */
#define MF_MODE_MERGED     1  /* all frames form one point set: the functional API with a leading batch (A.6)      */

/* Voxel map + bin edges.  Replaces the buffers of BaseProjectionLayer
 * (base_projection_layer.py:153-181). */
typedef struct mf_grid {
    uint32_t struct_size;               /* = sizeof(mf_grid), set by the caller: every entry point that takes the
                                           struct refuses a different value (a binding written against another ABI
                                           version fails with MF_ERR_INVALID instead of reading past its struct) */
    int32_t size0, size1, size2;        /* map_height, map_width, map_depth (each <= 1024)            */
    int32_t channels;                   /* feature_size C                                              */
    const float *bins_x;                /* device, n_edges_x floats (world axis 0)                     */
    const float *bins_y;                /* device, n_edges_y floats (world axis 1)                     */
    const float *bins_z;                /* device, n_edges_z floats (world axis 2)                     */
    int32_t n_edges_x, n_edges_y, n_edges_z;   /* must be size1+1, size0+1, size2+1                    */
    float *map;                         /* device fp32 [size0*size1*size2*channels], in place          */
} mf_grid;

/* A batch of posed RGB-D(+feature) frames.  Replaces the observation dict of
 * BaseProjectionLayer.update (base_projection_layer.py:309-325). */
typedef struct mf_frames {
    uint32_t struct_size;               /* = sizeof(mf_frames), checked like mf_grid.struct_size            */
    int32_t n_frames;
    int32_t height, width;              /* camera resolution of `depth` and `cam_rays`                 */
    const float *cam_rays;              /* device [height*width*3]: the layer's `rays` buffer
                                           (project_camera_rays, projection.py:34-74)                  */
    const float *poses;                 /* device [n_frames*12]: origin[3] then R[3][3] row-major,
                                           R = stack([eye x up, up, -eye], -1) (projection.py:104-105),
                                           computed by the host                                        */
    const float *depth;                 /* device [n_frames*height*width] metres                       */
    const void *feat;                   /* device, per feat_kind; NULL for MF_FEAT_ONES                */
    int32_t feat_kind;
    int32_t feat_height, feat_width;    /* resolution of `feat`; height%feat_height==0 etc.
                                           (repeat_interleave upsampling, base_projection_layer.py:322-325) */
    float min_depth, max_depth;         /* bin_rays min_ray_depth / max_ray_depth (0, 10)              */
    int32_t *label_status;              /* optional (may be NULL), MF_FEAT_LABEL_* only: a device-visible
                                           int32 (pinned host memory works) that is set to 1 when ANY pixel of
                                           the batch (valid depth or not: the reference's one_hot looks at the
                                           whole image) carries a class id outside [0, channels); the call then
                                           leaves the map untouched, like the reference, whose one_hot raises
                                           before anything is written (semantic_projection_layer.py:203-209).
                                           With NULL such ids count as an all-zero feature row.              */
    int32_t poses_on_host;              /* 0: `poses` is device memory.  1: `poses` points at HOST memory that is
                                           read during the call and travels as a kernel argument (n_frames must be
                                           1): the per-step caller (one frame per update, agent.py:107-111) then
                                           needs no upload of 48 bytes, whose copy from pageable memory makes the
                                           host wait for the stream's earlier work                           */
} mf_frames;

MF_API int mf_version(void);
MF_API const char *mf_last_error(void);

/* sizeof(mf_grid) / sizeof(mf_frames) as this library was compiled: a binding compares them with its own
 * mirror of the structs when it loads the library (mass_amd/_lib.py does, and refuses to import on a
 * mismatch).  Either pointer may be NULL. */
MF_API int mf_struct_sizes(size_t *grid_bytes, size_t *frames_bytes);

/* ---- parity entry points (one per reference function) --------------------- */

/* transform_rays, projection.py:77-110.  out[f][p][i] = ((r0*R[i][0] + r1*R[i][1]) + r2*R[i][2]).
 * cam_rays device [n_pixels*3]; poses device [n_frames*12]; out device [n_frames*n_pixels*3]. */
MF_API int mf_transform_rays(const float *cam_rays, int64_t n_pixels, const float *poses, int32_t n_frames,
                      float *out, void *stream);

/* bin_rays, projection.py:113-230, per-pixel (uncompacted) form: for every ray
 * writes the voxel index triple (int64, axis 1 flipped), the three in-voxel
 * ratios (axis 1 as 1-r) and the validity criterion; invalid pixels get
 * ratio 0.  origin device [n_frames*3]; rays device [n_frames*n_pixels*3]
 * (world frame) or, if rays_per_frame == 0, [n_pixels*3] shared by all frames;
 * depth device [n_frames*n_pixels].  Any output pointer may be NULL. */
MF_API int mf_bin_rays(const float *bins0, int32_t n0, const float *bins1, int32_t n1,
                const float *bins2, int32_t n2,
                const float *origin, const float *rays, int32_t rays_per_frame,
                const float *depth, int32_t n_frames, int64_t n_pixels,
                float min_depth, float max_depth,
                int64_t *ind0, int64_t *ind1, int64_t *ind2,
                float *ratio0, float *ratio1, float *ratio2, uint8_t *valid, void *stream);

/* Fused a3+a4 exactly as the hot path evaluates them (camera rays + poses ->
 * indices), same outputs as mf_bin_rays; exists so tests can prove the fused
 * geometry equals transform_rays followed by bin_rays bit for bit. */
MF_API int mf_unproject_bin(const mf_grid *grid, const mf_frames *frames,
                     int64_t *ind_x, int64_t *ind_y, int64_t *ind_z,
                     float *ratio_x, float *ratio_y, float *ratio_z, uint8_t *valid, void *stream);

/* ---- the hot path ---------------------------------------------------------- */

/* Bytes of device workspace mf_fuse_frames / mf_update_feature_map need for at
 * most n_points input points (pixels x frames) in n_groups sequential groups
 * (n_frames for MF_MODE_SEQUENTIAL, 1 for MF_MODE_MERGED).  0 on bad input. */
MF_API size_t mf_fuse_workspace_bytes(const mf_grid *grid, int64_t n_points, int32_t n_groups);

/* BaseProjectionLayer.update for a batch of frames
 * (base_projection_layer.py:282-343 = transform_rays + bin_rays +
 * update_feature_map, and the one_hot / ones_like front ends of
 * semantic_projection_layer.py:203-214 / occupancy_projection_layer.py:159-161).
 * Updates grid->map in place.  n_frames <= 256 per call.
 * The workspace needs no initialisation and holds nothing between calls.
 * Sequential frames of class ids / ones are bucketed on 4x4x8 map tiles in a tile-local entry format
 * picked on the device from a sample of the call's points - unrelated / sparse frames: one
 * contribution per corner of a point's footprint; real scenes: the corners of a pixel patch summed
 * per (voxel, class) before they are written (aggregated entries) - and fused by the all-integer
 * fuse_cells_kernel (fuse_dense_kernel over point records for maps of more than 2^17 tiles);
 * everything else (dense fp32 features, blend weights outside [0, 1], odd map shapes) takes
 * fuse_tiles_kernel.  The choice is a function of the call's arguments and data only: the library
 * keeps no state between calls, and the integer kernels give run-to-run identical bits.
 * Such a call (several sequential frames of class ids / ones) reads its probe's verdict back on
 * the host - it WAITS for `stream`'s earlier work and ~30 us - and launches only the kernels of the
 * chosen format; MF_PROBE_SYNC=0 (environment, read once) launches every variant instead and lets
 * the device pick (no wait; aggregated entries are then not used). */
MF_API int mf_fuse_frames(const mf_grid *grid, const mf_frames *frames, float interpolation_weight,
                   int32_t mode, void *workspace, size_t workspace_bytes, void *stream);

/* One observation onto several maps: what the reference's agent does per simulator step,
 * `for name in update_map: self.feature_maps[name].update(observations)`
 * (navigation_policy.py:164-171, maps built at agent.py:107-111).  grids[m] / frames[m] /
 * interpolation_weights[m] / workspaces[m] describe map m exactly as a mf_fuse_frames call would
 * (its own map buffer, channels, features, label_status, blend weight, workspace); the maps share
 * their voxel grid (sizes; the edges of grids[0] are the ones used - the caller passes maps whose
 * edges are equal) and the frames their rays, poses and depth.  n_maps <= 4.
 * The result on every map is that of its own mf_fuse_frames call (the same bits wherever that call
 * takes the all-integer single-pass kernels, i.e. frames of a real scene).  A single group
 * (one frame, or MF_MODE_MERGED) is bucketed ONCE - the points, their tiles and records do not
 * depend on the features - and the maps' tile kernels run side by side on streams the library
 * keeps per host thread, forked after what `stream` holds at the call and joined into it before
 * the call returns; other calls are issued map after map.  A class id out of range on one map
 * (its label_status set) leaves that map untouched and the others updated, as in the loop. */
MF_API int mf_fuse_frame_maps(const mf_grid *grids, const mf_frames *frames, const float *interpolation_weights,
                       int32_t n_maps, int32_t mode, void *const *workspaces, const size_t *workspace_bytes,
                       void *stream);

/* The same update in two halves, for callers that fuse batch after batch: `stage` buckets the
 * frames' points into the workspace (unproject, bin, count, scatter: the map is not read), `commit`
 * applies a staged workspace to the map.  Staging batch k+1 on a second stream while batch k is
 * being committed overlaps the two (the tile kernels leave half of the CUs' wave slots free);
 * commits must stay in order on one stream, each workspace is owned by its batch from stage to
 * the end of commit, and both calls take the same grid / frames / weight / mode arguments (the
 * blend weight decides with the feature kind how the points are bucketed). */
MF_API int mf_fuse_frames_stage(const mf_grid *grid, const mf_frames *frames, float interpolation_weight,
                         int32_t mode, void *workspace, size_t workspace_bytes, void *stream);
MF_API int mf_fuse_frames_commit(const mf_grid *grid, const mf_frames *frames, float interpolation_weight,
                          int32_t mode, void *workspace, size_t workspace_bytes, void *stream);

/* update_feature_map, projection.py:233-351, for already binned points:
 * ind0/1/2 int64 and ratio0/1/2 fp32 device [n]; feat per feat_kind with one
 * row per point ([n] labels or [n][C] fp32).  Only grid->size*, channels and
 * map are read from `grid`. */
MF_API int mf_update_feature_map(const mf_grid *grid, int64_t n,
                          const int64_t *ind0, const int64_t *ind1, const int64_t *ind2,
                          const float *ratio0, const float *ratio1, const float *ratio2,
                          const void *feat, int32_t feat_kind, float interpolation_weight,
                          void *workspace, size_t workspace_bytes, void *stream);

/* ---- whole-map reductions used by the callers of the layers ----- */

/* navigation_policy.py:208-218: out[y*size1 + x] = 1 if any voxel z in [z0, z1) of column
 * (y, x) has sum_c |map| > threshold, else 0 (uint8, device).  The caller turns it into the
 * navigable image and applies the 2-D padding (:220-221). */
MF_API int mf_column_occupied(const float *map, int32_t size0, int32_t size1, int32_t size2, int32_t channels,
                              int32_t z0, int32_t z1, float threshold, uint8_t *out, void *stream);

/* agent.py:330-331,391-392: out[y][x][c] = max over z of map[y][x][z][c] (data.amax(dim=2)). */
MF_API int mf_amax_z(const float *map, int32_t size0, int32_t size1, int32_t size2, int32_t channels, float *out,
                     void *stream);

/* The counters an episode reports about a map, in one pass: out[0] = voxels with a non-zero
 * channel (`(data != 0).any(-1).sum()`), out[1] = sum of |map| in units of 2^-24 (an exact integer
 * sum, each term truncated: the same bits whatever the order; divide by 2^24 for `data.abs().sum()`).
 * out: 2 words, scratch: 2 * MF_MAP_STATS_PARTS words, both device memory (no initialisation). */
#define MF_MAP_STATS_PARTS 2048
MF_API int mf_map_stats(const float *map, int32_t size0, int32_t size1, int32_t size2, int32_t channels,
                        uint64_t *out, uint64_t *scratch, void *stream);

/* ---- instance extraction, SemanticProjectionLayer.find ---------- */

/* cv2.findContours(RETR_LIST) + cv2.boundingRect (semantic_projection_layer.py:323-328) on
 * a HOST uint8 image [height][width]: bounding boxes (x, y, w, h) of every outer and hole
 * border, 8-connected foreground (Suzuki-Abe border following).  reverse_order != 0 lists
 * the border found last first, as OpenCV does.  Returns the number of borders (which may
 * exceed max_boxes; only max_boxes are written) or <0.  Parity unpinned (cv2 absent). */
MF_API int mf_contour_boxes(const uint8_t *img, int32_t height, int32_t width, int32_t reverse_order,
                            int32_t *boxes, int32_t max_boxes);

/* Per-box moments of one class channel (semantic_projection_layer.py:331-357):
 * for box b = (x, y, w, h) over all z, with m = map[y'][x'][z][category]:
 *   out[b] = { sum m, sum m^2, sum m*cx[x'], sum m*cy[y'], sum m*cz[z] }  (5 doubles-as-floats)
 * and, if feat != NULL, feat_out[b][j] = sum m * feat[y'][x'][z][j].
 * map device fp32 [size0][size1][size2][channels]; cx/cy/cz device fp32 voxel-centre world
 * coordinates per index; boxes device int32 [n_boxes][4]; feat device fp32
 * [size0][size1][size2][feat_channels] or NULL; out device fp32 [n_boxes][5];
 * feat_out device fp32 [n_boxes][feat_channels]. */
MF_API int mf_roi_moments(const float *map, int32_t size0, int32_t size1, int32_t size2, int32_t channels,
                          int32_t category, const float *cx, const float *cy, const float *cz,
                          const int32_t *boxes, int32_t n_boxes, const float *feat, int32_t feat_channels,
                          float *out, float *feat_out, void *stream);

/* ---- diagnostics ------------------------------------------------------------ */

/* Stage timing of the fuse pipeline for the roofline report (bench.py): after
 * mf_profile_enable(1), each of the next (up to 256) mf_fuse_frames /
 * mf_update_feature_map calls records HIP events on its stream between its
 * stages, without synchronising.  mf_profile_read(call, ms) waits for profiled
 * call number `call` (0-based since the enable) and writes milliseconds
 *   ms[0] zero + count   ms[1] scan   ms[2] scatter   ms[3] tile kernels   ms[4] their sum
 * (a stage / commit pair counts as one call, recorded when the commit is issued);
 * it returns the number of calls recorded so far, or <0.
 * Process-wide, not thread safe; off by default. */
MF_API int mf_profile_enable(int32_t on);
MF_API int mf_profile_read(int32_t call, float *ms /* [5] host */);

/* Which tile kernel took the most recent multi-frame call of class-id / ones frames issued with
 * this workspace (same grid, n_points and n_groups as that call): 0 fuse_tiles_kernel,
 * 2 fuse_dense_kernel, 3 fuse_cells_kernel (contributions), 4 fuse_cells_kernel over aggregated
 * entries (real scenes); < 0 on error.  Reads one word of the workspace back and
 * WAITS for `stream`.  Used by the tests to prove which path ran. */
MF_API int mf_fuse_last_mode(const mf_grid *grid, int64_t n_points, int32_t n_groups, const void *workspace,
                             void *stream);

/* ---- matching (experimentation.py:261-265, 277-280, 284-287) --------------- */

#define MF_METRIC_L2       0  /* || f0_i - f1_j ||_2, difference form (reference arithmetic)          */
#define MF_METRIC_L2_GEMM  1  /* sqrt(max(|a|^2 + |b|^2 - 2 a.b, 0)) on fp32 MFMA                      */
#define MF_METRIC_COSINE   2  /* 1 - a.b / (|a||b|) on fp32 MFMA (not in the reference; config 4 extra) */

/* f0 device [n0][d], f1 device [n1][d], out device [n0][n1], all fp32. */
MF_API int mf_pairwise_distance(const float *f0, int32_t n0, const float *f1, int32_t n1, int32_t d,
                         float *out, int32_t metric, void *stream);

/* scipy.optimize.linear_sum_assignment (minimise; rectangular allowed) on a
 * HOST cost matrix [n0][n1] (float64, row-major).  Writes min(n0,n1) pairs to
 * row_ind/col_ind (host), rows ascending.  Returns the pair count or <0. */
MF_API int mf_linear_sum_assignment(const double *cost, int32_t n0, int32_t n1,
                             int64_t *row_ind, int64_t *col_ind);

#ifdef __cplusplus
}
#endif
#endif /* MASSFUSE_H */
