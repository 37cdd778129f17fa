// reduce.hip — whole-map reductions the callers of the map layers run every few steps
// (SURVEY 8 f2), as single-pass HBM-streaming kernels over the [y][x][z][C] map.
//
//   mf_column_occupied  /root/reference/mass/navigation_policy.py:208-218
//       navigable = logical_not((norm(data, p=1, dim=3) > thr)[:, :, slice].any(dim=2))
//       (this kernel produces the `any` mask; the 2-D max_pool2d padding of :220-221 stays
//        in torch on the tiny [H, W] image)
//   mf_amax_z           /root/reference/agent.py:330-331, 391-392:  data.amax(dim=2)
//   mf_map_stats        the counters an episode reports about its maps (occupied voxels, sum |map|): what
//       `(data != 0).any(-1).sum()` and `data.abs().sum()` give, in ONE pass over the map (bench.py --workload episode:
//       the torch expressions move the 3.6 GB map five times and build a 3.6 GB temporary)
//
// A (y, x) column of the map is one contiguous run of D*C floats, so both kernels give a
// workgroup one column and read it with consecutive lanes on consecutive floats.
#include "common.h"

namespace mf {

constexpr int RT = 256;

// out[col] = 1 if any voxel z in [z0, z1) of the column has sum_c |m| > thr
__global__ __launch_bounds__(RT) void column_occupied_kernel(const float *__restrict__ map, int D, int C, int z0,
                                                              int z1, float thr, int zchunk, uint8_t *out)
{
    extern __shared__ float col[];          // [zchunk][C]
    __shared__ int found;
    const size_t base = (size_t)blockIdx.x * D * C;
    if (threadIdx.x == 0) found = 0;
    __syncthreads();
    for (int za = z0; za < z1; za += zchunk) {
        const int nz = min(zchunk, z1 - za);
        const float *src = map + base + (size_t)za * C;
        for (int i = threadIdx.x; i < nz * C; i += RT) col[i] = src[i];
        __syncthreads();
        for (int z = threadIdx.x; z < nz; z += RT) {
            float s = 0.0f;
            for (int c = 0; c < C; ++c) s += fabsf(col[z * C + c]);
            if (s > thr) found = 1;          // benign race: every writer stores 1
        }
        __syncthreads();
        if (found) break;
    }
    if (threadIdx.x == 0) out[blockIdx.x] = (uint8_t)found;
}

// out[col][c] = max_z map[col][z][c]
// vec4: the column is read as float4s.  Thread t of the first P4 * (RT / P4) threads strides the column by S4 = P4 * (RT / P4)
// float4s, P4 = C / gcd(C, 4) float4s being the period after which a float4 starts on the same channel again: its four lanes
// see four fixed channels (4 t + k) mod C all the way down, and the threads that share them meet in LDS at the end
// (16-byte loads: 0.78 -> ~0.6 ms on the 3.6 GB semantic map).
__global__ __launch_bounds__(RT) void amax_z_kernel(const float *__restrict__ map, int D, int C, int P4, float *out)
{
    __shared__ float part[4 * RT];
    const float *src = map + (size_t)blockIdx.x * D * C;
    float *dst = out + (size_t)blockIdx.x * C;
    if (P4 > 0) {
        const int rep = RT / P4, S4 = P4 * rep, n4 = (D * C) >> 2;
        float4 m = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
        if ((int)threadIdx.x < S4) {
            const float4 *s4 = reinterpret_cast<const float4 *>(src);
            for (int i = threadIdx.x; i < n4; i += S4) {
                const float4 v = s4[i];
                m.x = fmaxf(m.x, v.x); m.y = fmaxf(m.y, v.y); m.z = fmaxf(m.z, v.z); m.w = fmaxf(m.w, v.w);
            }
        }
        reinterpret_cast<float4 *>(part)[threadIdx.x] = m;
        __syncthreads();
        // the threads t0, t0 + P4, ... share their four channels: folded by thread t0 first ...
        if ((int)threadIdx.x < P4) {
            float4 r = reinterpret_cast<float4 *>(part)[threadIdx.x];
            for (int j = 1; j < rep; ++j) {
                const float4 v = reinterpret_cast<float4 *>(part)[threadIdx.x + j * P4];
                r.x = fmaxf(r.x, v.x); r.y = fmaxf(r.y, v.y); r.z = fmaxf(r.z, v.z); r.w = fmaxf(r.w, v.w);
            }
            reinterpret_cast<float4 *>(part)[threadIdx.x] = r;
        }
        __syncthreads();
        // ... then channel c collects the lanes (c - first channel of t0's float4) mod C < 4 over the P4 threads
        if ((int)threadIdx.x < C) {
            const int c = threadIdx.x;
            float r = -INFINITY;
            for (int t0 = 0; t0 < P4; ++t0) {
                const int c0 = (4 * t0) % C;
                const int k = c - c0 < 0 ? c - c0 + C : c - c0;
                if (k < 4) r = fmaxf(r, part[4 * t0 + k]);
            }
            dst[c] = r;
        }
    } else if (C <= RT) {
        // threads t < S = C * (RT / C) stride the column by S, so a thread always sees channel t % C
        const int rep = RT / C, S = C * rep, n = D * C;
        float m = -INFINITY;
        if ((int)threadIdx.x < S)
            for (int e = threadIdx.x; e < n; e += S) m = fmaxf(m, src[e]);
        part[threadIdx.x] = m;
        __syncthreads();
        if ((int)threadIdx.x < C) {
            float r = part[threadIdx.x];
            for (int k = 1; k < rep; ++k) r = fmaxf(r, part[threadIdx.x + k * C]);
            dst[threadIdx.x] = r;
        }
    } else {
        for (int c = threadIdx.x; c < C; c += RT) {
            float m = -INFINITY;
            for (int z = 0; z < D; ++z) m = fmaxf(m, src[(size_t)z * C + c]);
            dst[c] = m;
        }
    }
}

// Per workgroup: voxels of its columns with a non-zero channel, and sum |x| as a 64-bit integer in units of 2^-24 (an exact,
// order-independent sum: the same bits whatever the number of workgroups or ranks).  part[2 b] / part[2 b + 1].
__global__ __launch_bounds__(RT) void map_stats_kernel(const float *__restrict__ map, int n_cols, int D, int C, unsigned magicC,
                                                        int vec4, unsigned long long *part)
{
    extern __shared__ int flag[];            // [D]
    __shared__ unsigned long long wsum[RT / 64];
    __shared__ int wcnt[RT / 64];
    const int n = D * C;
    unsigned long long acc = 0ull;
    int cnt = 0;
    auto take = [&](float x, unsigned e) {
        x = fabsf(x);
        if (x != 0.0f) {
            flag[magicC ? __umulhi(e, magicC) : e] = 1;                       // voxel e / C (benign race: every writer stores 1)
            acc += (unsigned long long)(fminf(x, 5.0e11f) * 16777216.0f);
        }
    };
    for (int col = blockIdx.x; col < n_cols; col += gridDim.x) {
        for (int z = threadIdx.x; z < D; z += RT) flag[z] = 0;
        __syncthreads();
        const float *src = map + (size_t)col * n;
        if (vec4) {
            const float4 *s4 = reinterpret_cast<const float4 *>(src);
            for (int i = threadIdx.x; i < n / 4; i += RT) {
                const float4 v = s4[i];
                const unsigned e = 4u * i;
                take(v.x, e); take(v.y, e + 1); take(v.z, e + 2); take(v.w, e + 3);
            }
        } else {
            for (int e = threadIdx.x; e < n; e += RT) take(src[e], (unsigned)e);
        }
        __syncthreads();
        for (int z = threadIdx.x; z < D; z += RT) cnt += flag[z];
        __syncthreads();
    }
    for (int o = 32; o > 0; o >>= 1) {
        cnt += __shfl_down(cnt, o, 64);
        acc += (unsigned long long)__shfl_down((long long)acc, o, 64);
    }
    if ((threadIdx.x & 63) == 0) { wsum[threadIdx.x >> 6] = acc; wcnt[threadIdx.x >> 6] = cnt; }
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long a = 0ull, c = 0ull;
        for (int w = 0; w < RT / 64; ++w) { a += wsum[w]; c += (unsigned long long)wcnt[w]; }
        part[2 * blockIdx.x] = c;
        part[2 * blockIdx.x + 1] = a;
    }
}

__global__ __launch_bounds__(RT) void map_stats_sum_kernel(const unsigned long long *__restrict__ part, int n_parts, unsigned long long *out)
{
    __shared__ unsigned long long sh[2][RT / 64];
    unsigned long long c = 0ull, a = 0ull;
    for (int i = threadIdx.x; i < n_parts; i += RT) { c += part[2 * i]; a += part[2 * i + 1]; }
    for (int o = 32; o > 0; o >>= 1) {
        c += (unsigned long long)__shfl_down((long long)c, o, 64);
        a += (unsigned long long)__shfl_down((long long)a, o, 64);
    }
    if ((threadIdx.x & 63) == 0) { sh[0][threadIdx.x >> 6] = c; sh[1][threadIdx.x >> 6] = a; }
    __syncthreads();
    if (threadIdx.x == 0) {
        c = 0ull; a = 0ull;
        for (int w = 0; w < RT / 64; ++w) { c += sh[0][w]; a += sh[1][w]; }
        out[0] = c; out[1] = a;
    }
}

}  // namespace mf

using namespace mf;

extern "C" {

int mf_column_occupied(const float *map, int32_t size0, int32_t size1, int32_t size2, int32_t channels, int32_t z0,
                       int32_t z1, float threshold, uint8_t *out, void *stream)
{
    if (!map || !out) return fail(MF_ERR_INVALID, "NULL pointer");
    if (size0 < 1 || size1 < 1 || size2 < 1 || channels < 1) return fail(MF_ERR_INVALID, "bad map shape");
    if (z0 < 0) z0 = 0;
    if (z1 > size2) z1 = size2;
    if (z1 < z0) z1 = z0;
    int zchunk = 8192 / channels;
    if (zchunk < 1) zchunk = 1;
    if (zchunk > size2) zchunk = size2;
    const size_t lds = (size_t)zchunk * channels * 4;
    if (lds > 60 * 1024) return fail(MF_ERR_INVALID, "channels = %d too large for the column kernel", channels);
    hipLaunchKernelGGL(column_occupied_kernel, dim3((unsigned)(size0 * size1)), dim3(RT), lds, (hipStream_t)stream,
                       map, size2, channels, z0, z1, threshold, zchunk, out);
    MF_LAUNCH_CHECK("column_occupied_kernel");
    return MF_OK;
}

int mf_amax_z(const float *map, int32_t size0, int32_t size1, int32_t size2, int32_t channels, float *out, void *stream)
{
    if (!map || !out) return fail(MF_ERR_INVALID, "NULL pointer");
    if (size0 < 1 || size1 < 1 || size2 < 1 || channels < 1) return fail(MF_ERR_INVALID, "bad map shape");
    // float4 path: columns that start on 16 bytes, a period of at most RT float4s, and channels that fit one pass of the fold
    int g = channels % 4 == 0 ? 4 : channels % 2 == 0 ? 2 : 1;
    int P4 = channels / g;
    if (((uintptr_t)map % 16 != 0) || (((long long)size2 * channels) % 4 != 0) || P4 > RT || channels > RT || channels < 4) P4 = 0;
    hipLaunchKernelGGL(amax_z_kernel, dim3((unsigned)(size0 * size1)), dim3(RT), 0, (hipStream_t)stream, map, size2,
                       channels, P4, out);
    MF_LAUNCH_CHECK("amax_z_kernel");
    return MF_OK;
}

int mf_map_stats(const float *map, int32_t size0, int32_t size1, int32_t size2, int32_t channels, uint64_t *out,
                 uint64_t *scratch, void *stream)
{
    if (!map || !out || !scratch) return fail(MF_ERR_INVALID, "NULL pointer");
    if (size0 < 1 || size1 < 1 || size2 < 1 || channels < 1) return fail(MF_ERR_INVALID, "bad map shape");
    if ((long long)size2 * channels > 0x7fffffffLL / 4) return fail(MF_ERR_INVALID, "column of %d x %d floats too long", size2, channels);
    if ((size_t)size2 * 4 > 60 * 1024) return fail(MF_ERR_INVALID, "map depth %d too large for the statistics kernel", size2);
    const int n_cols = size0 * size1;
    const int blocks = n_cols < MF_MAP_STATS_PARTS ? n_cols : MF_MAP_STATS_PARTS;
    const unsigned magicC = channels == 1 ? 0u : (unsigned)(((1ull << 32) + channels - 1) / channels);
    const int vec4 = ((uintptr_t)map % 16 == 0) && (((long long)size2 * channels) % 4 == 0);
    hipLaunchKernelGGL(map_stats_kernel, dim3((unsigned)blocks), dim3(RT), (size_t)size2 * 4, (hipStream_t)stream, map, n_cols,
                       size2, channels, magicC, vec4, (unsigned long long *)scratch);
    MF_LAUNCH_CHECK("map_stats_kernel");
    hipLaunchKernelGGL(map_stats_sum_kernel, dim3(1), dim3(RT), 0, (hipStream_t)stream, (const unsigned long long *)scratch, blocks,
                       (unsigned long long *)out);
    MF_LAUNCH_CHECK("map_stats_sum_kernel");
    return MF_OK;
}

}  // extern "C"

// ----------------------------------------------------------------------------
// per-box moments of one class channel (SemanticProjectionLayer.find)
// ----------------------------------------------------------------------------
namespace mf {

__device__ __forceinline__ float block_sum(float v, float *sh)
{
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    float t = 0.0f;
    for (int w = 0; w < RT / 64; ++w) t += sh[w];
    return t;
}

// one workgroup per box; voxels with m == 0 (almost all of them) cost one strided read
__global__ __launch_bounds__(RT) void roi_moments_kernel(const float *__restrict__ map, int size1, int size2, int C,
                                                          int category, const float *__restrict__ cx,
                                                          const float *__restrict__ cy, const float *__restrict__ cz,
                                                          const int *__restrict__ boxes, const float *__restrict__ feat,
                                                          int FC, float *out, float *feat_out)
{
    __shared__ float sh[RT / 64];
    __shared__ int nz_count;
    __shared__ int nz_vox[1024];
    __shared__ float nz_m[1024];
    const int bx = boxes[blockIdx.x * 4], by = boxes[blockIdx.x * 4 + 1], bw = boxes[blockIdx.x * 4 + 2],
              bh = boxes[blockIdx.x * 4 + 3];
    const int n = bw * bh * size2;
    float s1 = 0, s2 = 0, sx = 0, sy = 0, sz = 0;
    for (int c = threadIdx.x; c < FC; c += RT) feat_out[(size_t)blockIdx.x * FC + c] = 0.0f;
    for (int base = 0; base < n; base += 1024) {
        if (threadIdx.x == 0) nz_count = 0;
        __syncthreads();
        for (int e = base + threadIdx.x; e < min(n, base + 1024); e += RT) {
            const int z = e % size2, r = e / size2, x = bx + r % bw, y = by + r / bw;
            const int vox = (y * size1 + x) * size2 + z;
            const float m = map[(size_t)vox * C + category];
            if (m != 0.0f) {
                s1 += m; s2 += m * m; sx += m * cx[x]; sy += m * cy[y]; sz += m * cz[z];
                if (feat) { const int k = atomicAdd(&nz_count, 1); nz_vox[k] = vox; nz_m[k] = m; }
            }
        }
        __syncthreads();
        if (feat)
            for (int c = threadIdx.x; c < FC; c += RT) {
                float acc = 0.0f;
                for (int k = 0; k < nz_count; ++k) acc += nz_m[k] * feat[(size_t)nz_vox[k] * FC + c];
                feat_out[(size_t)blockIdx.x * FC + c] += acc;
            }
        __syncthreads();
    }
    s1 = block_sum(s1, sh); s2 = block_sum(s2, sh); sx = block_sum(sx, sh); sy = block_sum(sy, sh);
    sz = block_sum(sz, sh);
    if (threadIdx.x == 0) {
        float *o = out + (size_t)blockIdx.x * 5;
        o[0] = s1; o[1] = s2; o[2] = sx; o[3] = sy; o[4] = sz;
    }
}

}  // namespace mf

extern "C" int mf_roi_moments(const float *map, int32_t size0, int32_t size1, int32_t size2, int32_t channels,
                              int32_t category, const float *cx, const float *cy, const float *cz,
                              const int32_t *boxes, int32_t n_boxes, const float *feat, int32_t feat_channels,
                              float *out, float *feat_out, void *stream)
{
    if (n_boxes == 0) return MF_OK;
    if (!map || !cx || !cy || !cz || !boxes || !out || n_boxes < 0) return mf::fail(MF_ERR_INVALID, "bad argument");
    if (category < 0 || category >= channels) return mf::fail(MF_ERR_INVALID, "category %d outside [0, %d)", category, channels);
    if (feat && (!feat_out || feat_channels < 1)) return mf::fail(MF_ERR_INVALID, "feat_out / feat_channels missing");
    (void)size0;
    hipLaunchKernelGGL(mf::roi_moments_kernel, dim3((unsigned)n_boxes), dim3(mf::RT), 0, (hipStream_t)stream, map, size1,
                       size2, channels, category, cx, cy, cz, boxes, feat, feat ? feat_channels : 0, out, feat_out);
    MF_LAUNCH_CHECK("roi_moments_kernel");
    return MF_OK;
}
