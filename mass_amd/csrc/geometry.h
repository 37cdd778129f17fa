// geometry.h — device-side restatement of the reference's per-pixel geometry
// (/root/reference/mass/utils/projection.py:77-110, 113-230, 280-323).
//
// Everything here must reproduce the reference's fp32 results bit for bit:
// one IEEE rounding per torch op, so this header is only included from
// translation units built with -ffp-contract=off (no FMA formation) and
// without fast-math; fp32 division is HIP's default correctly rounded one.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mf {

// torch.bucketize(p, edges, right=True): number of edges <= p, NaN -> n
// (ATen upper bound: `if (!(mid_val > val)) start = mid + 1`).  The edges come
// from torch.arange on the host and are data, not a formula, so
// the uniform-grid estimate below is only a starting guess: the four edges
// around it are fetched together and counted (no loop, no branch), which is
// the answer whenever it lies inside that window and the window's ends prove
// it (the guess is off by one at most for edges that come from arange); any
// other case, and any grid of fewer than four edges, falls back to bisection.
__device__ __forceinline__ int upper_bound_bisect(const float *__restrict__ b, int n, float p)
{
    int start = 0, end = n;
    while (start < end) {
        const int mid = start + ((end - start) >> 1);
        if (!(b[mid] > p)) start = mid + 1; else end = mid;
    }
    return start;
}

__device__ __forceinline__ int upper_bound_edges(const float *__restrict__ b, int n, float p)
{
    if (!(p == p)) return n;
    if (n < 4) return upper_bound_bisect(b, n, p);
    const float b0 = b[0];
    const float step = b[1] - b0;
    const float t = (p - b0) * __builtin_amdgcn_rcpf(step);   // a guess only: verified against the edges below
    int g;                                                     // guess of the answer
    if (!(t >= 0.0f)) g = 0;
    else if (t >= (float)n) g = n;
    else g = (int)t + 1;
    int lo = g - 2;                                            // window of four edges lo .. lo + 3 around the guess
    lo = lo < 0 ? 0 : (lo > n - 4 ? n - 4 : lo);
    const float e0 = b[lo], e1 = b[lo + 1], e2 = b[lo + 2], e3 = b[lo + 3];
    const int cnt = (int)(e0 <= p) + (int)(e1 <= p) + (int)(e2 <= p) + (int)(e3 <= p);   // ascending edges: a prefix
    // the count is the answer if the edge before the window is known to be <= p (there is none, or one inside the
    // window already is) and the edge after it is known to be > p (there is none, or one inside the window already is)
    const bool ok = (cnt > 0 || lo == 0) && (cnt < 4 || lo + 4 == n);
    if (ok) return lo + cnt;
    return upper_bound_bisect(b, n, p);
}

struct Bins {
    const float *b0, *b1, *b2;   // edges of world axis 0, 1, 2
    int n0, n1, n2;              // edge counts
};

// bin_rays for one point (projection.py:182-229).  k1 / r1 come back already
// flipped (n1 - 2 - k, 1 - r) like the reference returns them.
__device__ __forceinline__ bool bin_point(const Bins &B, float p0, float p1, float p2, float depth,
                                          float min_d, float max_d,
                                          int &k0, int &k1, int &k2, float &r0, float &r1, float &r2)
{
    k0 = upper_bound_edges(B.b0, B.n0, p0) - 1;
    k1 = upper_bound_edges(B.b1, B.n1, p1) - 1;
    k2 = upper_bound_edges(B.b2, B.n2, p2) - 1;
    const bool ok = (depth >= min_d) && (depth <= max_d) &&
                    k0 >= 0 && k0 < B.n0 - 1 && k1 >= 0 && k1 < B.n1 - 1 && k2 >= 0 && k2 < B.n2 - 1;
    r0 = r1 = r2 = 0.0f;
    if (ok) {
        float lo, hi;
        lo = B.b0[k0]; hi = B.b0[k0 + 1]; r0 = (p0 - lo) / (hi - lo);
        lo = B.b1[k1]; hi = B.b1[k1 + 1]; r1 = (p1 - lo) / (hi - lo);
        lo = B.b2[k2]; hi = B.b2[k2 + 1]; r2 = (p2 - lo) / (hi - lo);
        r1 = 1.0f - r1;
    }
    k1 = B.n1 - 2 - k1;
    return ok;
}

// transform_rays for one ray (projection.py:109-110): ((r0*R[i][0] + r1*R[i][1]) + r2*R[i][2])
__device__ __forceinline__ void rotate_ray(const float *__restrict__ R, float c0, float c1, float c2,
                                           float &q0, float &q1, float &q2)
{
    q0 = (c0 * R[0] + c1 * R[1]) + c2 * R[2];
    q1 = (c0 * R[3] + c1 * R[4]) + c2 * R[5];
    q2 = (c0 * R[6] + c1 * R[7]) + c2 * R[8];
}

// One axis of the 8-corner footprint (projection.py:280-316).
struct AxisFoot { int lo, hi; float wlo, whi; };

__device__ __forceinline__ AxisFoot axis_foot(int k, float r, int size)
{
    AxisFoot a;
    if (r < 0.5f) {
        a.lo = k - 1 < 0 ? 0 : k - 1; a.hi = k;
        a.wlo = 0.5f - r; a.whi = r + 0.5f;
    } else {
        a.lo = k; a.hi = k + 1 > size - 1 ? size - 1 : k + 1;
        a.wlo = 1.5f - r; a.whi = r - 0.5f;
    }
    return a;
}

// corner weight (projection.py:319-323): 1e-9 + (w0*w1)*w2
__device__ __forceinline__ float corner_weight(float w0, float w1, float w2)
{
    float p = w0 * w1;
    p = p * w2;
    return 1e-9f + p;
}

}  // namespace mf
