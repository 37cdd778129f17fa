// match.hip — pairwise instance-feature distances between two maps.
//
// Replaces  torch.linalg.norm(f0.unsqueeze(1) - f1.unsqueeze(0), dim=2)
// at /root/reference/mass/utils/experimentation.py:261-265 (features, D = 256)
// and :277-280 (3-d goals).  The reference materialises an [N0, N1, D] temporary;
// here one kernel produces the [N0, N1] cost matrix directly.
//
//   MF_METRIC_L2       difference form sum_k (a_k - b_k)^2, the reference's own
//                      arithmetic (no cancellation when the two instances are
//                      the same object seen twice, which is the real use case).
//   MF_METRIC_L2_GEMM  |a|^2 + |b|^2 - 2 a.b with the a.b contraction on the
//   MF_METRIC_COSINE   fp32 matrix cores (v_mfma_f32_32x32x2_f32: exact fp32
//                      FMA chain, so no precision is given up for using MFMA).
#include "common.h"

namespace mf {

constexpr int DT = 16;      // output tile edge of the difference kernel
constexpr int DK = 64;      // k chunk

__global__ __launch_bounds__(DT * DT) void pairwise_diff_kernel(const float *__restrict__ f0, int n0,
                                                                  const float *__restrict__ f1, int n1, int d,
                                                                  float *__restrict__ out)
{
    __shared__ float A[DT][DK + 1];
    __shared__ float B[DT][DK + 1];
    const int tx = threadIdx.x % DT, ty = threadIdx.x / DT;
    const int i0 = blockIdx.y * DT, j0 = blockIdx.x * DT;
    float acc = 0.0f;
    for (int k0 = 0; k0 < d; k0 += DK) {
        for (int t = threadIdx.x; t < DT * DK; t += DT * DT) {
            const int r = t / DK, k = t - r * DK;
            A[r][k] = (i0 + r < n0 && k0 + k < d) ? f0[(size_t)(i0 + r) * d + k0 + k] : 0.0f;
            B[r][k] = (j0 + r < n1 && k0 + k < d) ? f1[(size_t)(j0 + r) * d + k0 + k] : 0.0f;
        }
        __syncthreads();
#pragma unroll 8
        for (int k = 0; k < DK; ++k) {
            const float df = A[ty][k] - B[tx][k];
            acc += df * df;
        }
        __syncthreads();
    }
    if (i0 + ty < n0 && j0 + tx < n1) out[(size_t)(i0 + ty) * n1 + j0 + tx] = sqrtf(acc);
}

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int GT = 64;      // output tile edge per workgroup (4 waves x 32x32)
constexpr int GK = 32;      // k chunk staged in LDS

// One 64x64 tile of a.b per workgroup; wave w owns the 32x32 sub-tile
// (w >> 1, w & 1).  v_mfma_f32_32x32x2_f32 operand maps: lane l supplies
// A[i = l & 31][k = l >> 5] and B[k = l >> 5][j = l & 31]; result register r
// of lane l is D[(r & 3) + 8 * (r >> 2) + 4 * (l >> 5)][l & 31].
__global__ __launch_bounds__(256) void pairwise_gemm_kernel(const float *__restrict__ f0, int n0,
                                                             const float *__restrict__ f1, int n1, int d,
                                                             float *__restrict__ out, int metric)
{
    __shared__ float S[2][GT][GK + 1];
    __shared__ float nrm[2][GT];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int wr = w >> 1, wc = w & 1;
    const int i0 = blockIdx.y * GT, j0 = blockIdx.x * GT;
    f32x16 acc;
    for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
    float nsum = 0.0f;
    for (int k0 = 0; k0 < d; k0 += GK) {
        for (int t = tid; t < GT * GK; t += 256) {
            const int r = t / GK, k = t - r * GK;
            S[0][r][k] = (i0 + r < n0 && k0 + k < d) ? f0[(size_t)(i0 + r) * d + k0 + k] : 0.0f;
            S[1][r][k] = (j0 + r < n1 && k0 + k < d) ? f1[(size_t)(j0 + r) * d + k0 + k] : 0.0f;
        }
        __syncthreads();
        if (tid < 2 * GT) {
            const float *row = S[tid >> 6][tid & 63];
            for (int k = 0; k < GK; ++k) nsum += row[k] * row[k];
        }
        const int ar = wr * 32 + (lane & 31), br = wc * 32 + (lane & 31), kh = lane >> 5;
#pragma unroll
        for (int kk = 0; kk < GK; kk += 2)
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(S[0][ar][kk + kh], S[1][br][kk + kh], acc, 0, 0, 0);
        __syncthreads();
    }
    if (tid < 2 * GT) nrm[tid >> 6][tid & 63] = nsum;
    __syncthreads();
    const int col = wc * 32 + (lane & 31);
    for (int r = 0; r < 16; ++r) {
        const int row = wr * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (i0 + row < n0 && j0 + col < n1) {
            const float na = nrm[0][row], nb = nrm[1][col], dot = acc[r];
            float v;
            if (metric == MF_METRIC_COSINE) v = 1.0f - dot / fmaxf(sqrtf(na) * sqrtf(nb), 1e-8f);
            else v = sqrtf(fmaxf(na + nb - 2.0f * dot, 0.0f));
            out[(size_t)(i0 + row) * n1 + j0 + col] = v;
        }
    }
}

}  // namespace mf

using namespace mf;

extern "C" int mf_pairwise_distance(const float *f0, int32_t n0, const float *f1, int32_t n1, int32_t d, float *out,
                                    int32_t metric, void *stream)
{
    if (n0 < 0 || n1 < 0 || d < 1) return fail(MF_ERR_INVALID, "bad shape n0=%d n1=%d d=%d", n0, n1, d);
    if (n0 == 0 || n1 == 0) return MF_OK;
    if (!f0 || !f1 || !out) return fail(MF_ERR_INVALID, "NULL pointer");
    hipStream_t st = (hipStream_t)stream;
    if (metric == MF_METRIC_L2) {
        hipLaunchKernelGGL(pairwise_diff_kernel, dim3((n1 + DT - 1) / DT, (n0 + DT - 1) / DT), dim3(DT * DT), 0, st,
                           f0, n0, f1, n1, d, out);
        MF_LAUNCH_CHECK("pairwise_diff_kernel");
    } else if (metric == MF_METRIC_L2_GEMM || metric == MF_METRIC_COSINE) {
        hipLaunchKernelGGL(pairwise_gemm_kernel, dim3((n1 + GT - 1) / GT, (n0 + GT - 1) / GT), dim3(256), 0, st,
                           f0, n0, f1, n1, d, out, metric);
        MF_LAUNCH_CHECK("pairwise_gemm_kernel");
    } else {
        return fail(MF_ERR_INVALID, "unknown metric %d", metric);
    }
    return MF_OK;
}
