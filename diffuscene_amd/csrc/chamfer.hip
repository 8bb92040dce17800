// 3-D Chamfer distance: the reference's only native kernels, ChamferDistancePytorch/chamfer3D/
// chamfer3D.cu (NmDistanceKernel :12-131, NmDistanceGradKernel :149-171), used by the FoldingNet shape auto-encoder
// loss (scene_synthesis/networks/foldingnet_autoencoder.py:381-383).
//
// Forward: brute-force nearest neighbour, VALU-bound (8 ops per point pair).  One wave owns 64*QPT query points
// (QPT per lane, so every LDS read of a target point feeds QPT distance evaluations); the target cloud streams through
// LDS in SoA tiles that every lane reads at the same address (LDS broadcast, no bank conflicts).  Strict '<' in index
// order: ties -> lowest index, as the reference.  d = ((dx*dx + dy*dy) + dz*dz), products rounded separately
// (-ffp-contract=off), bit-identical to oracle/chamfer_ref.py.
//
// Backward: the reference scatters with fp32 atomicAdd (order-dependent sums).  Here every point gathers its own
// contributions -- its direct term plus, scanning the other cloud's index array, the points that chose it -- in index
// order, accumulated in fp64: deterministic, no atomics, same O(n*m) integer scan cost as the forward.
#include "dsc_common.h"

namespace {

constexpr int TILE = 1024;   // target points per LDS tile (12 KB)

template <int QPT>           // queries per lane
__global__ __launch_bounds__(256) void chamfer_nn_kernel(const float* __restrict__ q, int n, const float* __restrict__ t,
                                                        int m, float* __restrict__ dist, int* __restrict__ idx) {
    __shared__ float sx[TILE], sy[TILE], sz[TILE];
    const int b = blockIdx.y;
    const float* qb = q + (long)b * n * 3;
    const float* tb = t + (long)b * m * 3;
    const int j0 = blockIdx.x * (256 * QPT) + threadIdx.x;
    float qx[QPT], qy[QPT], qz[QPT], best[QPT];
    int bi[QPT];
#pragma unroll
    for (int u = 0; u < QPT; ++u) {
        const int j = j0 + u * 256;
        const int jc = j < n ? j : n - 1;
        qx[u] = qb[jc * 3 + 0]; qy[u] = qb[jc * 3 + 1]; qz[u] = qb[jc * 3 + 2];
        best[u] = INFINITY; bi[u] = 0;
    }
    for (int k0 = 0; k0 < m; k0 += TILE) {
        const int cnt = min(TILE, m - k0);
        __syncthreads();
        for (int k = threadIdx.x; k < cnt; k += 256) {
            sx[k] = tb[(long)(k0 + k) * 3 + 0]; sy[k] = tb[(long)(k0 + k) * 3 + 1]; sz[k] = tb[(long)(k0 + k) * 3 + 2];
        }
        __syncthreads();
#pragma unroll 4
        for (int k = 0; k < cnt; ++k) {
            const float tx = sx[k], ty = sy[k], tz = sz[k];
#pragma unroll
            for (int u = 0; u < QPT; ++u) {
                const float dx = tx - qx[u], dy = ty - qy[u], dz = tz - qz[u];
                const float d = (dx * dx + dy * dy) + dz * dz;
                if (d < best[u]) { best[u] = d; bi[u] = k0 + k; }
            }
        }
    }
#pragma unroll
    for (int u = 0; u < QPT; ++u) {
        const int j = j0 + u * 256;
        if (j < n) { dist[(long)b * n + j] = best[u]; idx[(long)b * n + j] = bi[u]; }
    }
}

// grad of cloud A: direct term 2 gA[j] (a_j - b_{idxA[j]}) minus sum over k with idxB[k] == j of 2 gB[k] (b_k - a_j)
__global__ __launch_bounds__(256) void chamfer_grad_kernel(const float* __restrict__ a, int n, const float* __restrict__ bb,
                                                          int m, const float* __restrict__ ga, const float* __restrict__ gb,
                                                          const int* __restrict__ idxa, const int* __restrict__ idxb,
                                                          float* __restrict__ grad_a) {
    __shared__ int si[TILE];
    const int b = blockIdx.y;
    const float* ab = a + (long)b * n * 3;
    const float* bp = bb + (long)b * m * 3;
    const int j = blockIdx.x * 256 + threadIdx.x;
    const bool live = j < n;
    const int jc = live ? j : n - 1;
    const float ax = ab[jc * 3 + 0], ay = ab[jc * 3 + 1], az = ab[jc * 3 + 2];
    const int j2 = idxa[(long)b * n + jc];
    const float g = ga[(long)b * n + jc] * 2.0f;
    // reference rounding of each term: g * (x1 - x2) in fp32
    double sxg = (double)(g * (ax - bp[j2 * 3 + 0]));
    double syg = (double)(g * (ay - bp[j2 * 3 + 1]));
    double szg = (double)(g * (az - bp[j2 * 3 + 2]));
    for (int k0 = 0; k0 < m; k0 += TILE) {
        const int cnt = min(TILE, m - k0);
        __syncthreads();
        for (int k = threadIdx.x; k < cnt; k += 256) si[k] = idxb[(long)b * m + k0 + k];
        __syncthreads();
        for (int k = 0; k < cnt; ++k) {
            if (si[k] == j && live) {
                const int kk = k0 + k;
                const float g2 = gb[(long)b * m + kk] * 2.0f;
                sxg -= (double)(g2 * (bp[kk * 3 + 0] - ax));
                syg -= (double)(g2 * (bp[kk * 3 + 1] - ay));
                szg -= (double)(g2 * (bp[kk * 3 + 2] - az));
            }
        }
    }
    if (live) {
        float* o = grad_a + ((long)b * n + j) * 3;
        o[0] = (float)sxg; o[1] = (float)syg; o[2] = (float)szg;
    }
}

// 4 queries per lane amortise the LDS reads, but only when that still leaves >= 2 blocks per CU
void launch_nn(const float* q, int n, const float* t, int m, float* dist, int* idx, int b, hipStream_t s) {
    const long blocks4 = (long)((n + 1023) / 1024) * b;
    if (blocks4 >= 512)
        hipLaunchKernelGGL(chamfer_nn_kernel<4>, dim3((n + 1023) / 1024, b), dim3(256), 0, s, q, n, t, m, dist, idx);
    else
        hipLaunchKernelGGL(chamfer_nn_kernel<1>, dim3((n + 255) / 256, b), dim3(256), 0, s, q, n, t, m, dist, idx);
}

}  // namespace

extern "C" int dsc_chamfer3d_forward_f32(const float* xyz1, const float* xyz2, float* dist1, float* dist2, int32_t* idx1,
                                         int32_t* idx2, int32_t b, int32_t n, int32_t m, dsc_stream_t stream) {
    if (!xyz1 || !xyz2 || !dist1 || !dist2 || !idx1 || !idx2 || b <= 0 || n <= 0 || m <= 0) return DSC_EINVAL;
    if (b > 65535) return DSC_ERANGE;
    hipStream_t s = static_cast<hipStream_t>(stream);
    DSC_CLEAR_STALE_ERROR();
    launch_nn(xyz1, n, xyz2, m, dist1, idx1, b, s);
    launch_nn(xyz2, m, xyz1, n, dist2, idx2, b, s);
    DSC_LAUNCH_CHECK();
    return 0;
}

extern "C" int dsc_chamfer3d_backward_f32(const float* xyz1, const float* xyz2, const float* graddist1,
                                          const float* graddist2, const int32_t* idx1, const int32_t* idx2,
                                          float* gradxyz1, float* gradxyz2, int32_t b, int32_t n, int32_t m,
                                          dsc_stream_t stream) {
    if (!xyz1 || !xyz2 || !graddist1 || !graddist2 || !idx1 || !idx2 || !gradxyz1 || !gradxyz2) return DSC_EINVAL;
    if (b <= 0 || n <= 0 || m <= 0) return DSC_EINVAL;
    if (b > 65535) return DSC_ERANGE;
    hipStream_t s = static_cast<hipStream_t>(stream);
    DSC_CLEAR_STALE_ERROR();
    hipLaunchKernelGGL(chamfer_grad_kernel, dim3((n + 255) / 256, b), dim3(256), 0, s, xyz1, n, xyz2, m, graddist1,
                       graddist2, idx1, idx2, gradxyz1);
    hipLaunchKernelGGL(chamfer_grad_kernel, dim3((m + 255) / 256, b), dim3(256), 0, s, xyz2, m, xyz1, n, graddist2,
                       graddist1, idx2, idx1, gradxyz2);
    DSC_LAUNCH_CHECK();
    return 0;
}
