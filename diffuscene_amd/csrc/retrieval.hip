// Shape retrieval after sampling: for every generated object, the nearest 3D-FUTURE model of the
// predicted class in the 32-d latent shape-code space -- reference ThreedFutureDataset.get_closest_furniture_to_objfeats
// / ..._and_size (scene_synthesis/datasets/threed_future_dataset.py:49-77), called once per box from
// scene_synthesis/utils.py:80-110.  Brute-force scan, HBM/L2-bound: one wave per query, lanes stride over the
// database rows, argmin by wave shuffles.
//
// Bit-exact index parity with the numpy reference: squared differences are rounded products (this file is compiled with
// -ffp-contract=off) summed in numpy's pairwise order for a contiguous fp32 vector of 32 (8 strided partial sums,
// combined as ((r0+r1)+(r2+r3))+((r4+r5)+(r6+r7))); size keys are float64 sums of 3 terms; ties go to the lowest
// database index (Python's stable sort / np.lexsort).
#include "dsc_common.h"

namespace {

__device__ __forceinline__ float sqdist32_numpy_order(const float* __restrict__ a, const float* __restrict__ q) {
    float r[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { const float d = a[j] - q[j]; r[j] = d * d; }
#pragma unroll
    for (int i = 8; i < 32; i += 8)
#pragma unroll
        for (int j = 0; j < 8; ++j) { const float d = a[i + j] - q[i + j]; const float s = d * d; r[j] = r[j] + s; }
    return ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
}

__global__ __launch_bounds__(256) void retrieve_kernel(const float* __restrict__ qfeat, const int* __restrict__ qlabel,
                                                      const double* __restrict__ qsize, const float* __restrict__ db,
                                                      const int* __restrict__ dblabel, const double* __restrict__ dbsize,
                                                      int nq, int ndb, int* __restrict__ out_idx,
                                                      float* __restrict__ out_dist) {
    const int lane = threadIdx.x & 63;
    const int qi = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (qi >= nq) return;
    float q[32];
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        const f32x4 t4 = *reinterpret_cast<const f32x4*>(qfeat + (long)qi * 32 + c * 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) q[c * 4 + e] = t4[e];
    }
    const int lab = qlabel[qi];
    double qs[3] = {0.0, 0.0, 0.0};
    if (qsize) { qs[0] = qsize[(long)qi * 3]; qs[1] = qsize[(long)qi * 3 + 1]; qs[2] = qsize[(long)qi * 3 + 2]; }
    double best_s = 1e300;
    float best_f = INFINITY;
    int best_i = 0x7fffffff;
    for (int j = lane; j < ndb; j += 64) {
        if (dblabel[j] != lab) continue;
        float row[32];
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            const f32x4 t4 = *reinterpret_cast<const f32x4*>(db + (long)j * 32 + c * 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) row[c * 4 + e] = t4[e];
        }
        const float f = sqdist32_numpy_order(row, q);
        double sk = 0.0;
        if (qsize) {
            const double d0 = dbsize[(long)j * 3] - qs[0], d1 = dbsize[(long)j * 3 + 1] - qs[1], d2 = dbsize[(long)j * 3 + 2] - qs[2];
            const double p0 = d0 * d0, p1 = d1 * d1, p2 = d2 * d2;
            sk = (p0 + p1) + p2;
        }
        // lexicographic (size key, feature key, index); rows visited by a lane are in increasing index order
        if (sk < best_s || (sk == best_s && f < best_f)) { best_s = sk; best_f = f; best_i = j; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const double os = __shfl_xor(best_s, o, 64);
        const float of = __shfl_xor(best_f, o, 64);
        const int oi = __shfl_xor(best_i, o, 64);
        const bool better = (os < best_s) || (os == best_s && (of < best_f || (of == best_f && oi < best_i)));
        if (better) { best_s = os; best_f = of; best_i = oi; }
    }
    if (lane == 0) {
        out_idx[qi] = (best_i == 0x7fffffff) ? -1 : best_i;
        if (out_dist) out_dist[qi] = best_f;
    }
}

}  // namespace

extern "C" int dsc_retrieve_nearest_f32(const float* query_feats, const int32_t* query_labels, const double* query_sizes,
                                        const float* db_feats, const int32_t* db_labels, const double* db_sizes,
                                        int32_t n_query, int32_t n_db, int32_t feat_dim, int32_t* out_index,
                                        float* out_dist, dsc_stream_t stream) {
    if (!query_feats || !query_labels || !db_feats || !db_labels || !out_index || n_query < 1 || n_db < 1) return DSC_EINVAL;
    if (feat_dim != 32) return DSC_ERANGE;
    if ((query_sizes != nullptr) != (db_sizes != nullptr)) return DSC_EINVAL;
    if (!dsc_aligned16(query_feats) || !dsc_aligned16(db_feats)) return DSC_EALIGN;
    DSC_CLEAR_STALE_ERROR();
    hipLaunchKernelGGL(retrieve_kernel, dim3((n_query + 3) / 4), dim3(256), 0, static_cast<hipStream_t>(stream),
                       query_feats, query_labels, query_sizes, db_feats, db_labels, db_sizes, n_query, n_db, out_index,
                       out_dist);
    DSC_LAUNCH_CHECK();
    return 0;
}
