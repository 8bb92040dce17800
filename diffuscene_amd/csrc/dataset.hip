// Training input pipeline on the device.  The cached 3D-FRONT rooms live in HBM as ragged arrays; one
// launch gathers B scenes and applies, per object row, what the reference's dataset decorator stack does per sample on
// the CPU inside DataLoader workers:
//   RotationAugmentation  (scene_synthesis/datasets/threed_front_dataset.py:313-371)
//   Jitter                (:559-567)
//   Scale_CosinAngle_ObjfeatsNorm (:481-513; Scale.scale :377-382)
//   Permutation           (:570-584)
//   Diffusion padding     (:888-925)
// and writes the (B, N, C) batch already in the denoiser's channel order (diffusion_scene_layout_ddpm.py:148-154).
// HBM-bound gather: B*N rows of <= 65 floats; one wave per output row, lanes over channels.
//
// Arithmetic follows numpy's promotion rules for the reference's expressions (float32 arrays, Python-float scalars,
// float64 bounds; see oracle/dataset_ref.py) -- compiled with -ffp-contract=off so a*b+c is two roundings as in numpy.
#include "dsc_common.h"

namespace {

struct EncArgs {
    const int64_t* offsets;
    const float* cls;
    const float* trans;
    const float* sizes;
    const float* angles;
    const float* feats;
    const int64_t* scene;
    const int* order;
    const double* rot;
    const double* jitter;
    float* out;
    int64_t* length;
    int64_t ld_out;
    int n_cls_in, feat_dim, permute_feats, B, N;
    double t_lo[3], t_hi[3], s_lo[3], s_hi[3], angle_min, f_lo, f_hi;
};

// Scale.scale: float32 value, clip / normalise in float64, final float32 cast by the Diffusion wrapper
__device__ __forceinline__ float scale_f(float x32, double lo, double hi) {
    double X = (double)x32;
    X = X < lo ? lo : (X > hi ? hi : X);
    X = (X - lo) / (hi - lo);
    return (float)(2.0 * X - 1.0);
}

// numpy's float remainder (sign of the divisor)
__device__ __forceinline__ double np_mod(double a, double b) {
    double m = fmod(a, b);
    if (m != 0.0) {
        if ((b < 0.0) != (m < 0.0)) m += b;
    } else {
        m = copysign(0.0, b);
    }
    return m;
}

__global__ __launch_bounds__(256) void encode_scene_batch_kernel(EncArgs a) {
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= (long)a.B * a.N) return;
    const int b = (int)(row / a.N), j = (int)(row % a.N);
    const int64_t s = a.scene[b];
    const int64_t start = a.offsets[s];
    const int L = (int)(a.offsets[s + 1] - start);
    if (j == 0 && lane == 0) a.length[b] = L;
    const int n_cls = a.n_cls_in - 1;
    const int C = 8 + n_cls + a.feat_dim;
    float* o = a.out + row * a.ld_out;
    if (j >= L) {                                            // end symbol
        for (int c = lane; c < C; c += 64) o[c] = (c >= 8 && c < 8 + n_cls) ? (c == 8 + n_cls - 1 ? 1.0f : -1.0f) : 0.0f;
        return;
    }
    const int64_t src = start + (a.order ? a.order[(long)b * a.N + j] : j);
    const int64_t src_f = (a.order && !a.permute_feats) ? start + j : src;
    const bool has_rot = a.rot != nullptr, has_jit = a.jitter != nullptr;
    const double rot = has_rot ? a.rot[b] : 0.0;
    for (int c = lane; c < C; c += 64) {
        float v;
        if (c < 3) {
            const float* t = a.trans + src * 3;
            float x32;
            if (has_rot) {
                // v.dot(R) in float64, R = [[cos,0,-sin],[0,1,0],[sin,0,cos]]
                const double cs = cos(rot), sn = sin(rot);
                const double v0 = t[0], v1 = t[1], v2 = t[2];
                double r;
                if (c == 0) r = (v0 * cs + v1 * 0.0) + v2 * sn;
                else if (c == 1) r = (v0 * 0.0 + v1 * 1.0) + v2 * 0.0;
                else r = (v0 * (-sn) + v1 * 0.0) + v2 * cs;
                if (has_jit) r = r + a.jitter[b * 3 + 0];
                x32 = (float)r;
            } else {
                x32 = t[c];
                if (has_jit) x32 = x32 + (float)a.jitter[b * 3 + 0];
            }
            v = scale_f(x32, a.t_lo[c], a.t_hi[c]);
        } else if (c < 6) {
            float x32 = a.sizes[src * 3 + (c - 3)];
            if (has_jit) x32 = x32 + (float)a.jitter[b * 3 + 1];
            v = scale_f(x32, a.s_lo[c - 3], a.s_hi[c - 3]);
        } else if (c < 8) {
            const float a32 = a.angles[src];
            double ang;
            float af = a32;
            if (has_rot) {
                // (v + rot) is a float32 add (Python-float scalar), the rest runs in float64 (0-d float64 angle_min)
                const float a1 = a32 + (float)rot;
                ang = np_mod((double)a1 - a.angle_min, 2.0 * 3.14159265358979323846) + a.angle_min;
                if (has_jit) ang = ang + a.jitter[b * 3 + 2];
            } else {
                if (has_jit) af = af + (float)a.jitter[b * 3 + 2];
                ang = (double)af;
            }
            v = (float)(c == 6 ? cos(ang) : sin(ang));
        } else if (c < 8 + n_cls) {
            // drop the start column (index n_cls_in-2), keep the end column; {0,1} -> {-1,+1}
            const int k = c - 8;
            const float x = a.cls[src * a.n_cls_in + (k == n_cls - 1 ? a.n_cls_in - 1 : k)];
            v = x * 2.0f - 1.0f;
        } else {
            v = scale_f(a.feats[src_f * a.feat_dim + (c - 8 - n_cls)], a.f_lo, a.f_hi);
        }
        o[c] = v;
    }
}

}  // namespace

extern "C" int dsc_encode_scene_batch_f32(const int64_t* offsets, const float* class_labels, const float* translations,
                                          const float* sizes, const float* angles, const float* objfeats,
                                          int32_t n_cls_in, int32_t feat_dim, const int64_t* scene, const int32_t* order,
                                          const double* rot, const double* jitter, int32_t permute_objfeats,
                                          const double* bounds, float* out, int64_t ld_out, int64_t* length, int32_t b,
                                          int32_t n, dsc_stream_t stream) {
    if (!offsets || !class_labels || !translations || !sizes || !angles || !scene || !bounds || !out || !length)
        return DSC_EINVAL;
    if (b <= 0 || n <= 0 || n_cls_in < 3 || feat_dim < 0 || (feat_dim > 0 && !objfeats)) return DSC_EINVAL;
    if (ld_out < 8 + (n_cls_in - 1) + feat_dim) return DSC_EINVAL;
    EncArgs a;
    a.offsets = offsets; a.cls = class_labels; a.trans = translations; a.sizes = sizes; a.angles = angles;
    a.feats = objfeats; a.scene = scene; a.order = order; a.rot = rot; a.jitter = jitter; a.out = out; a.length = length;
    a.ld_out = ld_out; a.n_cls_in = n_cls_in; a.feat_dim = feat_dim; a.permute_feats = permute_objfeats; a.B = b; a.N = n;
    for (int i = 0; i < 3; ++i) {
        a.t_lo[i] = bounds[i]; a.t_hi[i] = bounds[3 + i]; a.s_lo[i] = bounds[6 + i]; a.s_hi[i] = bounds[9 + i];
    }
    a.angle_min = bounds[12]; a.f_lo = bounds[13]; a.f_hi = bounds[14];
    const long rows = (long)b * n;
    DSC_CLEAR_STALE_ERROR();
    hipLaunchKernelGGL(encode_scene_batch_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0,
                       static_cast<hipStream_t>(stream), a);
    DSC_LAUNCH_CHECK();
    return 0;
}
