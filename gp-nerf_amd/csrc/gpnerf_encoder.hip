// gpnerf_encoder.hip -- the two memory-bound glue operators of the image encoder (libs/encoders/UNet.py), per frame:
//   * affine InstanceNorm2d (no running statistics) fused with the residual add and the activation that follow it
//     (UNet.py:38-53: relu(bn(conv)) / relu(bn2(conv2) + identity); :117-120: elu(bn(conv)))
//   * bilinear x2 upsampling with align_corners=True (UNet.py:129)
// The convolutions themselves stay library calls (MIOpen); stock torch spends ~35 launches of repeat / batch-norm / clamp /
// add kernels and a slow generic upsample on what these two kernels do.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/gpnerf_hip.h"

namespace {

__device__ __forceinline__ float block_sum(float v, float* red) {
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, nw = blockDim.x >> 6;
    __syncthreads();                       // red[] may still be read from the previous reduction
    if (lane == 0) red[wave] = v;
    __syncthreads();
    float t = 0.f;
    for (int i = 0; i < nw; ++i) t += red[i];
    return t;
}

// one workgroup per (image, channel) plane of an NCHW tensor: mean, biased variance (two passes over the plane, which is
// L2-resident after the first), then y = act((x - mean) * rstd * gamma + beta [+ residual])
__global__ void __launch_bounds__(1024) instance_norm_act_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                                const float* __restrict__ beta, const float* __restrict__ residual,
                                                                const int C, const long hw, const float eps, const int act,
                                                                float* __restrict__ out) {
    __shared__ float red[16];
    const long plane = blockIdx.x;
    const int c = (int)(plane % C);
    const float* p = x + plane * hw;
    const bool vec = (hw & 3) == 0;
    float s = 0.f;
    if (vec) {
        const float4* p4 = reinterpret_cast<const float4*>(p);
        for (long i = threadIdx.x; i < (hw >> 2); i += blockDim.x) { const float4 v = p4[i]; s += (v.x + v.y) + (v.z + v.w); }
    } else {
        for (long i = threadIdx.x; i < hw; i += blockDim.x) s += p[i];
    }
    const float mean = block_sum(s, red) / (float)hw;
    float q = 0.f;
    if (vec) {
        const float4* p4 = reinterpret_cast<const float4*>(p);
        for (long i = threadIdx.x; i < (hw >> 2); i += blockDim.x) {
            const float4 v = p4[i];
            const float a = v.x - mean, b = v.y - mean, cc = v.z - mean, d = v.w - mean;
            q += (a * a + b * b) + (cc * cc + d * d);
        }
    } else {
        for (long i = threadIdx.x; i < hw; i += blockDim.x) { const float a = p[i] - mean; q += a * a; }
    }
    const float var = block_sum(q, red) / (float)hw;
    const float g = gamma[c] / sqrtf(var + eps), b0 = beta[c] - mean * g;
    const float* r = residual ? residual + plane * hw : nullptr;
    float* o = out + plane * hw;
    auto f = [&](float v, float rv) {
        float y = fmaf(v, g, b0) + rv;
        if (act == 1) y = fmaxf(y, 0.f);
        else if (act == 2) y = y > 0.f ? y : expm1f(y);
        return y;
    };
    if (vec) {
        const float4* p4 = reinterpret_cast<const float4*>(p);
        const float4* r4 = reinterpret_cast<const float4*>(r);
        float4* o4 = reinterpret_cast<float4*>(o);
        for (long i = threadIdx.x; i < (hw >> 2); i += blockDim.x) {
            const float4 v = p4[i];
            const float4 rv = r ? r4[i] : make_float4(0.f, 0.f, 0.f, 0.f);
            o4[i] = make_float4(f(v.x, rv.x), f(v.y, rv.y), f(v.z, rv.z), f(v.w, rv.w));
        }
    } else {
        for (long i = threadIdx.x; i < hw; i += blockDim.x) o[i] = f(p[i], r ? r[i] : 0.f);
    }
}

// F.interpolate(scale_factor=2, mode='bilinear', align_corners=True) on [planes][H][W]: src = dst * (in - 1) / (out - 1)
__global__ void upsample2x_kernel(const float* __restrict__ x, const long planes, const int H, const int W, float* __restrict__ out) {
    const int OH = 2 * H, OW = 2 * W;
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= planes * OH * OW) return;
    const int ox = (int)(i % OW), oy = (int)((i / OW) % OH);
    const long pl = i / ((long)OW * OH);
    const float sy = OH > 1 ? (float)(H - 1) / (float)(OH - 1) : 0.f, sx = OW > 1 ? (float)(W - 1) / (float)(OW - 1) : 0.f;
    const float fy = sy * (float)oy, fx = sx * (float)ox;
    const int y0 = (int)fy, x0 = (int)fx;
    const int y1 = min(y0 + 1, H - 1), x1 = min(x0 + 1, W - 1);
    const float ty = fy - (float)y0, tx = fx - (float)x0;
    const float* p = x + pl * (long)H * W;
    const float top = p[(long)y0 * W + x0] * (1.f - tx) + p[(long)y0 * W + x1] * tx;
    const float bot = p[(long)y1 * W + x0] * (1.f - tx) + p[(long)y1 * W + x1] * tx;
    out[i] = top * (1.f - ty) + bot * ty;
}

}  // namespace

extern "C" {

int gpnerf_instance_norm_act(const float* x, const float* gamma, const float* beta, const float* residual, int32_t n, int32_t c,
                             int64_t hw, float eps, int32_t act, float* out, void* stream) {
    if (n == 0 || c == 0 || hw == 0) return GPNERF_OK;
    if (!x || !gamma || !beta || !out || n < 0 || c < 0 || hw < 0 || act < 0 || act > 2) return GPNERF_E_ARG;
    // a plane is one workgroup: wide workgroups for the large early planes (memory-level parallelism), narrow for the rest
    const int threads = hw >= 32768 ? 1024 : (hw >= 4096 ? 512 : 256);
    hipLaunchKernelGGL(instance_norm_act_kernel, dim3((unsigned)((long)n * c)), dim3(threads), 0, reinterpret_cast<hipStream_t>(stream),
                       x, gamma, beta, residual, (int)c, (long)hw, eps, (int)act, out);
    return hipGetLastError() == hipSuccess ? GPNERF_OK : GPNERF_E_LAUNCH;
}

int gpnerf_upsample2x(const float* x, int64_t planes, int32_t h, int32_t w, float* out, void* stream) {
    if (planes == 0) return GPNERF_OK;
    if (!x || !out || planes < 0 || h < 1 || w < 1) return GPNERF_E_ARG;
    const long total = (long)planes * h * w * 4;
    hipLaunchKernelGGL(upsample2x_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), x,
                       (long)planes, (int)h, (int)w, out);
    return hipGetLastError() == hipSuccess ? GPNERF_OK : GPNERF_E_LAUNCH;
}

}  // extern "C"
