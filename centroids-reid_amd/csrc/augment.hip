// Input side of the path: the per-image transforms of datasets/transforms/build.py:16-31 AFTER the Resize, applied to a whole
// uint8 batch on the device in one pass:
//     train:  RandomHorizontalFlip -> Pad(P, fill 0) -> RandomCrop(H, W) -> ToTensor -> Normalize -> RandomErasing
//     test:   ToTensor -> Normalize
// The random draws are made on the host (transforms.py, same generators and order as the reference's pipeline) and arrive as
// eight int32 per image; the kernel is a pure function of (pixels, draws).  Arithmetic of an output value, as torch does it in
// fp32: t = u8 / 255, y = (t - mean[c]) / std[c]; the black border that T.Pad adds lies INSIDE the crop and is normalised like
// any pixel ((0 - mean) / std); RandomErasing writes its fill values into the NORMALISED tensor
// (datasets/transforms/random_erasing.py:47-52: the un-normalised PIXEL_MEAN, a quirk the restatement keeps).
// Output: fp32 NCHW [B, 3, H, W] (the tensor the reference's DataLoader yields), or directly the stem convolution's operand
// (zero-padded NHWC4 [B, H + 8, W + 6, 4] in the compute dtype, image at rows / columns 3..: what creid_image_to_nhwc4_pad
// would make of the NCHW tensor) -- the fp32 batch is then never written.  HBM-bound: 3 bytes in, 12 (or 8 / 16) out per pixel.
#include "common.hpp"

namespace {
struct AugGeom {
  int B, H, W, pad;
  float mean[3], stdv[3], erase[3];
};

// params per image: {flip, crop_top, crop_left, erase, ex, ey, eh, ew}
__device__ __forceinline__ void aug_pixel(const AugGeom& g, const unsigned char* __restrict__ src, const int* __restrict__ prm,
                                          int b, int y, int x, float (&v)[3]) {
  int flip = 0, top = g.pad, left = g.pad, er = 0, ex = 0, ey = 0, eh = 0, ew = 0;   // test mode: the identity crop
  if (prm) {
    const int* p = prm + (int64_t)b * 8;
    flip = p[0]; top = p[1]; left = p[2]; er = p[3]; ex = p[4]; ey = p[5]; eh = p[6]; ew = p[7];
  }
  if (er && (unsigned)(y - ex) < (unsigned)eh && (unsigned)(x - ey) < (unsigned)ew) {
#pragma unroll
    for (int c = 0; c < 3; ++c) v[c] = g.erase[c];
    return;
  }
  const int sy = y + top - g.pad, sx0 = x + left - g.pad;           // position in the (flipped) un-padded image
  unsigned u[3] = {0u, 0u, 0u};                                     // T.Pad's fill
  if ((unsigned)sy < (unsigned)g.H && (unsigned)sx0 < (unsigned)g.W) {
    const int sx = flip ? g.W - 1 - sx0 : sx0;
    const unsigned char* q = src + (((int64_t)b * g.H + sy) * g.W + sx) * 3;
    u[0] = q[0]; u[1] = q[1]; u[2] = q[2];
  }
#pragma unroll
  for (int c = 0; c < 3; ++c) v[c] = ((float)u[c] / 255.0f - g.mean[c]) / g.stdv[c];
}

__global__ __launch_bounds__(256) void augment_nchw_kernel(AugGeom g, const unsigned char* __restrict__ src,
                                                           const int* __restrict__ prm, float* __restrict__ out) {
  const int64_t total = (int64_t)g.B * g.H * g.W;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int x = (int)(i % g.W);
    const int y = (int)((i / g.W) % g.H);
    const int b = (int)(i / ((int64_t)g.W * g.H));
    float v[3];
    aug_pixel(g, src, prm, b, y, x, v);
#pragma unroll
    for (int c = 0; c < 3; ++c) out[(((int64_t)b * 3 + c) * g.H + y) * g.W + x] = v[c];
  }
}

template <typename T> struct Px4;
template <> struct Px4<float> {
  static __device__ __forceinline__ void st(float* p, const float (&v)[4]) {
    *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
  }
};
template <> struct Px4<unsigned short> {
  static __device__ __forceinline__ void st(unsigned short* p, const float (&v)[4]) {
    *reinterpret_cast<uint2*>(p) = make_uint2(f32x2_to_bf16x2_bits(v[0], v[1]), f32x2_to_bf16x2_bits(v[2], v[3]));
  }
};

template <> struct Px4<_Float16> {
  static __device__ __forceinline__ void st(_Float16* p, const float (&v)[4]) {
    *reinterpret_cast<uint2*>(p) = make_uint2(F16T::pack2(v[0], v[1]), F16T::pack2(v[2], v[3]));
  }
};

template <typename T>
__global__ __launch_bounds__(256) void augment_stem_kernel(AugGeom g, const unsigned char* __restrict__ src,
                                                           const int* __restrict__ prm, T* __restrict__ out) {
  const int PH = g.H + 8, PW = g.W + 6;
  const int64_t total = (int64_t)g.B * PH * PW;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int px = (int)(i % PW);
    const int py = (int)((i / PW) % PH);
    const int b = (int)(i / ((int64_t)PW * PH));
    const int y = py - 3, x = px - 3;
    float v[4] = {0.f, 0.f, 0.f, 0.f};                               // the convolution's own zero padding, 4th channel 0
    if ((unsigned)y < (unsigned)g.H && (unsigned)x < (unsigned)g.W) {
      float w[3];
      aug_pixel(g, src, prm, b, y, x, w);
      v[0] = w[0]; v[1] = w[1]; v[2] = w[2];
    }
    Px4<T>::st(out + i * 4, v);
  }
}
}  // namespace

extern "C" int creid_augment_u8(const uint8_t* src_hwc, const int32_t* params, int64_t B, int64_t H, int64_t W, int64_t pad,
                                float mean0, float mean1, float mean2, float std0, float std1, float std2, float erase0,
                                float erase1, float erase2, int32_t layout, int32_t dtype, void* out, void* stream) {
  CREID_CHECK_ARG(src_hwc && out && B > 0 && H > 0 && W > 0 && pad >= 0 && B * (H + 8) * (W + 6) < (1LL << 40));
  CREID_CHECK_ARG(std0 != 0.f && std1 != 0.f && std2 != 0.f && (layout == 0 || layout == 1));
  AugGeom g;
  g.B = (int)B; g.H = (int)H; g.W = (int)W; g.pad = (int)pad;
  g.mean[0] = mean0; g.mean[1] = mean1; g.mean[2] = mean2;
  g.stdv[0] = std0; g.stdv[1] = std1; g.stdv[2] = std2;
  g.erase[0] = erase0; g.erase[1] = erase1; g.erase[2] = erase2;
  hipStream_t s = as_stream(stream);
  const int64_t total = layout == 0 ? B * H * W : B * (H + 8) * (W + 6);
  const unsigned blocks = (unsigned)((total + 255) / 256 < 16384 ? (total + 255) / 256 : 16384);
  if (layout == 0) {
    if (dtype != CREID_F32) return CREID_E_DTYPE;                    // the reference tensor is fp32
    hipLaunchKernelGGL(augment_nchw_kernel, dim3(blocks), dim3(256), 0, s, g, src_hwc, params, (float*)out);
  } else if (dtype == CREID_F32) {
    hipLaunchKernelGGL(augment_stem_kernel<float>, dim3(blocks), dim3(256), 0, s, g, src_hwc, params, (float*)out);
  } else if (dtype == CREID_BF16) {
    hipLaunchKernelGGL(augment_stem_kernel<unsigned short>, dim3(blocks), dim3(256), 0, s, g, src_hwc, params,
                       (unsigned short*)out);
  } else if (dtype == CREID_F16) {
    hipLaunchKernelGGL(augment_stem_kernel<_Float16>, dim3(blocks), dim3(256), 0, s, g, src_hwc, params, (_Float16*)out);
  } else {
    return CREID_E_DTYPE;
  }
  CREID_LAUNCH_RET();
}
