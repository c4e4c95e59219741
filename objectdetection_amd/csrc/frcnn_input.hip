// Input side of the path: ragged ground truth -> the zero-padded [B,G,4] / [B,G] tensors the
// assignment kernels expect, with the horizontal-flip augmentation of the boxes, and the
// matching left-right flip of the image batch.  Reference: Faster_RCNN/module/data.py:9-52
// (data_preprocess, augmentation_data, random_augmentation), shipdetection/train.py:28-45.
// gfx950, wave64.
#include "frcnn_common.h"

namespace frcnn {

// flip decision of image b: explicit flag, or Philox counter key (stream 6) above 2^31,
// i.e. uniform > 0.5 (module/data.py:44-45)
__device__ __forceinline__ bool flip_of(const uint8_t* flip, int b, uint64_t seed, int image_base) {
    if (flip) return flip[b] != 0;
    return counter_key(0u, 6u, (uint32_t)(image_base + b), seed) > 0x80000000u;
}

// one thread per (image, slot)
__global__ void __launch_bounds__(256)
pad_gt_kernel(const float4* __restrict__ boxes, const int64_t* __restrict__ labels,
              const int32_t* __restrict__ offsets, const uint8_t* __restrict__ flip, int use_flip,
              uint64_t seed, int image_base, int B, int G, float4* __restrict__ gt,
              int64_t* __restrict__ gt_label, int32_t* __restrict__ count) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= B * G) return;
    const int b = t / G, g = t % G;
    const int lo = offsets[b], n = offsets[b + 1] - lo;
    if (g == 0 && count) count[b] = n;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    int64_t l = 0;
    if (g < n) {
        v = boxes[lo + g];
        if (labels) l = labels[lo + g];
        if (use_flip && flip_of(flip, b, seed, image_base)) {
            const float xmin = 1.0f - v.w, xmax = 1.0f - v.y;     // module/data.py:28-29
            v.y = xmin; v.w = xmax;
        }
    }
    gt[t] = v;
    if (gt_label) gt_label[t] = l;
}

// images [B,H,W,C] f32 contiguous; one thread per 4-byte element, a row at a time so that
// reads and writes of a wave stay contiguous (W*C elements per row)
__global__ void __launch_bounds__(256)
flip_images_kernel(const float* __restrict__ src, const uint8_t* __restrict__ flip, uint64_t seed,
                   int image_base, int H, int W, int C, float* __restrict__ dst) {
    const int b = blockIdx.z, y = blockIdx.y;
    const bool f = flip_of(flip, b, seed, image_base);
    const size_t row = ((size_t)b * H + y) * (size_t)W * C;
    const int n = W * C;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const int x = i / C, c = i - x * C;
        const int sx = f ? W - 1 - x : x;
        dst[row + i] = src[row + (size_t)sx * C + c];
    }
}

// the same with 16-byte stores (and 16-byte loads for images that are not flipped): one thread
// per four consecutive elements of a row; needs W*C % 4 == 0 and 16-byte aligned tensors
__global__ void __launch_bounds__(256)
flip_images_vec4_kernel(const float* __restrict__ src, const uint8_t* __restrict__ flip,
                        uint64_t seed, int image_base, int H, int W, int C,
                        float* __restrict__ dst) {
    const int b = blockIdx.z, y = blockIdx.y;
    const bool f = flip_of(flip, b, seed, image_base);
    const size_t row = ((size_t)b * H + y) * (size_t)W * C;
    const int n4 = (W * C) >> 2;
    const float4* s4 = reinterpret_cast<const float4*>(src + row);
    float4* d4 = reinterpret_cast<float4*>(dst + row);
    for (int q = blockIdx.x * blockDim.x + threadIdx.x; q < n4; q += gridDim.x * blockDim.x) {
        if (!f) { d4[q] = s4[q]; continue; }
        const int i = q * 4;
        int x = i / C, c = i - x * C;
        float v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            v[u] = src[row + (size_t)(W - 1 - x) * C + c];
            if (++c == C) { c = 0; ++x; }
        }
        d4[q] = make_float4(v[0], v[1], v[2], v[3]);
    }
}

}  // namespace frcnn

using namespace frcnn;

extern "C" int frcnn_pad_gt(const float* boxes, const int64_t* labels, const int32_t* offsets,
                            const uint8_t* flip, int use_flip, uint64_t seed, int image_base,
                            int B, int G, float* gt, int64_t* gt_label, int32_t* count,
                            frcnn_stream_t stream) {
    if (!boxes || !offsets || !gt || B <= 0 || G <= 0) return FRCNN_E_BADARG;
    if ((reinterpret_cast<uintptr_t>(boxes) | reinterpret_cast<uintptr_t>(gt)) & 15u)
        return FRCNN_E_BADARG;
    if ((long long)B * G > 0x7FFFFFFFll) return FRCNN_E_UNSUPPORTED;
    const int total = B * G;
    FRCNN_LAUNCH(pad_gt_kernel, dim3((total + 255) / 256), dim3(256), 0, (hipStream_t)stream,
                 reinterpret_cast<const float4*>(boxes), labels, offsets, flip, use_flip, seed,
                 image_base, B, G, reinterpret_cast<float4*>(gt), gt_label, count);
    return FRCNN_OK;
}

extern "C" int frcnn_flip_images(const float* images, const uint8_t* flip, uint64_t seed,
                                 int image_base, int B, int H, int W, int C, float* out,
                                 frcnn_stream_t stream) {
    if (!images || !out || images == out || B <= 0 || H <= 0 || W <= 0 || C <= 0)
        return FRCNN_E_BADARG;
    if (B > 65535 || H > 65535 || (long long)W * C > 0x7FFFFFFFll) return FRCNN_E_UNSUPPORTED;
    const int n = W * C;
    const bool vec4 = (n % 4) == 0 &&
                      ((reinterpret_cast<uintptr_t>(images) | reinterpret_cast<uintptr_t>(out)) & 15u) == 0;
    if (vec4) {
        const int bx = (n / 4 + 255) / 256 < 4 ? (n / 4 + 255) / 256 : 4;
        FRCNN_LAUNCH(flip_images_vec4_kernel, dim3(bx, H, B), dim3(256), 0, (hipStream_t)stream,
                     images, flip, seed, image_base, H, W, C, out);
    } else {
        const int bx = n >= 1024 ? 4 : 1;
        FRCNN_LAUNCH(flip_images_kernel, dim3(bx, H, B), dim3(256), 0, (hipStream_t)stream, images,
                     flip, seed, image_base, H, W, C, out);
    }
    return FRCNN_OK;
}
