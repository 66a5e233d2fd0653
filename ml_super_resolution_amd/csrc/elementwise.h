// elementwise.h -- internal launch prototypes of elementwise.hip
#pragma once
#include "conv_kernels.hip.h"

namespace srx {
constexpr int kReduceBlocks = 1024;  // partial slots a reduction may use (srx_reduce_scratch_bytes)

// chunk_kb: a chunk grows while twice its size stays within this many KiB; db: two LDS buffers (software pipeline);
// grid: cap of the persistent grid; depth: chunks of loads in flight per workgroup (0 = automatic);
// throttle: requests a wave keeps in flight (0 = counted waits only, < 0 = the launcher's choice); even: 0 = subpixel_pipe_kernel's chunks of 4 blocks,
// 1 = subpixel_even_kernel (chunks of any whole number of blocks) where a workgroup needs at most two trips,
// 1 + t = that kernel with at least t trips, whatever their number
struct SubpixelTune { int chunk_kb, db, grid, depth, throttle, even; };
hipError_t launch_subpixel(const float* in, float* out, int N, int H, int W, int C, int r, bool inverse,
                           const SubpixelTune& tune, hipStream_t s);
hipError_t launch_stream_copy(const float* in, float* out, size_t bytes, hipStream_t s);
hipError_t launch_mse(const float* pred, const float* target, size_t n, float inv, float* loss, int accumulate,
                      float* dpred, float* scratch, hipStream_t s);
hipError_t launch_l2(const float* w, const float* mask, size_t n, float scale, float* loss, int accumulate, float* scratch,
                     hipStream_t s);
hipError_t launch_adam(float* w, const float* g, float* m, float* v, size_t n, float lr_t, float b1, float b2, float eps,
                       float gs, hipStream_t s);
hipError_t launch_adam_dev(float* w, const float* g, float* m, float* v, size_t n, void* state, float b1, float b2, float eps,
                           float gs, hipStream_t s);
hipError_t launch_momentum(float* w, const float* g, float* acc, size_t n, float lr, float mom, float cap, float gs,
                           hipStream_t s);
hipError_t launch_rownorm_loss(const float* pred, const float* target, size_t rows, size_t row_len, float* loss,
                               float* dpred, float* norms, hipStream_t s);
size_t ssim_scratch_bytes(int N);
hipError_t launch_ssim(const float* a, const float* b, float* out, int N, int H, int W, int C, float max_val, float* scratch,
                       hipStream_t s);
hipError_t launch_u8_to_float(const uint8_t* in, float* out, size_t n, hipStream_t s);
hipError_t launch_gaussian_blur(const float* in, float* out, float* tmp, int N, int H, int W, int C, float sigma, hipStream_t s);
hipError_t launch_resize_bilinear(const float* in, float* out, int N, int H, int W, int C, int OH, int OW, hipStream_t s);
hipError_t launch_act_bwd(const float* dy, const float* y, float* dpre, size_t n, int act, hipStream_t s);
hipError_t launch_affine(const float* x, float* out, size_t n, float a, float b, hipStream_t s);
hipError_t launch_saturate_u8(const float* x, uint8_t* out, size_t n, hipStream_t s);
hipError_t launch_psnr(const float* a, const float* b, float* out, int N, size_t per_image, float max_val, hipStream_t s);
hipError_t launch_upsample_nearest(const float* in, float* out, int N, int H, int W, int C, int f, hipStream_t s);
hipError_t launch_upsample_nearest_bwd(const float* dout, float* din, int N, int H, int W, int C, int f, hipStream_t s);
hipError_t launch_add_relu_grad(const float* a, const float* b, const float* y, float* out, size_t n, hipStream_t s);
hipError_t launch_poison_lds(hipStream_t s);
}  // namespace srx
