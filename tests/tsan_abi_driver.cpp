// Test driver (CPU only) for tests/test_host_logic.py::test_abi_host_code_is_thread_safe_under_tsan: linked with a
// ThreadSanitizer build of the HOST side of srx_api.hip.  The kernel launchers of the other translation units are
// stubbed: this exercises the ABI's argument checking, planner, knob caches and error reporting from two threads.
#include <stdio.h>
#include <string.h>
#include <atomic>
#include <thread>

#include "../include/srx.h"
#include "../ml_super_resolution_amd/csrc/elementwise.h"
#include "../ml_super_resolution_amd/csrc/launchers.h"

namespace srx {
#define STUB_CONV(name) \
    bool name(const ConvKey&, const ConvArgs&, int, size_t, hipStream_t, hipError_t* err) { *err = hipSuccess; return true; }
STUB_CONV(launch_conv_k3c64) STUB_CONV(launch_conv_k3c32) STUB_CONV(launch_conv_c4) STUB_CONV(launch_conv_misc)
STUB_CONV(launch_pipe_k3c64) STUB_CONV(launch_pipe_other) STUB_CONV(launch_pipe_strip) STUB_CONV(launch_conv_generic)
#define STUB_WGRAD(name) \
    bool name(const ConvKey&, const WgradArgs&, int, size_t, hipStream_t, hipError_t* err) { *err = hipSuccess; return true; }
STUB_WGRAD(launch_wgrad) STUB_WGRAD(launch_wgrad_lin) STUB_WGRAD(launch_wgrad_lin_strip) STUB_WGRAD(launch_wgrad_lin_pack3) STUB_WGRAD(launch_wgrad_generic) STUB_WGRAD(launch_wgrad_pipe) STUB_WGRAD(launch_wgrad_rows_full)
bool launch_wgrad_narrow(const ConvKey&, const WgradArgs&, int, hipStream_t, hipError_t* err) { *err = hipSuccess; return false; }
bool launch_conv_narrow(const ConvKey&, const ConvArgs&, hipStream_t, hipError_t* err) { *err = hipSuccess; return false; }
bool launch_conv_rows3x3(const ConvKey&, const ConvArgs&, long, hipStream_t, hipError_t* err) { *err = hipSuccess; return false; }
bool launch_conv_1x1(const ConvKey&, const ConvArgs&, long, hipStream_t, hipError_t* err) { *err = hipSuccess; return false; }
bool launch_conv_pack3(const ConvKey&, const ConvArgs&, long, hipStream_t, hipError_t* err) { *err = hipSuccess; return false; }
bool launch_conv_kwrows(const ConvKey&, const ConvArgs&, long, hipStream_t, hipError_t* err) { *err = hipSuccess; return false; }
hipError_t launch_reduce_partials(const float*, int, int, int, int, float*, float*, const float*, float, hipStream_t) { return hipSuccess; }
hipError_t launch_reduce_partials_pairs(const float*, int, int, int, int, float*, float*, int, int, hipStream_t) { return hipSuccess; }
bool launch_wgrad_kwcols(const ConvKey&, const WgradArgs&, int, long, int*, hipStream_t, hipError_t* e) { *e = hipSuccess; return false; }
bool launch_wgrad_1x1(const ConvKey&, const WgradArgs&, int, int*, hipStream_t, hipError_t* e) { *e = hipSuccess; return false; }
bool launch_wgrad_rows_strip(const ConvKey&, const WgradArgs&, int, size_t, bool, hipStream_t, hipError_t* e) { *e = hipSuccess; return true; }
bool launch_wgrad_lin_pairs(const ConvKey&, const WgradPairs&, int, int, bool, size_t, hipStream_t, hipError_t* e) { *e = hipSuccess; return true; }
hipError_t launch_subpixel(const float*, float*, int, int, int, int, int, bool, const SubpixelTune&, hipStream_t) { return hipSuccess; }
hipError_t launch_stream_copy(const float*, float*, size_t, hipStream_t) { return hipSuccess; }
hipError_t launch_mse(const float*, const float*, size_t, float, float*, int, float*, float*, hipStream_t) { return hipSuccess; }
hipError_t launch_l2(const float*, const float*, size_t, float, float*, int, float*, hipStream_t) { return hipSuccess; }
hipError_t launch_adam(float*, const float*, float*, float*, size_t, float, float, float, float, float, hipStream_t) { return hipSuccess; }
hipError_t launch_adam_dev(float*, const float*, float*, float*, size_t, void*, float, float, float, float, hipStream_t) { return hipSuccess; }
hipError_t launch_momentum(float*, const float*, float*, size_t, float, float, float, float, hipStream_t) { return hipSuccess; }
hipError_t launch_rownorm_loss(const float*, const float*, size_t, size_t, float*, float*, float*, hipStream_t) { return hipSuccess; }
size_t ssim_scratch_bytes(int N) { return (size_t)N * 64; }
hipError_t launch_ssim(const float*, const float*, float*, int, int, int, int, float, float*, hipStream_t) { return hipSuccess; }
hipError_t launch_u8_to_float(const uint8_t*, float*, size_t, hipStream_t) { return hipSuccess; }
hipError_t launch_gaussian_blur(const float*, float*, float*, int, int, int, int, float, hipStream_t) { return hipSuccess; }
hipError_t launch_resize_bilinear(const float*, float*, int, int, int, int, int, int, hipStream_t) { return hipSuccess; }
hipError_t launch_act_bwd(const float*, const float*, float*, size_t, int, hipStream_t) { return hipSuccess; }
hipError_t launch_affine(const float*, float*, size_t, float, float, hipStream_t) { return hipSuccess; }
hipError_t launch_saturate_u8(const float*, uint8_t*, size_t, hipStream_t) { return hipSuccess; }
hipError_t launch_psnr(const float*, const float*, float*, int, size_t, float, hipStream_t) { return hipSuccess; }
hipError_t launch_upsample_nearest(const float*, float*, int, int, int, int, int, hipStream_t) { return hipSuccess; }
hipError_t launch_upsample_nearest_bwd(const float*, float*, int, int, int, int, int, hipStream_t) { return hipSuccess; }
hipError_t launch_add_relu_grad(const float*, const float*, const float*, float*, size_t, hipStream_t) { return hipSuccess; }
hipError_t launch_poison_lds(hipStream_t) { return hipSuccess; }
}  // namespace srx

static std::atomic<int> failures{0};

static void worker(int id) {
    alignas(16) static float buf[2][64];
    for (int it = 0; it < 20000; ++it) {
        srx_conv_desc d;
        memset(&d, 0, sizeof(d));
        d.N = 1 + (it % 7) + id; d.H = 17 + (it % 5); d.W = 41; d.Cin = 64; d.Cout = 64; d.KH = 3; d.KW = 3; d.stride = 1;
        const size_t a = srx_conv2d_workspace_bytes(&d, SRX_OP_BWD_FILTER);
        const size_t b = srx_conv2d_workspace_bytes(&d, SRX_OP_BWD_FILTER);
        if (a == 0 || a != b) failures++;
        if ((it & 15) == 0) srx_set_conv_path(it & 16 ? 1 : 0);
        // an argument error: the text is thread-local and names THIS thread's stride (1 and 2 are implemented)
        d.stride = 3 + id;
        if (srx_conv2d_fwd(&d, buf[id], buf[id], nullptr, nullptr, buf[id], nullptr, 0, nullptr) != SRX_ERR_UNSUPPORTED) failures++;
        char want[32];
        snprintf(want, sizeof(want), "stride %d:", 3 + id);
        if (!strstr(srx_last_error(), want)) failures++;
        // a planned (stubbed) launch
        d.stride = 1;
        if (srx_conv2d_fwd(&d, buf[id], buf[id], nullptr, nullptr, buf[id], nullptr, 0, nullptr) != SRX_OK) failures++;
    }
}

int main() {
    std::thread t0(worker, 0), t1(worker, 1);
    t0.join();
    t1.join();
    if (failures.load()) { printf("tsan driver: %d wrong results\n", failures.load()); return 1; }
    printf("tsan driver ok\n");
    return 0;
}
