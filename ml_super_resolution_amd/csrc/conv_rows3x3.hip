// conv_rows3x3.hip -- forward 3x3 convolution from 32 input channels into 17..32 output channels on large inputs: ESPCN's f3
// (32 -> 3 r^2, stored through the sub-pixel map) on whole images (espcn/espcn/model_espcn.py:127-134,
// espcn/espcn/experiment_test.py:148-184).  The pipelined family (conv_pipe_kernel) has no column-strip instance for two output
// chunks and no ragged-channel store, so on images too wide for full-width tiles the layer ran conv_mfma_kernel (two
// workgroups per CU that stage their tile between MFMA phases with per-lane integer code): 53 % of the fp32 peak at 720 x 1280.
// This is conv_pack3_kernel's structure: ONE workgroup of 8 waves per CU, a tile of 8 output rows x 48 output columns whose
// input halo is fetched into registers while the previous tile is computed, weights stationary in accumulation registers,
// MFMAs pinned in blocks of (3 sub-tiles x 4 k-steps) with the LDS fragments of the next block issued first, bias as the
// accumulators' initial value, activation and stores per row.  wave = (16-channel chunk, row phase).  Same products in the same
// order as conv_mfma_kernel (one chain over taps and channel groups per output): BIT-IDENTICAL to it -- tested.
// Measured (ESPCN f3 at 720 x 1280 LR): 172.9 -> 157.4 us; slower than conv_mfma_kernel below ~150,000 pixels (256^2: 20.3 vs
// 18.5 us), hence the route's threshold.  The 64-input-channel instance (ESPCN's f2; 144 weight registers: one wave per SIMD,
// nothing to overlap its epilogue and LDS latency with) was built too and lost to conv_mfma_kernel (334 vs 318 us with tanh, 304
// vs 304 with ReLU): not instantiated.
#include "launchers.h"
namespace srx {
namespace {

constexpr int kRTH = 8, kRTW = 48;
// waves per workgroup: 8 (two per SIMD) where a wave's registers allow it -- 32 input channels: 72 weights --, 4 for 64 input
// channels (144 weights in accumulation registers + 128 registers of prefetched tile: one wave per SIMD owns all 512)
template <int CIN> struct RowsWaves { static constexpr int value = CIN >= 64 ? 4 : 8; };

// 12 MFMAs of one (tap, 16-channel group): three accumulators x four k-steps, weights from accumulation registers
__device__ __forceinline__ void mfma12_a(f32x4 (&c)[3], float w0, float w1, float w2, float w3, const f32x4 b0, const f32x4 b1, const f32x4 b2) {
    asm volatile("s_nop 1\n\t"
                 "v_mfma_f32_16x16x4_f32 %0, %3, %7, %0\n\t"  "v_mfma_f32_16x16x4_f32 %1, %3, %11, %1\n\t" "v_mfma_f32_16x16x4_f32 %2, %3, %15, %2\n\t"
                 "v_mfma_f32_16x16x4_f32 %0, %4, %8, %0\n\t"  "v_mfma_f32_16x16x4_f32 %1, %4, %12, %1\n\t" "v_mfma_f32_16x16x4_f32 %2, %4, %16, %2\n\t"
                 "v_mfma_f32_16x16x4_f32 %0, %5, %9, %0\n\t"  "v_mfma_f32_16x16x4_f32 %1, %5, %13, %1\n\t" "v_mfma_f32_16x16x4_f32 %2, %5, %17, %2\n\t"
                 "v_mfma_f32_16x16x4_f32 %0, %6, %10, %0\n\t" "v_mfma_f32_16x16x4_f32 %1, %6, %14, %1\n\t" "v_mfma_f32_16x16x4_f32 %2, %6, %18, %2"
                 : "+v"(c[0]), "+v"(c[1]), "+v"(c[2])
                 : "a"(w0), "a"(w1), "a"(w2), "a"(w3),
                   "v"(b0[0]), "v"(b0[1]), "v"(b0[2]), "v"(b0[3]), "v"(b1[0]), "v"(b1[1]), "v"(b1[2]), "v"(b1[3]),
                   "v"(b2[0]), "v"(b2[1]), "v"(b2[2]), "v"(b2[3]));
}

template <int CIN>
__global__ __launch_bounds__(64 * RowsWaves<CIN>::value, 1) void conv_rows3x3_kernel(const ConvArgs a, int units_total) {
    constexpr int kRNW = RowsWaves<CIN>::value;
    constexpr int NPH = kRNW / 2;               // row phases per chunk
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int PS = Lds<CIN>::PS;
    constexpr int NG = CIN / 16;
    constexpr int NBLK = 9 * NG;
    constexpr int RSW = kRTW + 2;
    constexpr int TPP = CIN / 4, PPP = 64 * kRNW / TPP;
    constexpr int NSLOT = (kRTH + 2) * RSW;
    constexpr int NPASS = (NSLOT + PPP - 1) / PPP;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, kq = lane >> 4;
    const int chunk = wave & 1, phase = wave >> 1;
    const int co = 16 * chunk + li;

    // stationary weights of the wave's 16 output channels: wr[tap][g][s] = w[tap][16 g + 4 kq + s][co], kept in accumulation registers
    float wr[NBLK * 4];
#pragma unroll
    for (int t = 0; t < NBLK; ++t)
#pragma unroll
        for (int sI = 0; sI < 4; ++sI) {
            const int tap = t / NG, g = t % NG;
            const int ci = 16 * g + 4 * kq + sI;
            const bool ok = co < a.Cout;
            const float v = a.w[ok ? ((size_t)tap * CIN + ci) * a.Cout + co : 0];
            wr[t * 4 + sI] = ok ? v : 0.0f;
        }
#pragma unroll
    for (int i = 0; i < NBLK * 4; ++i) {
        const float t = wr[i];
        asm volatile("v_accvgpr_write_b32 %0, %1" : "=a"(wr[i]) : "v"(t));
    }
    const int cb = 16 * chunk + 4 * kq;             // D layout: the lane's four output channels
    f32x4 bias4 = {0.f, 0.f, 0.f, 0.f};
    if (a.bias) {
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (cb + e < a.Cout) bias4[e] = a.bias[cb + e];
    }
    const float slope = act_slope(a.act);
    const bool vec = ((a.Cout & 3) == 0) && (cb + 3 < a.Cout) && !a.d2s_r;
    // sub-pixel store mode (see conv_epilogue): channel ch of an LR pixel goes to HR row + ch / rc, element ch % rc of its segment
    int eo[4] = {0, 1, 2, 3};
    if (a.d2s_r) {
        const int hr_row = a.OW * a.d2s_rc;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int ch = cb + e, dy = ch / a.d2s_rc;
            eo[e] = dy * hr_row + (ch - dy * a.d2s_rc);
        }
    }

    const int u0 = (int)(((long)blockIdx.x * units_total) / gridDim.x), u1 = (int)(((long)(blockIdx.x + 1) * units_total) / gridDim.x);
    const int c4 = tid % TPP, sp = tid / TPP;
    f32x4 pre[NPASS];
    auto tile_of = [&](int u, int& n, int& h0, int& ow0, int& th, int& tw) {
        h0 = u % a.OH;
        const int t2 = u / a.OH;
        const int tx = t2 % a.NTX;
        n = t2 / a.NTX;
        ow0 = tx * kRTW;
        th = a.OH - h0 < kRTH ? a.OH - h0 : kRTH;
        if (u1 - u < th) th = u1 - u;
        tw = a.OW - ow0 < kRTW ? a.OW - ow0 : kRTW;
    };
    // the NEXT tile's input halo into registers (slot s <-> (row s / RSW, column s % RSW); out-of-image slots: out-of-range offset -> 0)
    auto fetch = [&](int u) {
        int n, h0, ow0, th, tw;
        tile_of(u, n, h0, ow0, th, tw);
        const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float*>(a.x) + (size_t)n * a.H * a.W * CIN, 0, a.H * a.W * CIN * 4, 0x00020000);
        const int hin = h0 - a.pad_t, win = ow0 - a.pad_l;
#pragma unroll
        for (int j = 0; j < NPASS; ++j) {
            const int sl = sp + j * PPP;
            const int r = sl / RSW, c = sl - r * RSW;
            const int ih = hin + r, iw = win + c;
            const bool ok = ((unsigned)ih < (unsigned)a.H) & ((unsigned)iw < (unsigned)a.W) & (r < th + 2);
            pre[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xrs, ok ? ((ih * a.W + iw) * CIN + 4 * c4) * 4 : kOobOffset, 0, 0));
        }
    };
    if (u0 < u1) fetch(u0);
    for (int u = u0; u < u1;) {
        int n, h0, ow0, th, tw;
        tile_of(u, n, h0, ow0, th, tw);
        const int un = u + th;
        lds_barrier();
#pragma unroll
        for (int j = 0; j < NPASS; ++j) {
            const int sl = sp + j * PPP;
            if (sl < NSLOT) *reinterpret_cast<f32x4*>(lds + sl * PS + 4 * c4) = pre[j];
        }
        lds_barrier();
        if (un < u1) fetch(un);
        for (int r = phase; r < th; r += NPH) {
            // B fragment of sub-tile g3 (columns 16 g3 + li), block (tap, g): pixel ((r + kh) RSW + 16 g3 + li + kw), channels 16 g + 4 kq ..
            const float* px = lds + ((r * RSW + li) * PS + 4 * kq);
            f32x4 acc[3] = {bias4, bias4, bias4};
            f32x4 c0 = *reinterpret_cast<const f32x4*>(px), c1 = *reinterpret_cast<const f32x4*>(px + 16 * PS),
                  c2 = *reinterpret_cast<const f32x4*>(px + 32 * PS);
#pragma unroll
            for (int t = 0; t < NBLK; ++t) {
                f32x4 n0 = c0, n1 = c1, n2 = c2;
                if (t + 1 < NBLK) {
                    const int tap1 = (t + 1) / NG, g1 = (t + 1) % NG;
                    const float* q = px + ((tap1 / 3) * RSW + (tap1 % 3)) * PS + 16 * g1;
                    n0 = *reinterpret_cast<const f32x4*>(q); n1 = *reinterpret_cast<const f32x4*>(q + 16 * PS); n2 = *reinterpret_cast<const f32x4*>(q + 32 * PS);
                }
                mfma12_a(acc, wr[t * 4], wr[t * 4 + 1], wr[t * 4 + 2], wr[t * 4 + 3], c0, c1, c2);
                c0 = n0; c1 = n1; c2 = n2;
            }
            // MFMA results are read by VALU code next: software covers the result latency (the MFMAs above are asm)
            asm volatile("s_nop 15\n\ts_nop 7" : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]));
            if (cb < a.Cout) {
#pragma unroll
                for (int g3 = 0; g3 < 3; ++g3) {
                    const int c = 16 * g3 + li;
                    if (c >= tw) continue;
                    const f32x4 v = act_apply4(acc[g3], a.act, slope);
                    if (vec) {
                        *reinterpret_cast<f32x4*>(a.y + ((((size_t)n * a.OH + h0 + r) * a.OW + ow0 + c) * a.Cout + cb)) = v;
                    } else if (a.d2s_r) {
                        float* yo = a.y + (size_t)n * a.OH * a.OW * a.Cout + (size_t)((h0 + r) * a.d2s_r * a.OW + ow0 + c) * a.d2s_rc;
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            if (cb + e < a.Cout) yo[eo[e]] = v[e];
                    } else {
                        float* yo = a.y + ((((size_t)n * a.OH + h0 + r) * a.OW + ow0 + c) * a.Cout + cb);
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            if (cb + e < a.Cout) yo[e] = v[e];
                    }
                }
            }
        }
        u = un;
    }
}

}  // namespace

// Returns true when this route took the launch.  min_pixels: below it the layer stays on conv_mfma_kernel.
bool launch_conv_rows3x3(const ConvKey& k, const ConvArgs& a, long min_pixels, hipStream_t s, hipError_t* err) {
    if (k.wt || a.skip || a.mask || a.stride != 1 || a.post_relu || k.kh != 3 || k.kw != 3) return false;
    if (!(a.Cin == 32 && a.Cout > 16 && a.Cout <= 32)) return false;
    if ((long)a.N * a.OH * a.OW < (min_pixels > 150000 ? min_pixels : 150000)) return false;
    if ((long)a.H * a.W * a.Cin * 4 >= (1L << 31) - 4096) return false;
    ConvArgs b = a;
    b.NTX = (a.OW + kRTW - 1) / kRTW;
    const long units = (long)a.N * b.NTX * a.OH;
    if (units >= (1L << 31)) return false;
    int cus = 256;
    {
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0) cus = n;
        else (void)hipGetLastError();
    }
    const long ranges = (units + 3) / 4;            // at least the four row phases of a chunk's waves
    const int grid = (int)(ranges < cus ? ranges : cus);
    const size_t lds = (size_t)(kRTH + 2) * (kRTW + 2) * (a.Cin + 4) * sizeof(float);
    // (> 64 KiB of dynamic LDS needs the attribute: raised once per host thread, outside any stream capture of later launches)
    static thread_local bool configured = false;
    if (!configured) {
        *err = hipFuncSetAttribute(reinterpret_cast<const void*>(conv_rows3x3_kernel<32>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (*err != hipSuccess) return true;
        configured = true;
    }
    hipLaunchKernelGGL((conv_rows3x3_kernel<32>), dim3(grid), dim3(64 * RowsWaves<32>::value), lds, s, b, (int)units);
    *err = hipGetLastError();
    return true;
}
}  // namespace srx
