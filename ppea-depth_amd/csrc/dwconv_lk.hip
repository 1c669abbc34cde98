// Large-kernel depthwise convolution for RepLKNet (31/29/27/13 + the 5x5 re-param branch),
// gfx950.  Replaces nn.Conv2d(groups=C) at networks/replknet_adapter.py:151-168, :225-239.
//
// Design (im2col-free, LDS sliding window, one *wave* per work item):
//   * a work item is one output tile TH x TW of one (n, c) plane; the 64 lanes of a wave
//     form an LY x LX grid and each lane owns an RY x RX register tile of outputs;
//   * the wave stages the tile plus its K/2 halo (zero filled outside the plane) into a
//     wave-private LDS region with coalesced NCHW row reads; no cross-wave sharing;
//   * because the whole wave works on ONE channel, the K*K filter taps are wave-uniform:
//     they are fetched with scalar loads and enter v_fma_f32 as SGPR operands, so the
//     inner loop is pure FMA: per staged LDS row a lane reads RX+K-1 floats and issues
//     up to RY*RX*K FMAs (19.6 FMA per LDS dword for the 31x31 / 3x8 configuration);
//   * the 5x5 branch re-uses the LDS tile of the large kernel (fwd), and in dgrad both
//     branches accumulate into the same registers, so x / dx cross HBM once.
// Roofline: vector-FMA bound (AI ~ 226 F/B at k=31); algorithmic bytes
// per image fp32 k31 = 2*128*48*160*4 + 128*961*4 = 8.36 MB (+3.93 MB for y_small).
#include "common.h"
#include <cstdlib>

namespace {

constexpr int round4(int v) { return (v + 3) & ~3; }
constexpr int lds_stride(int iw) { return (round4(iw) % 32 == 0) ? round4(iw) + 4 : round4(iw); }

template <int K, int RY, int RX, int LX, int LY>
struct Cfg {
    static constexpr int P = K / 2;
    static constexpr int TH = LY * RY, TW = LX * RX;
    static constexpr int IH = TH + K - 1, IW = TW + K - 1;
    static constexpr int STRIDE = lds_stride(IW);
    static constexpr int LDS_FLOATS = IH * STRIDE;
    static_assert(LX * LY <= 64, "lane grid exceeds a wave");
};

// Stage rows [y0-P, y0-P+IH) x cols [x0-P, x0-P+IW) of one plane into LDS, zero outside.
template <typename T, typename CF>
__device__ __forceinline__ void stage_tile(float* tile, const T* plane, int H, int W, int y0, int x0,
                                           int lane) {
    for (int idx = lane; idx < CF::IH * CF::IW; idx += WAVE) {
        const int r = idx / CF::IW, c = idx - r * CF::IW;
        const int gy = y0 - CF::P + r, gx = x0 - CF::P + c;
        float v = 0.f;
        if (gy >= 0 && gy < H && gx >= 0 && gx < W) v = ld_f32<T>(plane + (long)gy * W + gx);
        tile[r * CF::STRIDE + c] = v;
    }
}

template <int N>
__device__ __forceinline__ void load_seg(float (&seg)[N], const float* row, bool aligned16) {
    // row is 4-byte aligned at least; use 16-byte reads when the caller guarantees alignment.
    if (aligned16) {
        constexpr int N4 = N / 4;
#pragma unroll
        for (int i = 0; i < N4; ++i) {
            const float4 v = reinterpret_cast<const float4*>(row)[i];
            seg[4 * i] = v.x; seg[4 * i + 1] = v.y; seg[4 * i + 2] = v.z; seg[4 * i + 3] = v.w;
        }
#pragma unroll
        for (int i = 4 * N4; i < N; ++i) seg[i] = row[i];
    } else {
#pragma unroll
        for (int i = 0; i < N; ++i) seg[i] = row[i];
    }
}

// acc[oy][ox] += sum_{ky,kx} w[ky][kx] * tile[(ly*RY+oy) + OFF + ky][(lx*RX+ox) + OFF + kx]
// (FLIP reverses both filter axes: the dgrad correlation).  `wc` must be wave-uniform.
template <int KK, int OFF, bool FLIP, typename CF, int RY, int RX>
__device__ __forceinline__ void accumulate(const float* tile, const float* __restrict__ wc,
                                           float (&acc)[RY][RX], int ly, int lx) {
    constexpr int SEG = RX + KK - 1;
    constexpr bool AL = (OFF % 4 == 0) && (RX % 4 == 0);
    const float* base = tile + (ly * RY + OFF) * CF::STRIDE + lx * RX + OFF;
#pragma unroll 1
    for (int rp = 0; rp < RY + KK - 1; ++rp) {
        float seg[SEG];
        load_seg<SEG>(seg, base + rp * CF::STRIDE, AL);
#pragma unroll
        for (int oy = 0; oy < RY; ++oy) {
            const int ky = rp - oy;                       // wave-uniform
            if (ky >= 0 && ky < KK) {
                const float* wr = wc + (FLIP ? (KK - 1 - ky) : ky) * KK;
#pragma unroll
                for (int kx = 0; kx < KK; ++kx) {
                    const float wv = wr[FLIP ? (KK - 1 - kx) : kx];
#pragma unroll
                    for (int ox = 0; ox < RX; ++ox) acc[oy][ox] = fmaf(wv, seg[ox + kx], acc[oy][ox]);
                }
            }
        }
    }
}

template <typename T, int RY, int RX>
__device__ __forceinline__ void store_tile(T* plane, const float (&acc)[RY][RX], int H, int W, int y0,
                                           int x0, int ly, int lx) {
#pragma unroll
    for (int oy = 0; oy < RY; ++oy) {
        const int gy = y0 + ly * RY + oy;
        if (gy >= H) continue;
#pragma unroll
        for (int ox = 0; ox < RX; ++ox) {
            const int gx = x0 + lx * RX + ox;
            if (gx < W) st_f32<T>(plane + (long)gy * W + gx, acc[oy][ox]);
        }
    }
}

// BWD = false: a = x, out0 = y_big, out1 = y_small (KS > 0)
// BWD = true : a = dy_big, b = dy_small (KS > 0), out0 = dx
constexpr int WPB = 2;   // waves per workgroup (items are wave-private; small groups pack LDS tighter)

template <typename T, int K, int KS, int RY, int RX, int LX, int LY, bool BWD>
__global__ __launch_bounds__(64 * WPB) void dwconv_lk_kernel(const T* __restrict__ a, const T* __restrict__ b,
                                                        const float* __restrict__ w_big,
                                                        const float* __restrict__ w_small,
                                                        T* __restrict__ out0, T* __restrict__ out1,
                                                        int C, int H, int W, int tiles_x, int tiles_y,
                                                        long n_items) {
    using CF = Cfg<K, RY, RX, LX, LY>;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    float* tile = smem + wave * CF::LDS_FLOATS;

    long item = (long)blockIdx.x * WPB + wave;
    const bool active = item < n_items;
    if (!active) item = n_items - 1;                       // keep barriers uniform
    const int tiles = tiles_x * tiles_y;
    const long plane_id = item / tiles;
    const int t = (int)(item - plane_id * tiles);
    const int ty = t / tiles_x, tx = t - ty * tiles_x;
    const int c = (int)(plane_id % C);
    const int y0 = ty * CF::TH, x0 = tx * CF::TW;
    const long plane_off = plane_id * (long)H * W;
    const bool lane_ok = active && lane < LX * LY;
    const int ll = lane < LX * LY ? lane : 0;
    const int ly = ll / LX, lx = ll - ly * LX;

    const float* wb = w_big + (long)c * K * K;
    const float* ws = (KS > 0) ? w_small + (long)c * KS * KS : nullptr;

    float acc[RY][RX];
#pragma unroll
    for (int i = 0; i < RY; ++i)
#pragma unroll
        for (int j = 0; j < RX; ++j) acc[i][j] = 0.f;

    stage_tile<T, CF>(tile, a + plane_off, H, W, y0, x0, lane);
    __syncthreads();
    accumulate<K, 0, BWD, CF, RY, RX>(tile, wb, acc, ly, lx);

    if constexpr (!BWD) {
        if (lane_ok) store_tile<T, RY, RX>(out0 + plane_off, acc, H, W, y0, x0, ly, lx);
        if constexpr (KS > 0) {
            if (out1 != nullptr) {                          // uniform
#pragma unroll
                for (int i = 0; i < RY; ++i)
#pragma unroll
                    for (int j = 0; j < RX; ++j) acc[i][j] = 0.f;
                accumulate<KS, (K - KS) / 2, false, CF, RY, RX>(tile, ws, acc, ly, lx);
                if (lane_ok) store_tile<T, RY, RX>(out1 + plane_off, acc, H, W, y0, x0, ly, lx);
            }
        }
    } else {
        if constexpr (KS > 0) {
            if (b != nullptr) {                             // uniform
                __syncthreads();
                stage_tile<T, CF>(tile, b + plane_off, H, W, y0, x0, lane);
                __syncthreads();
                accumulate<KS, (K - KS) / 2, true, CF, RY, RX>(tile, ws, acc, ly, lx);
            }
        }
        if (lane_ok) store_tile<T, RY, RX>(out0 + plane_off, acc, H, W, y0, x0, ly, lx);
    }
}

// Any odd K (slow path for kernel sizes without a tuned instantiation): one thread per output.
template <typename T, bool FLIP>
__global__ void dwconv_generic_kernel(const T* __restrict__ x, const float* __restrict__ w,
                                      T* __restrict__ y, int C, int H, int W, int K, long total,
                                      int accumulate_into) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int gx = (int)(i % W);
    const int gy = (int)((i / W) % H);
    const long plane = i / ((long)H * W);
    const int c = (int)(plane % C);
    const int P = K / 2;
    const T* xp = x + plane * (long)H * W;
    const float* wc = w + (long)c * K * K;
    float acc = 0.f;
    for (int ky = 0; ky < K; ++ky) {
        const int sy = gy + ky - P;
        if (sy < 0 || sy >= H) continue;
        for (int kx = 0; kx < K; ++kx) {
            const int sx = gx + kx - P;
            if (sx < 0 || sx >= W) continue;
            const float wv = FLIP ? wc[(K - 1 - ky) * K + (K - 1 - kx)] : wc[ky * K + kx];
            acc = fmaf(wv, ld_f32<T>(xp + (long)sy * W + sx), acc);
        }
    }
    if (accumulate_into) acc += ld_f32<T>(y + i);
    st_f32<T>(y + i, acc);
}

// wgrad: block = (c, ky); thread t -> kx = t & 31, row slice = t >> 5 (8 slices).
__global__ __launch_bounds__(256) void dwconv_wgrad_kernel(const float* __restrict__ x,
                                                           const float* __restrict__ dy,
                                                           float* __restrict__ dw, int N, int C, int H,
                                                           int W, int K) {
    __shared__ float red[8][32];
    const int c = blockIdx.x, ky = blockIdx.y;
    const int kx = threadIdx.x & 31, sl = threadIdx.x >> 5;
    const int P = K / 2;
    float acc = 0.f;
    if (kx < K) {
        for (int rowi = sl; rowi < N * H; rowi += 8) {
            const int n = rowi / H, i = rowi - n * H;
            const int sy = i + ky - P;
            if (sy < 0 || sy >= H) continue;
            const float* dyr = dy + ((long)(n * C + c) * H + i) * W;
            const float* xr = x + ((long)(n * C + c) * H + sy) * W;
            const int j0 = max(0, P - kx), j1 = min(W, W + P - kx);
            for (int j = j0; j < j1; ++j) acc = fmaf(dyr[j], xr[j + kx - P], acc);
        }
    }
    red[sl][kx] = acc;
    __syncthreads();
    if (sl == 0 && kx < K) {
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) s += red[i][kx];
        dw[((long)c * K + ky) * K + kx] = s;
    }
}

struct Shape { int N, C, H, W; };

template <typename T, int K, int KS, int RY, int RX, int LX, int LY, bool BWD>
int launch_cfg(const T* a, const T* b, const float* wb, const float* ws, T* o0, T* o1, Shape s,
               hipStream_t st) {
    using CF = Cfg<K, RY, RX, LX, LY>;
    const int tiles_x = (s.W + CF::TW - 1) / CF::TW, tiles_y = (s.H + CF::TH - 1) / CF::TH;
    const long n_items = (long)s.N * s.C * tiles_x * tiles_y;
    if (n_items == 0) return 0;
    const long blocks = (n_items + WPB - 1) / WPB;
    const size_t lds = (size_t)WPB * CF::LDS_FLOATS * sizeof(float);
    static_assert(WPB * CF::LDS_FLOATS * sizeof(float) <= 64 * 1024, "LDS tile too large");
    auto kern = dwconv_lk_kernel<T, K, KS, RY, RX, LX, LY, BWD>;
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(64 * WPB), lds, st, a, b, wb, ws, o0, o1, s.C, s.H,
                       s.W, tiles_x, tiles_y, n_items);
    return launch_status();
}

// relative cost of covering H x W with TH x TW tiles at RY x RX outputs per lane
inline long cover_cost(int H, int W, int RY, int RX, int LX, int LY) {
    const long ty = (H + LY * RY - 1) / (LY * RY), tx = (W + LX * RX - 1) / (LX * RX);
    return ty * tx * RY * RX;
}

#define PPEA_TRY(K_, RY_, RX_, LX_, LY_)                                                        \
    {                                                                                           \
        const long cst = cover_cost(s.H, s.W, RY_, RX_, LX_, LY_);                              \
        if (best < 0 || cst < best) { best = cst; pick = idx; }                                 \
        ++idx;                                                                                  \
    }
#define PPEA_RUN(K_, RY_, RX_, LX_, LY_)                                                        \
    if (pick == idx++)                                                                          \
        return KS5 ? launch_cfg<T, K_, 5, RY_, RX_, LX_, LY_, BWD>(a, b, wb, ws, o0, o1, s, st) \
                   : launch_cfg<T, K_, 0, RY_, RX_, LX_, LY_, BWD>(a, b, wb, ws, o0, o1, s, st);

// (K, RY, RX, LX, LY): tile = (LY*RY) x (LX*RX); every tile+halo fits <= ~18 KB of LDS per wave so
// that >= 8 waves (2 per SIMD) are resident per CU -- one wave alone issues v_fma at half rate.
#define CFGS_31(X) X(31, 3, 4, 8, 8) X(31, 2, 8, 8, 8)
#define CFGS_29(X) X(29, 3, 5, 8, 8) X(29, 3, 4, 8, 8)
#define CFGS_27(X) X(27, 1, 8, 5, 12) X(27, 1, 8, 4, 12) X(27, 2, 8, 8, 8)
#define CFGS_13(X) X(13, 1, 4, 5, 6) X(13, 1, 4, 4, 6) X(13, 2, 4, 8, 8)

template <typename T, bool BWD>
int dispatch(const T* a, const T* b, const float* wb, const float* ws, T* o0, T* o1, Shape s, int K,
             int KS, hipStream_t st) {
    const bool KS5 = (KS == 5);
    long best = -1;
    int pick = -1, idx = 0;
    switch (K) {
        case 31: { CFGS_31(PPEA_TRY) idx = 0; CFGS_31(PPEA_RUN) break; }
        case 29: { CFGS_29(PPEA_TRY) idx = 0; CFGS_29(PPEA_RUN) break; }
        case 27: { CFGS_27(PPEA_TRY) idx = 0; CFGS_27(PPEA_RUN) break; }
        case 13: { CFGS_13(PPEA_TRY) idx = 0; CFGS_13(PPEA_RUN) break; }
        default: break;
    }
    return PPEA_ERR_UNSUPPORTED;
}

template <typename T>
int run_generic(const T* x, const float* w, T* y, Shape s, int K, bool flip, int accumulate_into,
                hipStream_t st) {
    const long total = (long)s.N * s.C * s.H * s.W;
    if (total == 0) return 0;
    const int bs = 256;
    const long blocks = (total + bs - 1) / bs;
    if (flip)
        hipLaunchKernelGGL((dwconv_generic_kernel<T, true>), dim3((unsigned)blocks), dim3(bs), 0, st, x, w,
                           y, s.C, s.H, s.W, K, total, accumulate_into);
    else
        hipLaunchKernelGGL((dwconv_generic_kernel<T, false>), dim3((unsigned)blocks), dim3(bs), 0, st, x, w,
                           y, s.C, s.H, s.W, K, total, accumulate_into);
    return launch_status();
}

inline bool bad_shape(int N, int C, int H, int W, int K, int KS) {
    return N < 0 || C <= 0 || H <= 0 || W <= 0 || K < 3 || K > 31 || (K & 1) == 0 ||
           !(KS == 0 || KS == 3 || KS == 5) || KS > K;
}

template <typename T>
int fwd_impl(const T* x, const float* wb, const float* ws, T* yb, T* ys, int N, int C, int H, int W,
             int K, int KS, void* stream) {
    if (bad_shape(N, C, H, W, K, KS)) return PPEA_ERR_UNSUPPORTED;
    if (ws == nullptr || ys == nullptr) { KS = 0; ws = nullptr; ys = nullptr; }
    hipStream_t st = (hipStream_t)stream;
    Shape s{N, C, H, W};
    if (KS == 0 || KS == 5) {
        const int r = dispatch<T, false>(x, nullptr, wb, ws, yb, ys, s, K, KS, st);
        if (r != PPEA_ERR_UNSUPPORTED) return r;
    }
    int r = run_generic<T>(x, wb, yb, s, K, false, 0, st);
    if (r == 0 && KS > 0) r = run_generic<T>(x, ws, ys, s, KS, false, 0, st);
    return r;
}

template <typename T>
int bwd_impl(const T* dyb, const T* dys, const float* wb, const float* ws, T* dx, int N, int C, int H,
             int W, int K, int KS, void* stream) {
    if (bad_shape(N, C, H, W, K, KS)) return PPEA_ERR_UNSUPPORTED;
    if (ws == nullptr || dys == nullptr) { KS = 0; ws = nullptr; dys = nullptr; }
    hipStream_t st = (hipStream_t)stream;
    Shape s{N, C, H, W};
    if (KS == 0 || KS == 5) {
        const int r = dispatch<T, true>(dyb, dys, wb, ws, dx, nullptr, s, K, KS, st);
        if (r != PPEA_ERR_UNSUPPORTED) return r;
    }
    int r = run_generic<T>(dyb, wb, dx, s, K, true, 0, st);
    if (r == 0 && KS > 0) r = run_generic<T>(dys, ws, dx, s, KS, true, 1, st);
    return r;
}

// measurement aid: one lane writes the constant-rate device clock (s_memrealtime, 100 MHz) to *slot
__global__ void timestamp_kernel(unsigned long long* slot) { *slot = wall_clock64(); }

}  // namespace

extern "C" {

int ppea_abi_version(void) { return PPEA_ABI_VERSION; }

int ppea_timestamp(void* slot, void* stream) {
    if (slot == nullptr) return PPEA_ERR_ARG;
    hipLaunchKernelGGL(timestamp_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, (unsigned long long*)slot);
    return launch_status();
}

int ppea_dwconv_lk_fwd_f32(const float* x, const float* w_big, const float* w_small, float* y_big,
                           float* y_small, int N, int C, int H, int W, int K, int KS, void* stream) {
    return fwd_impl<float>(x, w_big, w_small, y_big, y_small, N, C, H, W, K, KS, stream);
}
int ppea_dwconv_lk_fwd_bf16(const uint16_t* x, const float* w_big, const float* w_small, uint16_t* y_big,
                            uint16_t* y_small, int N, int C, int H, int W, int K, int KS, void* stream) {
    return fwd_impl<uint16_t>(x, w_big, w_small, y_big, y_small, N, C, H, W, K, KS, stream);
}
int ppea_dwconv_lk_bwd_data_f32(const float* dy_big, const float* dy_small, const float* w_big,
                                const float* w_small, float* dx, int N, int C, int H, int W, int K, int KS,
                                void* stream) {
    return bwd_impl<float>(dy_big, dy_small, w_big, w_small, dx, N, C, H, W, K, KS, stream);
}
int ppea_dwconv_lk_bwd_data_bf16(const uint16_t* dy_big, const uint16_t* dy_small, const float* w_big,
                                 const float* w_small, uint16_t* dx, int N, int C, int H, int W, int K,
                                 int KS, void* stream) {
    return bwd_impl<uint16_t>(dy_big, dy_small, w_big, w_small, dx, N, C, H, W, K, KS, stream);
}
int ppea_dwconv_lk_bwd_filter_f32(const float* x, const float* dy, float* dw, int N, int C, int H, int W,
                                  int K, void* stream) {
    if (bad_shape(N, C, H, W, K, 0)) return PPEA_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(dwconv_wgrad_kernel, dim3(C, K), dim3(256), 0, (hipStream_t)stream, x, dy, dw, N, C,
                       H, W, K);
    return launch_status();
}

}  // extern "C"
