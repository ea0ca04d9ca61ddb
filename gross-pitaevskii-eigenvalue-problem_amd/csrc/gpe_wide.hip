// gpe_wide.hip -- launchers of the wide kernel set (gpe_wide.h).  gfx950 only.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <stdlib.h>

#include "gpe_wide.h"
#include "gpe_wide_api.h"

static size_t small4(const NetDesc& nd, int H) { return (size_t)(((4 + (nd.n_lin - 2) + nd.n_out) * H + 4 + 3) & ~3); }

bool wide_shape_ok(int H) { return H == 128 || H == 256; }

int wide_groups(int H, int64_t N, int num_cu) {
    const int nsplit = H / 128;
    int64_t g = std::min<int64_t>((N + 15) / 16, num_cu / nsplit);
    if (g >= 8) g &= ~(int64_t)7;          // multiples of 8: the halves of a tile group land on one XCD
    return (int)std::max<int64_t>(g, 1);
}

#ifdef GPE_FAST_BUILD
#define W_FOR_SHAPES(X) X(256, 1, 0) X(256, 5, 1)
#define W_FOR_FWD_ONLY(X)
#define W_NOUT2 0
#else
#define W_FOR_SHAPES(X) X(256, 1, 0) X(256, 3, 1) X(256, 4, 1) X(256, 5, 1) X(128, 1, 0) X(128, 3, 1) X(128, 4, 1) X(128, 5, 1)
#define W_FOR_FWD_ONLY(X) X(256, 5, 2) X(256, 7, 3) X(128, 5, 2) X(128, 7, 3)
#define W_NOUT2 1
#endif

void wide_init() {
    static bool done = false;
    if (done) return;
    done = true;
    const int lds = 160 * 1024;
#define ATTR_F(HH, CC, EE)                                                                                               \
    (void)hipFuncSetAttribute((const void*)w_forward<HH, CC, EE, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);   \
    if (W_NOUT2) (void)hipFuncSetAttribute((const void*)w_forward<HH, CC, EE, W_NOUT2 ? 2 : 1>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
#define ATTR_B(HH, CC, EE)                                                                                                                \
    (void)hipFuncSetAttribute((const void*)w_bwd_map<HH, CC, EE, HH / 128, false, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);   \
    (void)hipFuncSetAttribute((const void*)w_bwd_map<HH, CC, EE, HH / 128, true, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);    \
    (void)hipFuncSetAttribute((const void*)w_bwd_map<HH, CC, EE, HH / 128, false, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);   \
    (void)hipFuncSetAttribute((const void*)w_bwd_map<HH, CC, EE, HH / 128, true, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);    \
    if (W_NOUT2) (void)hipFuncSetAttribute((const void*)w_bwd_map<HH, CC, EE, HH / 128, false, W_NOUT2 ? 2 : 1>, hipFuncAttributeMaxDynamicSharedMemorySize, lds); \
    if (W_NOUT2) (void)hipFuncSetAttribute((const void*)w_bwd_map<HH, CC, EE, HH / 128, true, W_NOUT2 ? 2 : 1>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);  \
    (void)hipFuncSetAttribute((const void*)w_bwd_out<HH, CC, EE, 1, HH / 128>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);          \
    if (W_NOUT2) (void)hipFuncSetAttribute((const void*)w_bwd_out<HH, CC, EE, W_NOUT2 ? 2 : 1, HH / 128>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    W_FOR_SHAPES(ATTR_F) W_FOR_FWD_ONLY(ATTR_F) W_FOR_SHAPES(ATTR_B)
#undef ATTR_F
#undef ATTR_B
}

template <int HH, int CC, int EE>
static void launch_fwd(const WideCall& a, int store) {
    const int NT = HH / 16;
    const int64_t ntiles = (a.N + 15) / 16;
    const unsigned grid = (unsigned)std::max<int64_t>(1, std::min<int64_t>(ntiles, a.num_cu));
    const size_t lds = (small4(a.nd, HH) + (size_t)CC * NT * 256 + (size_t)W_NW * a.nd.n_out * CC * 16) * sizeof(float);
    if (a.nd.n_out == 1)
        hipLaunchKernelGGL((w_forward<HH, CC, EE, 1>), dim3(grid), dim3(512), lds, a.stream, a.nd, a.theta, a.Wpk, a.pts, a.stored, a.O,
                           a.N, a.ld, store);
    else
        hipLaunchKernelGGL((w_forward<HH, CC, EE, W_NOUT2 ? 2 : 1>), dim3(grid), dim3(512), lds, a.stream, a.nd, a.theta, a.Wpk, a.pts,
                           a.stored, a.O, a.N, a.ld, store);
}

// H = 128: several tiles per pass (w_forward_mt).  TP by LDS: TP C NT 1 KiB + small operands <= 160 KB -> 4 tiles up to C = 4, 3 at C = 5, 2 at C = 7
template <int CC, int EE, int TP>
static void launch_fwd_mt(const WideCall& a, int store) {
    constexpr int HH = 128, NT = HH / 16;
    const int64_t ntiles = (a.N + 15) / 16, npass = (ntiles + TP - 1) / TP;
    const unsigned grid = (unsigned)std::max<int64_t>(1, std::min<int64_t>(npass, a.num_cu));
    const size_t lds = (small4(a.nd, HH) + (size_t)TP * CC * NT * 256 + (size_t)TP * W_NW * a.nd.n_out * CC * 16) * sizeof(float);
    static bool attr_done = false;
    if (!attr_done) {
        (void)hipFuncSetAttribute((const void*)w_forward_mt<HH, CC, EE, 1, TP>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (W_NOUT2) (void)hipFuncSetAttribute((const void*)w_forward_mt<HH, CC, EE, W_NOUT2 ? 2 : 1, TP>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_done = true;
    }
    if (a.nd.n_out == 1)
        hipLaunchKernelGGL((w_forward_mt<HH, CC, EE, 1, TP>), dim3(grid), dim3(512), lds, a.stream, a.nd, a.theta, a.Wpk, a.pts, a.stored, a.O, a.N, a.ld, store);
    else
        hipLaunchKernelGGL((w_forward_mt<HH, CC, EE, W_NOUT2 ? 2 : 1, TP>), dim3(grid), dim3(512), lds, a.stream, a.nd, a.theta, a.Wpk, a.pts, a.stored, a.O,
                           a.N, a.ld, store);
}

static int wide_fwd_mt() {
#ifdef GPE_FAST_BUILD
    return 0;
#else
    // opt-in (GPE_WIDE_FWD_MT=1): measured SLOWER than w_forward at H = 128 -- cfg3 forward 0.644 against 0.610 ms (f_forward_coop<128>: 0.580),
    // cfg4 1.56-1.58 against 1.52-1.54 (1.46), profiles/r04/wide_forward_mt_ab.txt
    static const int mt = [] { const char* v = getenv("GPE_WIDE_FWD_MT"); return v ? atoi(v) : 0; }();
    return mt;
#endif
}
const char* wide_forward_kernel(int H) { return (H == 128 && wide_fwd_mt()) ? "w_forward_mt" : "w_forward"; }

int wide_forward(const WideCall& a, int store) {
    if (!W_NOUT2 && a.nd.n_out != 1) return -1;
#ifndef GPE_FAST_BUILD
    if (a.H == 128 && wide_fwd_mt()) {
        if (a.C == 1 && a.E == 0) { launch_fwd_mt<1, 0, 4>(a, store); return (int)hipGetLastError(); }
        if (a.C == 3 && a.E == 1) { launch_fwd_mt<3, 1, 4>(a, store); return (int)hipGetLastError(); }
        if (a.C == 4 && a.E == 1) { launch_fwd_mt<4, 1, 4>(a, store); return (int)hipGetLastError(); }
        if (a.C == 5 && a.E == 1) { launch_fwd_mt<5, 1, 3>(a, store); return (int)hipGetLastError(); }
        if (a.C == 5 && a.E == 2) { launch_fwd_mt<5, 2, 3>(a, store); return (int)hipGetLastError(); }
        if (a.C == 7 && a.E == 3) { launch_fwd_mt<7, 3, 2>(a, store); return (int)hipGetLastError(); }
    }
#endif
#define CASE_F(HH, CC, EE) if (a.H == HH && a.C == CC && a.E == EE) { launch_fwd<HH, CC, EE>(a, store); return (int)hipGetLastError(); }
    W_FOR_SHAPES(CASE_F) W_FOR_FWD_ONLY(CASE_F)
#undef CASE_F
    return -1;
}

template <int HH, int CC, int EE>
static void launch_bwd(const WideCall& a) {
    constexpr int NS = HH / 128, NT = HH / 16;
    const int G = wide_groups(HH, a.N, a.num_cu);
    const unsigned grid = (unsigned)(G * NS);
    const int L = a.nd.n_lin - 1;
    const int no = a.nd.n_out;
    const size_t lds_o = ((size_t)((no * HH + no + 3) & ~3) + small4(a.nd, HH)) * sizeof(float);
    const size_t lds_m = ((size_t)7 * HH + 4 + small4(a.nd, HH) + (W_MAP_PIPE(HH, CC) ? 2 : 1) * ((size_t)CC * NT * 256 + (size_t)CC * W_NW * F_TILE)) * sizeof(float);
    float* zcur = a.Z0;
    float* znext = a.Z1;
    const char* envt = getenv("GPE_WIDE_TOP");                 // 0: the output layer as a launch of its own (w_bwd_out)
    const bool fuse_top = !(envt && atoi(envt) == 0);
    if (!fuse_top) {
        if (no == 1)
            hipLaunchKernelGGL((w_bwd_out<HH, CC, EE, 1, NS>), dim3(grid), dim3(512), lds_o, a.stream, a.nd, a.theta, a.pts, a.stored, a.Ob,
                               zcur, a.gslab, a.N, a.ld, a.Ppad, G);
        else
            hipLaunchKernelGGL((w_bwd_out<HH, CC, EE, W_NOUT2 ? 2 : 1, NS>), dim3(grid), dim3(512), lds_o, a.stream, a.nd, a.theta, a.pts,
                               a.stored, a.Ob, zcur, a.gslab, a.N, a.ld, a.Ppad, G);
    }
#define MAP_ARGS a.nd, j, a.theta, a.WpkT, a.pts, a.stored, zcur, znext, a.gslab, a.N, a.Ppad, G, a.Ob, a.ld
#define MAP_LAUNCH(FIRST_, TOP_) hipLaunchKernelGGL((w_bwd_map<HH, CC, EE, NS, FIRST_, TOP_>), dim3(grid), dim3(512), lds_m, a.stream, MAP_ARGS)
    for (int j = L - 1; j >= 1; --j) {
        const bool top = fuse_top && j == L - 1;
        if (!top) { if (j > 1) MAP_LAUNCH(false, 0); else MAP_LAUNCH(true, 0); }
        else if (no == 1) { if (j > 1) MAP_LAUNCH(false, 1); else MAP_LAUNCH(true, 1); }
        else { if (j > 1) MAP_LAUNCH(false, (W_NOUT2 ? 2 : 1)); else MAP_LAUNCH(true, (W_NOUT2 ? 2 : 1)); }
        std::swap(zcur, znext);
    }
#undef MAP_LAUNCH
#undef MAP_ARGS
}

int wide_backward(const WideCall& a) {
    if (!W_NOUT2 && a.nd.n_out != 1) return -1;
    if ((size_t)a.C * a.nd.n_out * (size_t)a.ld * sizeof(float) >= ((size_t)1 << 32)) return -1;      // the seed array goes through one buffer descriptor
#define CASE_B(HH, CC, EE) if (a.H == HH && a.C == CC && a.E == EE) { launch_bwd<HH, CC, EE>(a); return (int)hipGetLastError(); }
    W_FOR_SHAPES(CASE_B)
#undef CASE_B
    return -1;
}

// diagnostic builds (-DGPE_STAMP) only: read and clear the per-phase cycle counters ([0..7] w_forward, [8..15] w_bwd_map)
extern "C" int gpe_debug_read_wide_stamps(unsigned long long out[16]) {
#ifdef GPE_STAMP
    if (hipDeviceSynchronize() != hipSuccess) return -2;
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(w_stamps), 16 * sizeof(unsigned long long)) != hipSuccess) return -2;
    unsigned long long z[16] = {0};
    if (hipMemcpyToSymbol(HIP_SYMBOL(w_stamps), z, sizeof z) != hipSuccess) return -2;
    return 0;
#else
    for (int i = 0; i < 16; ++i) out[i] = 0;
    return -1;
#endif
}
extern "C" int gpe_debug_read_wide_trace(unsigned long long out[256]) {
#ifdef GPE_STAMP
    if (hipDeviceSynchronize() != hipSuccess) return -2;
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(w_trace), 256 * sizeof(unsigned long long)) != hipSuccess) return -2;
    return 0;
#else
    for (int i = 0; i < 16; ++i) out[i] = 0;
    return -1;
#endif
}
