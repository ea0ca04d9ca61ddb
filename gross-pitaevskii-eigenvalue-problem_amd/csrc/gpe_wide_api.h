// gpe_wide_api.h -- host entry points of the wide kernel set (gpe_wide.h), compiled in its own translation unit
// (gpe_wide.hip) so that the two kernel sets build in parallel.
#pragma once
#include "gpe_common.h"

struct WideCall {
    NetDesc nd;
    const float* theta; const float* Wpk; const float* WpkT;
    Pts pts;
    float* stored;               // [tile][L-1][C][H/16][256]
    float* O; const float* Ob;   // output jets / their adjoint [C][n_out][ld]
    float* Z0; float* Z1;        // adjoint-jet ping-pong buffers [tile][C][H/16][256]
    float* gslab;                // [G][Ppad]
    int64_t N, ld;
    int Ppad, H, C, E, num_cu;
    hipStream_t stream;
};
// shapes the set is compiled for: uniform hidden width 256 (any dim), or 128
bool wide_shape_ok(int H);
// number of gradient slabs (= tile groups) a reverse pass over N points writes; every slab is written completely
int wide_groups(int H, int64_t N, int num_cu);
void wide_init();                                        // dynamic-LDS attributes, once per process
int wide_forward(const WideCall& a, int store_acts);     // hipSuccess (0), or -1: channel pair not compiled
const char* wide_forward_kernel(int H);                  // "w_forward_mt" (H = 128: several tiles per pass) or "w_forward"
int wide_backward(const WideCall& a);                    // launches w_bwd_out + one w_bwd_map per hidden->hidden map
