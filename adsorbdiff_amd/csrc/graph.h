// Parameters of the periodic top-K neighbour search (graph.hip), shared with the EquiformerV2 path (eqv2_*.hip).
#pragma once
#include "common.h"

struct GraphParams {
    const float* pos;
    const float* cell;
    const int32_t* batch;
    const int32_t* atom_offset;
    int r0, r1, r2;
    float rc2;
    int K;
    int N;
    int32_t* nbr_cnt;
    int32_t* nbr_src;
    int32_t* nbr_shift;
    int32_t* img_cnt;
    int32_t* flags;
    // static-atom cache (optional)
    const int32_t* moving;   // [N] 1 = atom moves between graph builds
    const int32_t* mov_idx;  // moving atoms, grouped by system
    const int32_t* mov_off;  // [B+1]
    float* cache_d2;         // [N,K]
    int32_t* cache_cid;      // [N,K]
    int32_t* cache_cnt;      // [N]
};

__device__ __forceinline__ void decode_shift(int c, int r0, int r1, int r2, float& sa, float& sb, float& sc) {
    const int n2 = 2 * r2 + 1, n1 = 2 * r1 + 1;
    const int ia = c / (n1 * n2);
    const int rem = c - ia * (n1 * n2);
    const int ib = rem / n2;
    const int ic = rem - ib * n2;
    sa = (float)(ia - r0);
    sb = (float)(ib - r1);
    sc = (float)(ic - r2);
}

int32_t adf_topk_launch(const GraphParams& p, int mode, hipStream_t s);
