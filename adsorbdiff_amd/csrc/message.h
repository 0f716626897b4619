// Shared declarations of the message kernels (message.hip: forward, two waves per SIMD, f16x3 / exact-f32 / non-uniform-centre
// modes; message_bwd.hip: the training step's fused backward).  The retired variants (message32 / message3 / message4 /
// message_il, all measured slower) live in scratch/experiments/.
#pragma once
#include "common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));

#define MSG_THREADS 512
#define MSG_WAVES 8
#define MSG_COLS 192
#define MSG_LDK 136  // halves per column row of the f16 weight image (128 + 8 pad: conflict-free b128 reads)

struct MsgParams {
    const float* rec;   // gather records [(N+1)][H/32][160] (gemm16.hip EPI 1 / adf_pack_records_kernel); row N zero
    const float* vec;
    const float* x;
    float* x_out;
    float* vec_out;
    const int32_t* tlist;    // optional: targets to evaluate (ascending atom indices); outputs are then compact rows
    int items;               // number of targets: N, or the length of tlist
    const int32_t* items_dev;  // optional: the list length on the device (items is then its upper bound)
    const int32_t* nptr;
    const int32_t* e_src;
    const float4* e_geom;
    const float* wpack;      // f32 image  [slice][R][192]
    const _Float16* wpack16; // f16 image  [slice][hi|lo][192][R]
    const float* bpack;      // [slice][192] bias (f32 mode) or bias * scale (f16 mode)
    const float* inv_scale;  // device scalar (f16 mode)
    const float* mu;
    int N, H, R, G, nslices;
    float inv_cutoff, coeff, sarg, env_a, env_b, env_c;
    float dmu2, dmusq, cstep;  // UNI: 2*dmu', dmu'^2, exp2(-2 dmu'^2) with dmu' = scaled spacing of the centres
    int env_pi;
    unsigned long long* kcount;  // optional: sum over 32-row blocks of contracted k length x 32-column blocks run (profiling)
};


