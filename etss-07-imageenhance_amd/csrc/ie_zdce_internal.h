// Internal (non-ABI) interface between the Zero-DCE plan and its kernel launchers.
#pragma once
#include "ie_common.h"

struct ZdceConvLaunch {
    const void* src0; const void* src1; const void* wpk; const float* bias; void* dst;
    const float* img; float* enhanced; float* adjust; unsigned char* u8;
    int N, H, W, y_begin, y_end, n_begin, n_count, rows_per_block;
    int nsrc, epi, variant;
    int dt0, dt1, dto;   // element types (IE_F16 / IE_BF16) of source 0, source 1 and the ReLU output
};
int ie_zdce_launch_mfma(const ZdceConvLaunch& a, hipStream_t st);
int ie_zdce_launch_conv1(const float* img, const float* wk, const float* bias, void* dst, int dtype, int N, int H,
                         int W, int y_begin, int y_end, int n_begin, int n_count, hipStream_t st);

// two-stage fused strip kernels (ie_zdce_fused.hip); kind = 12, 45, 67 or 345 (e_conv3 + e_conv4 on stage A, e_conv5 on stage B)
struct ZdceFusedLaunch {
    int kind;
    int dt_a0, dt_a1;    // element types of stage A's HBM sources (K45: x3; K67: x2, x5)
    int dt_h;            // element type of the LDS hand-off (x1 / x4 / x6), of stage B's weights and of K67's P
    int dt_o;            // element type of the HBM outputs (K12: x2 and P; K45: x5)
    const void* a_src0; const void* a_src1; const void* b_src; const float* img;
    const void* a_w; const float* a_bias; const void* b_w; const float* b_bias; const float* b2_bias;
    const void* a2_w; const float* a2_bias;   // kind 345: e_conv4
    void* a_dst; void* b_dst; float* enhanced; float* adjust; unsigned char* u8;
    int N, H, W, y_begin, y_end, n_begin, n_count, rows_per_block;
    void* stamps;   // diagnostic builds only
};
int ie_zdce_launch_fused(const ZdceFusedLaunch& a, hipStream_t st);

// round-4 schedule (ie_zdce_x.hip): kind 12 = K12X (e_conv1 + e_conv2: image -> x2), kind 67 = K67X (e_conv1 again + e_conv6 + e_conv7 + curve: image, x2, x5 -> outputs)
struct ZdceXLaunch {
    int kind;
    int dt_h;            // element type of x1 / x6 (LDS only), of e_conv1's operands and of e_conv7's weights; K12X: also x2's
    int dt_a0, dt_a1;    // K67X: element types of x2 and x5
    const float* img;
    void* x2;            // K12X: destination; K345X / K67X: source
    const void* x5;      // K345X: destination; K67X: source
    const void* w1; const float* b1;     // e_conv1 as [cout half][ky][lane][8] fragments over [R, G, B, 0] pixels (K = 8 kx taps x 4 channels)
    const void* w1q;                     // K67X: e_conv1 as [cout half][lane][8]: ONE K = 32 fragment over a 3-row window (lane groups 0-2: row ky, (kx, c) pairs 0..7; group 3: the three (kx 2, B) taps)
    const void* w2; const float* b2;     // K12X: e_conv2 fragments
    const void* w6; const float* b6;     // K67X: e_conv6 fragments
    const void* w3; const float* b3; const void* w4; const float* b4; const void* w5; const float* b5;     // K345X: e_conv3 / 4 / 5 fragments ([cout half][src * 9 + tap][lane][8])
    int dt_3;            // K345X: element type of x3 / x4 / x5 (dt_a0 = x2's)
    const void* w7; const float* b7;     // K67X: e_conv7 over [x1, x6], rows permuted (chan_k67) and scaled by 2 / ln 2: [tile][18][lane][8]; bias in row order
    float* enhanced; float* adjust; unsigned char* u8;
    int N, H, W, y_begin, y_end, n_begin, n_count, rows_per_block;
    void* stamps;        // diagnostic builds only
};
int ie_zdce_launch_x(const ZdceXLaunch& a, hipStream_t st);
