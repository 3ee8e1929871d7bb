// Launch parameters shared by the two CReFF kernels (creff.hip: fp32 VALU; creff_mfma.hip: split-fp16 matrix cores).
#pragma once
#include "arseg_common.h"

struct CreffParams {
    const float *hr, *lr, *wq, *bq, *wk, *bk, *wv, *bv, *wf, *bf;
    float *p_out, *logits;
    int N, C, Hp, Wp, hp, wp, n_cls, log_softmax;
    unsigned p_bytes, l_bytes;
    float sy, sx;   // align_corners=True source scales (hp-1)/(Hp-1), (wp-1)/(Wp-1)
    int mfma_tile_rows;   // matrix-core kernel: 0 = choose by launch size, 8 or 16 = pin the tile height
};

// creff_mfma.hip; returns ARSEG_EUNSUPPORTED for shapes it does not cover (the caller then uses the VALU kernel)
int arseg_creff_mfma_launch(const CreffParams &p, hipStream_t st);
