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
// creff_roll.hip: the rolling warp + CReFF kernel for C = 64, heads of up to 16 classes (arguments already checked by arseg_creff_warp_fwd_ex);
// seg_rows = 0: default; max_wgs = 0: one workgroup per compute unit; dry_run: admission only (ARSEG_OK / ARSEG_EUNSUPPORTED, nothing is launched or
// dereferenced; n_cls > 0 stands for "with a head")
int arseg_creff_roll_launch(const float *const *ref_nhwc_host, const int16_t *mv_q, int H, int W, const float *lr, const float *wq,
                            const float *bq, const float *wk, const float *bk, const float *wv, const float *bv, float *p_out,
                            int p_layout, const float *wf, const float *bf, int n_cls, float *logits, int log_softmax, int N, int Hp,
                            int Wp, int hp, int wp, int seg_rows, int max_wgs, bool dry_run, hipStream_t st);
