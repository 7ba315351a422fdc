// Kernel-side argument block of pfd_attention_f16 (attention.hip, attention3.hip).
#pragma once
#include "pfd_common.h"

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

struct AttnParams {
  const half_t* Q;
  const half_t* K;
  const half_t* Vt;
  half_t* O;
  long ldq, ldk, ldvt, ldo;
  long q_bs, k_bs, vt_bs, o_bs;
  int B, H, Nq, Nk, D;
  float scale_log2;
};

// attention3.hip: the software-pipelined d = 40 kernel (64 queries per wave); `takes` = shapes it is built for
bool pfd_attention3_takes(const AttnParams& p);
void pfd_attention3_launch(const AttnParams& p, hipStream_t s);
