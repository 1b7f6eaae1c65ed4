// The fused mesh decoder with its linear products on the bf16 MFMA: csrc/meshdec.hip compiled a second time (see MD_BF16 there).
// Entry points: pdf_mesh_level_fwd_bf16, pdf_mesh_level_bwd_bf16 (same argument block, same tape layout as the fp32 build).
#define MD_BF16 1
#include "meshdec.hip"
