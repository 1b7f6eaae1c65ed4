// The fused mesh decoder with its linear products as x3 arithmetic (six bf16 MFMAs per fp32 product on operands split as they are fed): csrc/meshdec.hip
// compiled a third time (see MD_X3 there).  Entry points: pdf_mesh_level_fwd_x3, pdf_mesh_level_bwd_x3 (same argument block, same tape layout as the fp32 build).
#define MD_BF16 1
#define MD_X3 1
#include "meshdec.hip"
