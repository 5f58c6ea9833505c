// LAMA_PREC_F16 (BASELINE configs[2]): fp16 activation tensors, fp16 weights, ONE v_mfma_f32_32x32x16_f16 product per MAC
// (body: conv_split3.inc, the half-I/O launches only)
#define CB_F16 1
#define CB_HALF 1
#include "conv_split3.inc"
