// LAMA_PREC_F16X3: the 3-term split convolution on v_mfma_f32_32x32x16_f16 (body: conv_split3.inc)
#define CB_F16 1
#include "conv_split3.inc"
