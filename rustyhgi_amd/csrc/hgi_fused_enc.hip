// Encode direction of the fused kernels (k_enc_tiles + launch_encode_fused).
// Layout: unpadded rows -- encode is bound by LDS latency chains, so the 13th wave per CU that the
// smaller footprint buys (k = 4: 11 936 B per wave) is worth more than the bank conflicts cost:
// 0.487 -> 0.461 ms per 64 x 4096^2 frames.
#define HGI_FUSED_ENCODE 1
#define HGI_S_PAD 0
#define HGI_S2_PAD 0
#include "hgi_fused_impl.h"
