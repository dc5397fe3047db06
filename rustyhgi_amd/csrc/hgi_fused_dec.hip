// Decode direction of the fused kernels (k_dec_tiles + launch_decode_fused).
// Layout: only the even rows of a tile live in LDS (odd rows stay in the registers they were loaded into),
// rows unpadded: k = 4 needs 4 848 B of LDS per wave; ~50 VGPRs, so 8 waves per SIMD.  Measured on the
// 64 x 4096^2 shard: row pad 0 / 16 / 32 B -> 0.396 / 0.403 / 0.402 ms.
#define HGI_FUSED_DECODE 1
#ifndef HGI_S_PAD
#define HGI_S_PAD 0
#endif
#ifndef HGI_S2_PAD
#define HGI_S2_PAD 8
#endif
#include "hgi_fused_impl.h"
