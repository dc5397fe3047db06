// Encode direction of the fused kernels (k_enc_tiles + launch_encode_fused).
// Layout: even rows only in LDS (odd rows stay in registers), unpadded: k = 4 needs 7 648 B per wave
// (21 waves per CU); the kernel is held to 96 VGPRs = 5 waves per SIMD.  Row pad 0 / 32 B -> 0.414 / 0.422 ms
// on the 64 x 4096^2 shard.
#define HGI_FUSED_ENCODE 1
#ifndef HGI_S_PAD
#define HGI_S_PAD 0
#endif
#ifndef HGI_S2_PAD
#define HGI_S2_PAD 0
#endif
#include "hgi_fused_impl.h"
