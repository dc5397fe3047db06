// Encode direction of the fused kernels (k_enc_tiles + launch_encode_fused).
// Layout: even rows only in LDS (odd rows stay in registers), unpadded: k = 4 needs 7 648 B per wave
// (21 waves per CU); the kernel is held to 96 VGPRs = 5 waves per SIMD.  Row pad 0 / 32 B -> 0.414 / 0.422 ms
// on the 64 x 4096^2 shard.
#define HGI_FUSED_ENCODE 1
#if defined(HGI_ENC_STORE_AUX) && !defined(HGI_STORE_AUX)
#define HGI_STORE_AUX HGI_ENC_STORE_AUX      // experiments: a store policy for this direction only
#endif
#ifndef HGI_S_PAD
#define HGI_S_PAD 0
#endif
#ifndef HGI_S2_PAD
#define HGI_S2_PAD 0
#endif
// Interior tiles are walked in bands of HGI_TILE_BAND tile rows, column-major inside a band (fast_tile()): tiles that share
// halo lines with their right neighbour are then dispatched a band height apart instead of back to back.  Measured in
// one process on the same planes (tools/ab.py, profiles/r02_ab_tile_order*.txt): encode 4 rows -3.6 ... -4.1 %, 8 rows -1.6 ... -2.5 %, 16 rows and more +2 %.
#ifndef HGI_TILE_ORDER
#define HGI_TILE_ORDER 3
#endif
#ifndef HGI_TILE_BAND
#define HGI_TILE_BAND 4
#endif
#include "hgi_fused_impl.h"
