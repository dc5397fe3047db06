// Decode direction of the fused kernels (k_dec_tiles + launch_decode_fused).
// Layout: only the even rows of a tile live in LDS (odd rows stay in the registers they were loaded into),
// rows unpadded: k = 4 needs 4 848 B of LDS per wave; ~50 VGPRs, so 8 waves per SIMD.  Measured on the
// 64 x 4096^2 shard: row pad 0 / 16 / 32 B -> 0.396 / 0.403 / 0.402 ms.
#define HGI_FUSED_DECODE 1
#if defined(HGI_DEC_STORE_AUX) && !defined(HGI_STORE_AUX)
#define HGI_STORE_AUX HGI_DEC_STORE_AUX      // experiments: a store policy for this direction only
#endif
#ifndef HGI_S_PAD
#define HGI_S_PAD 0
#endif
#ifndef HGI_S2_PAD
#define HGI_S2_PAD 8
#endif
// Interior tiles are walked in bands of HGI_TILE_BAND tile rows, column-major inside a band (fast_tile()): tiles that share
// halo lines with their right neighbour are then dispatched a band height apart instead of back to back.  Measured in
// one process on the same planes (tools/ab.py, profiles/r02_ab_tile_order*.txt; bands dealt contiguously to the XCDs, round 2): decode 8 rows
// -3.0 ... -4.0 %, 4 rows -2 ... -2.6 %.  Round 3 deals the bands round-robin (block_role, g.xmode) and there four rows win in both
// directions (profiles/r03_order_sweep.txt); the height is capped by address span at run time (band_rows()).
#ifndef HGI_TILE_ORDER
#define HGI_TILE_ORDER 3
#endif
#ifndef HGI_TILE_BAND
#define HGI_TILE_BAND 4
#endif
#include "hgi_fused_impl.h"
