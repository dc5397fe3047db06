// Decode direction of the fused kernels (k_dec_tiles + launch_decode_fused).
// Layout: rows padded by 16 B -- decode sits on the HBM floor (0.385-0.40 ms per 64 x 4096^2 frames) and
// loses 5 % to LDS bank conflicts without the pad.  k = 4: 10 224 B of LDS per wave, 16 waves per CU.
#define HGI_FUSED_DECODE 1
#define HGI_S_PAD 16
#define HGI_S2_PAD 8
#include "hgi_fused_impl.h"
