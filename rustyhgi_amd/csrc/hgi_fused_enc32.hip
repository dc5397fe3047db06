// Encode, 128 x 32 tiles: the latency build (same source as hgi_fused_enc.hip).  1920 x 1080 L4: 14.0 -> 9.5 us.
#define HGI_TILE_H 32
#include "hgi_fused_enc.hip"
