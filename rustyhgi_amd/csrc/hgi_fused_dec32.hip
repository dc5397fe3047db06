// Decode, 128 x 32 tiles: the latency build (same source as hgi_fused_dec.hip; picked by hgi_capi.hip for calls
// with too few tiles to fill the GPU).  1920 x 1080 L4: 9.4 -> 6.8 us.
#define HGI_TILE_H 32
#include "hgi_fused_dec.hip"
