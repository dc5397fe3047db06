// Decode, 128 x 16 tiles: the single-small-frame build (same source as hgi_fused_dec.hip; pyramids up to four levels).
#define HGI_TILE_H 16
#include "hgi_fused_dec.hip"
