// Encode, 128 x 16 tiles: the single-small-frame build (same source as hgi_fused_enc.hip; pyramids up to four levels).
// A lone frame's launch ends when its slowest wave does, and a wave's chain is mostly its own VALU work: half the rows
// per wave halve the finest level's share of it.  Picked by hgi_capi.hip (use_tile_rows) below a few hundred tiles.
#define HGI_TILE_H 16
#include "hgi_fused_enc.hip"
