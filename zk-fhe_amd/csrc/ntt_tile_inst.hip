// One tile size of the NTT kernel per translation unit: compile with -DZK_TILE_LOGN=k (3..13).
#include "ntt_tile.hip.hpp"

#ifndef ZK_TILE_LOGN
#error "define ZK_TILE_LOGN"
#endif
#define ZK_CAT2(a, b) a##b
#define ZK_CAT(a, b) ZK_CAT2(a, b)

int ZK_CAT(zk_launch_tile_, ZK_TILE_LOGN)(zkfhe_ctx *ctx, const zk::TileArgs &a, unsigned tiles, unsigned cols) {
  return zk::launch_tile<ZK_TILE_LOGN>(ctx, a, tiles, cols);
}
