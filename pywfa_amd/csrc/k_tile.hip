// k_tile.hip — translation unit of the tiled wide-wavefront kernel (wfa_tile.hpp).
#include "wfa_tile.hpp"

namespace wfa {
template <bool FULL, bool TWO, int NCH>
static int launch_tile_t(const TileArgs& a, int grid, int threads, size_t smem, hipStream_t stream) {
  if (hipFuncSetAttribute(reinterpret_cast<const void*>(&wfa_tile_kernel<FULL, TWO, NCH>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) != hipSuccess) {
    (void)hipGetLastError();
    return -1;
  }
  hipLaunchKernelGGL((wfa_tile_kernel<FULL, TWO, NCH>), dim3(grid), dim3(threads), smem, stream, a);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}
template <bool FULL, bool TWO, int NCH>
static int tile_occupancy_t(int threads, size_t smem) {
  int n = 0;
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&wfa_tile_kernel<FULL, TWO, NCH>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, wfa_tile_kernel<FULL, TWO, NCH>, threads, smem) != hipSuccess) { (void)hipGetLastError(); return 1; }
  return n > 0 ? n : 1;
}
#define WFA_TILE_DISPATCH(FN, ...)                                                         \
  switch ((full ? 8 : 0) + (two ? 4 : 0) + (nch - 1)) {                                   \
    case 0: return FN<false, false, 1>(__VA_ARGS__); case 1: return FN<false, false, 2>(__VA_ARGS__);   \
    case 2: return FN<false, false, 3>(__VA_ARGS__); case 3: return FN<false, false, 4>(__VA_ARGS__);   \
    case 4: return FN<false, true, 1>(__VA_ARGS__); case 5: return FN<false, true, 2>(__VA_ARGS__);     \
    case 6: return FN<false, true, 3>(__VA_ARGS__); case 7: return FN<false, true, 4>(__VA_ARGS__);     \
    case 8: return FN<true, false, 1>(__VA_ARGS__); case 9: return FN<true, false, 2>(__VA_ARGS__);     \
    case 10: return FN<true, false, 3>(__VA_ARGS__); case 11: return FN<true, false, 4>(__VA_ARGS__);   \
    case 12: return FN<true, true, 1>(__VA_ARGS__); case 13: return FN<true, true, 2>(__VA_ARGS__);     \
    case 14: return FN<true, true, 3>(__VA_ARGS__); default: return FN<true, true, 4>(__VA_ARGS__);     \
  }
// (round 6) int32 cells: reads beyond 32 000 bases; the geometries the host picks for them (gap-affine 256 columns, gap-affine-2p 128)
template <bool FULL, bool TWO, int NCH>
static int launch_tile32_t(const TileArgs& a, int grid, int threads, size_t smem, hipStream_t stream) {
  if (hipFuncSetAttribute(reinterpret_cast<const void*>(&wfa_tile_kernel<FULL, TWO, NCH, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) != hipSuccess) {
    (void)hipGetLastError();
    return -1;
  }
  hipLaunchKernelGGL((wfa_tile_kernel<FULL, TWO, NCH, true>), dim3(grid), dim3(threads), smem, stream, a);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}
template <bool FULL, bool TWO, int NCH>
static int tile_occupancy32_t(int threads, size_t smem) {
  int n = 0;
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&wfa_tile_kernel<FULL, TWO, NCH, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, wfa_tile_kernel<FULL, TWO, NCH, true>, threads, smem) != hipSuccess) { (void)hipGetLastError(); return 1; }
  return n > 0 ? n : 1;
}
int launch_tile(bool full, bool two, const TileArgs& a, int grid, int threads, size_t smem, hipStream_t stream, bool w32) {
  const int nch = a.g.Wt / 64;
  if (nch < 1 || nch > 4 || a.g.Wt % 64) return -1;
  if (w32) {
    if (two) { if (nch != 2) return -1; return full ? launch_tile32_t<true, true, 2>(a, grid, threads, smem, stream) : launch_tile32_t<false, true, 2>(a, grid, threads, smem, stream); }
    if (nch != 4) return -1;
    return full ? launch_tile32_t<true, false, 4>(a, grid, threads, smem, stream) : launch_tile32_t<false, false, 4>(a, grid, threads, smem, stream);
  }
  WFA_TILE_DISPATCH(launch_tile_t, a, grid, threads, smem, stream)
}
int tile_occupancy(bool full, bool two, int nch, int threads, size_t smem, bool w32) {
  if (nch < 1 || nch > 4) return 1;
  if (w32) {
    if (two) return full ? tile_occupancy32_t<true, true, 2>(threads, smem) : tile_occupancy32_t<false, true, 2>(threads, smem);
    return full ? tile_occupancy32_t<true, false, 4>(threads, smem) : tile_occupancy32_t<false, false, 4>(threads, smem);
  }
  WFA_TILE_DISPATCH(tile_occupancy_t, threads, smem)
}
}  // namespace wfa
