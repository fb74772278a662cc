// conv engine: instance group 2 of the DMA-pipelined patch GEMM kernel (conv_dma.h)
#include "conv_dma.h"

namespace p2i {
int dispatch_patch_dma_g2(const TileCfg& c, int KG, const PatchGeom& g, dim3 grid, size_t lds, hipStream_t s) {
  P2I_DMA_CASE(128, 256, 2, 8, 1)
  P2I_DMA_CASE(64, 256, 1, 8, 1)
  P2I_DMA_CASE(128, 128, 2, 8, 1)
  P2I_DMA_CASE(64, 128, 2, 8, 1)
  P2I_DMA_CASE(32, 128, 1, 8, 1)
  return -1;
}
}  // namespace p2i
