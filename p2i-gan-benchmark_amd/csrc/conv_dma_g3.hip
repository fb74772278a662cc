// conv engine: instance group 3 of the DMA-pipelined patch GEMM kernel (conv_dma.h)
#include "conv_dma.h"

namespace p2i {
int dispatch_patch_dma_g3(const TileCfg& c, int KG, const PatchGeom& g, dim3 grid, size_t lds, hipStream_t s) {
  P2I_DMA_CASE(128, 256, 2, 4, 1)
  P2I_DMA_CASE(64, 256, 1, 4, 1)
  P2I_DMA_CASE(64, 128, 2, 4, 1)
  P2I_DMA_CASE(32, 128, 1, 4, 1)
  P2I_DMA_CASE(64, 256, 1, 4, 2)
  P2I_DMA_CASE(64, 128, 2, 4, 2)
  P2I_DMA_CASE(32, 128, 1, 4, 2)
  P2I_DMA_CASE(64, 128, 2, 2, 1)
  P2I_DMA_CASE(32, 128, 1, 2, 1)
  return -1;
}
}  // namespace p2i
