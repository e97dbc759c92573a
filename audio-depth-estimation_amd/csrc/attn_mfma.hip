// bf16 MFMA flash-attention kernels for the full-width binaural head dims (placeholder: not yet taken).
#include "adn_common.h"

int adn_attn_mfma_fwd(const AdnAttnDesc* d, hipStream_t st) {
  (void)d; (void)st;
  return 0;
}

int adn_attn_mfma_bwd(const AdnAttnDesc* d, hipStream_t st) {
  (void)d; (void)st;
  return 0;
}
