#include "decoder_kernel.h"
int dec_run_ber_ns3(int mode, const DecArgs& g, int kp, int grid, hipStream_t stream) {
  if (mode == MODE_BER_TRAIN) return dec_run_mode<3, MODE_BER_TRAIN>(g, kp, grid, stream);
  return dec_run_mode<3, MODE_BER_EVAL>(g, kp, grid, stream);
}
