#include "decoder_kernel.h"
int dec_run_nb_ns3(int mode, const DecArgs& g, int kp, int grid, hipStream_t stream) {
  if (mode == MODE_NB_STATS) return dec_run_mode<3, MODE_NB_STATS>(g, kp, grid, stream);
  if (mode == MODE_NB_LOSS) return dec_run_mode<3, MODE_NB_LOSS>(g, kp, grid, stream);
  return dec_run_mode<3, MODE_NB_GRAD>(g, kp, grid, stream);
}
