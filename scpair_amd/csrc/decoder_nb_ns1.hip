#include "decoder_kernel.h"
int dec_run_nb_ns1(int mode, const DecArgs& g, int kp, int grid, hipStream_t stream) {
  if (mode == MODE_NB_STATS) return dec_run_mode<1, MODE_NB_STATS>(g, kp, grid, stream);
  // dispersion = 'feature' (theta_param) has its own instantiations of the likelihood and gradient passes
  if (mode == MODE_NB_LOSS) return g.theta_param ? dec_run_mode<1, MODE_NB_LOSS_TH>(g, kp, grid, stream) : dec_run_mode<1, MODE_NB_LOSS>(g, kp, grid, stream);
  return g.theta_param ? dec_run_mode<1, MODE_NB_GRAD_TH>(g, kp, grid, stream) : dec_run_mode<1, MODE_NB_GRAD>(g, kp, grid, stream);
}
