// placeholder until the backward kernels land (next commit)
#include "fa_launch.h"
namespace fa {
int launch_bwd_delta(const BwdK&, int, int, hipStream_t) { return -2; }
int launch_bwd_dkdv(const BwdK&, int, int, hipStream_t) { return -2; }
int launch_bwd_dq(const BwdK&, int, int, hipStream_t) { return -2; }
int bwd_block_m() { return 128; }
int bwd_block_n() { return 128; }
}  // namespace fa
