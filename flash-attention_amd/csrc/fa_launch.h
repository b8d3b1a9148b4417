// Host-side launch entry points of the kernel translation units (internal to libfa_gfx950.so).
#pragma once
#include <hip/hip_runtime.h>
#include "fa_kernel_params.h"

namespace fa {

// Forward.  `nw` = waves per workgroup (4 or 8); query block = 32*nw rows.  Returns 0, -1 (launch
// failure) or -2 (no kernel built for this dtype/head-dim/nw).
int launch_fwd(const FwdK& p, int dtype_bf16, int d, int nw, hipStream_t stream);
int fwd_block_m(int nw);
// Software-pipelined forward (fa_fwd_il.hip); nw = 4 or 8 waves per workgroup.  No softcap / ALiBi variant.
int launch_fwd_il(const FwdK& p, int dtype_bf16, int d, int nw, hipStream_t stream);

// Backward: delta = rowsum(dO*O) pre-pass, dK/dV kernel (loops over query blocks),
// dQ kernel (loops over key blocks).  Same return convention.
int launch_bwd_delta(const BwdK& p, int dtype_bf16, int d, hipStream_t stream);
int launch_bwd_dkdv(const BwdK& p, int dtype_bf16, int d, hipStream_t stream);
int launch_bwd_dq(const BwdK& p, int dtype_bf16, int d, hipStream_t stream);
int bwd_block_m();   // query rows per dQ workgroup
int bwd_block_n();   // key rows per dK/dV workgroup

}  // namespace fa
