// grid_cus.hpp -- how many compute units the PERSISTENT grids (one workgroup per CU: stream-K convolutions, split weight gradients,
// stem / 1x1 gradients) are sized for: the device's CU count minus the CUs reserved with sc_set_reserved_cus (device.hip).
// Why: in a multi-GPU step RCCL's all-reduce kernels need CUs of their own; a chip-filling persistent grid leaves them none until
// its workgroups retire (`--hip.reserve_cus N`).  0 reserved (the default) = every CU.
#pragma once
namespace sc {
int grid_cus();
}
