// The transform kernels (kernels_ntt.cuh) as their own translation unit: compiled with the max-ILP machine scheduler, see that header.
#define SBN_NTT_KERNELS_HERE
#include "kernels_ntt.cuh"
template __global__ void ntt_fast_pass_kernel<0>(NttPassParams p, u32 kperm);
template __global__ void ntt_fast_pass_kernel<1>(NttPassParams p, u32 kperm);
