// nos_mapexact.hip — launcher of the reference-exact NDT map statistics (mapexact_kernels.hpp).
// Its own translation unit because it is compiled with -ffp-contract=off (csrc/Makefile): the kernels reproduce the
// reference binary's rounding, so no multiply-add may be fused unless the source says fma().
#include "nos_internal.hpp"

#include "mapexact_kernels.hpp"

namespace nosd {

static_assert(kMapReferenceFmaMask == nos::mapexact::kReferenceFmaMask, "Settings::map_fma_mask default out of step");

// acc: scratch [V][12].  All pointers are device memory; nothing is synchronised here.
hipError_t launch_map_exact(const double* px, const double* py, const double* pz, const uint32_t* sorted_idx,
                            const uint32_t* seg_offset, const uint32_t* seg_count, uint32_t n_voxels, int fma_mask,
                            int eigen_version, double* acc, double* mean, double* sqrt_info, unsigned char* valid,
                            double* evals, double* evecs, uint32_t* first_idx, hipStream_t stream) {
  using namespace nos::mapexact;
  if (n_voxels == 0) return hipSuccess;
  const unsigned acc_blocks = (n_voxels + kAccWavesPerBlock - 1) / kAccWavesPerBlock;
  hipLaunchKernelGGL(voxel_accumulate_exact_kernel, dim3(acc_blocks), dim3(64 * kAccWavesPerBlock), 0, stream, px, py, pz,
                     sorted_idx, seg_offset, seg_count, n_voxels, fma_mask, acc);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return e;
  const FinalizeParams prm{5, 0.01, 0.01, fma_mask, eigen_version};
  hipLaunchKernelGGL(voxel_finalize_exact_kernel, dim3((n_voxels + 63) / 64), dim3(64), 0, stream, acc, sorted_idx,
                     seg_offset, seg_count, n_voxels, prm, mean, sqrt_info, valid, evals, evecs, first_idx);
  return hipGetLastError();
}

}  // namespace nosd
