// GPU raster voting: the overlay step of the reference's post-stage,
// R:scripts/road_segmentation/determine_class.py:97-120 `get_weighted_scores` -- gpd.overlay(labels, predictions,
// how="intersection") followed by area(intersection) / area(label) -- in raster form on the tile grid: the detection masks are
// already on the device, bit-packed (rs_dets.masks layout), the (tile-clipped, :62-95) road labels are rasterised to the same
// layout, and the intersection area of every (label, detection) pair is a popcount of the AND of two bit rows.
// HBM-bound integer work: n_lab * n_det * h * ceil(w/8) bytes read per tile (L2-resident after the first label).
// One workgroup per (label, detection-chunk); 32-bit words; a shuffle + LDS reduction in a fixed order (exact integers anyway).
#include "common.h"

namespace {

__global__ __launch_bounds__(256) void mask_overlap_kernel(const uint8_t* __restrict__ det, int n_det, const uint8_t* __restrict__ lab,
                                                           int n_lab, int words /* per mask, 32-bit, masks padded to 4 bytes by the caller's layout check */,
                                                           int* __restrict__ inter, int* __restrict__ lab_area) {
  __shared__ int red[4];
  const int l = blockIdx.x, d = blockIdx.y;          // d == n_det: the label's own area
  const unsigned int* L = (const unsigned int*)lab + (long long)l * words;
  const unsigned int* D = d < n_det ? (const unsigned int*)det + (long long)d * words : nullptr;
  int c = 0;
  for (int i = threadIdx.x; i < words; i += 256) c += __popc(D ? (L[i] & D[i]) : L[i]);
  for (int off = 32; off > 0; off >>= 1) c += __shfl_down(c, off);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = c;
  __syncthreads();
  if (threadIdx.x == 0) {
    const int tot = red[0] + red[1] + red[2] + red[3];
    if (d < n_det) inter[(long long)l * n_det + d] = tot;
    else lab_area[l] = tot;
  }
}

}  // namespace

int launch_mask_overlap(const uint8_t* det, int n_det, const uint8_t* lab, int n_lab, int h, int w, int* inter, int* lab_area, hipStream_t s) {
  RS_CHECK(det && lab && inter && lab_area && n_det >= 0 && n_lab > 0 && h > 0 && w > 0, RS_ERR_ARG, "mask overlap: bad argument");
  const long long bytes = (long long)h * ((w + 7) / 8);
  RS_CHECK(bytes % 4 == 0, RS_ERR_UNSUPPORTED, "mask overlap: h * ceil(w/8) = %lld bytes per mask must be a multiple of 4", bytes);
  hipLaunchKernelGGL(mask_overlap_kernel, dim3(n_lab, n_det + 1), dim3(256), 0, s, det, n_det, lab, n_lab, (int)(bytes / 4), inter, lab_area);
  RS_HIP(hipGetLastError());
  return RS_OK;
}
