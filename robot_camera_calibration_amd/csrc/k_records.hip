// k_records.hip -- the per-batch table of pose records that ranks exchange: what one process of the
// reference hands the next over the "tag_detections" topic (real_preprocessing/src/corner_detections.cpp:41-56,78: id,
// size, the FOUR pixel corners) plus the pose of camera_pose.cpp:163-164, as RCC_REC_DOUBLES doubles per slot, packed
// on the device straight from the detector's record buffer (no host round trip before the all-gather).
//   slot = frame * targets_per_frame + q;   a slot without a target is all zeros
//   [0] valid  [1] global frame index  [2] id  [3] ncorners  [4..6] rvec  [7..9] tvec  [10] rms  [11..18] corners bl,br,tr,tl (x,y)
#include "rcc_internal.h"

__global__ __launch_bounds__(256) void k_pack_records(const rcc_detection* __restrict__ det, const int32_t* __restrict__ ndet, int nframes,
                                                      int tpf, int det_stride, int frame_offset, int capacity, double* __restrict__ table)
{
  const int slot = blockIdx.x * 256 + threadIdx.x;
  if (slot >= capacity) return;
  const int f = slot / tpf, q = slot - f * tpf;
  double* o = table + (size_t)slot * RCC_REC_DOUBLES;
  // slots beyond this batch (a batch shorter than the table: the ragged last one) are cleared, so that a gather of the
  // whole table never hands on the previous batch's records
  const int n = f < nframes ? min(ndet[f], tpf) : 0;
  if (q >= n) {
#pragma unroll
    for (int i = 0; i < RCC_REC_DOUBLES; ++i) o[i] = 0.0;
    return;
  }
  const rcc_detection& d = det[(size_t)f * det_stride + q];
  o[0] = 1.0; o[1] = (double)(f + frame_offset); o[2] = (double)d.id; o[3] = (double)d.ncorners;
  o[4] = d.rvec[0]; o[5] = d.rvec[1]; o[6] = d.rvec[2];
  o[7] = d.tvec[0]; o[8] = d.tvec[1]; o[9] = d.tvec[2];
  o[10] = d.rms;
#pragma unroll
  for (int c = 0; c < 4; ++c) { o[11 + 2 * c] = d.corners[c][0]; o[12 + 2 * c] = d.corners[c][1]; }
}

hipError_t rcc_launch_pack_records(rcc_handle* h, int nframes, int frame_offset, double* d_table, hipStream_t s)
{
  const bool fid = h->cfg.target_kind == RCC_TARGET_FIDUCIAL;
  const int tpf = fid ? h->cfg.max_targets : 1;
  const int n = h->rec_capacity;            // >= nframes * tpf: checked where the batch is accepted
  if (n <= 0) return hipSuccess;
  hipLaunchKernelGGL(k_pack_records, dim3((n + 255) / 256), dim3(256), 0, s, h->d_det, h->d_ndet, nframes, tpf, tpf, frame_offset, n, d_table);
  return hipGetLastError();
}
