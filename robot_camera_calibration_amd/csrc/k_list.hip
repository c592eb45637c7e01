// k_list.hip -- stage a4.2: per-frame ordering of the dense pass's (unordered) candidate list by
// (y,x) and list-level radius suppression.  One 256-thread block per frame.
//
// The dense pass compacts with one atomic per wave, so arrival order is not deterministic; ranking
// by the unique key (y<<16 | x) restores the order the specification uses, which makes every later
// tie-break ("smaller index wins") reproducible.  Definition: DESIGN.md section 3 (a4.2): entry i
// survives iff no entry j within Chebyshev distance nms_radius has a larger score, or an equal
// score and a smaller index.
#include "rcc_internal.h"

#define LIST_MAX 4096   // max_candidates upper bound (LDS: 2 x 32 KiB)

__global__ __launch_bounds__(256) void k_list_sort_nms(const rcc_cand* __restrict__ cand,
                                                       const int32_t* __restrict__ cand_count, int cap,
                                                       int nms_radius, int max_kept, int kstride,
                                                       rcc_cand* __restrict__ pre, int32_t* __restrict__ npre,
                                                       rcc_frame_corners* __restrict__ fc)
{
  __shared__ rcc_cand raw[LIST_MAX];
  __shared__ rcc_cand srt[LIST_MAX];
  __shared__ int s_cnt[256];
  __shared__ int s_off[257];
  const int f = blockIdx.x;
  const int tid = threadIdx.x;
  const int count = cand_count[f];
  rcc_frame_corners* out = fc + f;
  if (tid == 0) {
    out->status = 0;
    out->ncand = count;
    out->nkept = 0;
    out->ncorners = 0;
  }
  if (count > cap) {
    if (tid == 0) { out->status = RCC_FRAME_CAND_OVERFLOW; npre[f] = 0; }
    return;
  }
  const int n = count;
  for (int i = tid; i < n; i += 256) raw[i] = cand[(size_t)f * cap + i];
  __syncthreads();
  // rank by key
  for (int i = tid; i < n; i += 256) {
    const rcc_cand e = raw[i];
    const uint32_t key = ((uint32_t)(uint16_t)e.y << 16) | (uint16_t)e.x;
    int rank = 0;
    for (int j = 0; j < n; ++j) {
      const rcc_cand o = raw[j];
      const uint32_t k2 = ((uint32_t)(uint16_t)o.y << 16) | (uint16_t)o.x;
      rank += (k2 < key);
    }
    srt[rank] = e;
  }
  __syncthreads();
  // suppression; each thread owns a contiguous chunk so the output keeps the order
  const int chunk = (n + 255) / 256;
  const int i0 = min(tid * chunk, n), i1 = min(i0 + chunk, n);
  int mykeep = 0;
  // reuse raw[] as the keep flags (x field)
  for (int i = i0; i < i1; ++i) {
    const rcc_cand e = srt[i];
    bool keep = true;
    for (int j = 0; j < n && keep; ++j) {
      if (j == i) continue;
      const rcc_cand o = srt[j];
      int dx = abs((int)o.x - (int)e.x), dy = abs((int)o.y - (int)e.y);
      if (dx <= nms_radius && dy <= nms_radius) {
        if (o.score > e.score || (o.score == e.score && j < i)) keep = false;
      }
    }
    raw[i].x = keep ? 1 : 0;
    mykeep += keep;
  }
  s_cnt[tid] = mykeep;
  __syncthreads();
  if (tid == 0) {
    int acc = 0;
    for (int t = 0; t < 256; ++t) { s_off[t] = acc; acc += s_cnt[t]; }
    s_off[256] = acc;
  }
  __syncthreads();
  const int total = s_off[256];
  if (total > max_kept) {
    if (tid == 0) { out->status = RCC_FRAME_KEPT_OVERFLOW; npre[f] = 0; }
    return;
  }
  int o = s_off[tid];
  for (int i = i0; i < i1; ++i)
    if (raw[i].x) pre[(size_t)f * kstride + o++] = srt[i];
  if (tid == 0) npre[f] = total;
}

hipError_t rcc_launch_list(rcc_handle* h, const rcc_cand* d_cand, const int32_t* d_cand_count,
                           int nframes, hipStream_t s)
{
  const rcc_config& c = h->cfg;
  if (nframes <= 0) return hipSuccess;
  int max_kept = c.max_kept < h->kept_cap ? c.max_kept : h->kept_cap;
  hipLaunchKernelGGL(k_list_sort_nms, dim3(nframes), dim3(256), 0, s, d_cand, d_cand_count,
                     c.max_candidates, c.nms_radius, max_kept, h->kept_cap, h->d_pre, h->d_npre, h->d_fc);
  return hipGetLastError();
}
