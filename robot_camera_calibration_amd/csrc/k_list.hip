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
#define LIST_MAXH 4096  // the bucket path needs one 16-bit counter per image row

// BUCKETS = 1 (height <= LIST_MAXH): counting sort by row + per-row ranking by x, and suppression scans only
// the rows within the radius -- O(n * candidates per row band) instead of two O(n^2) passes (the order and the
// survivors are the same: the key (y, x) is unique, and a suppressor lies within nms_radius rows by definition).
template <int BUCKETS>
__global__ __launch_bounds__(256) void k_list_sort_nms(const rcc_cand* __restrict__ cand,
                                                       const int32_t* __restrict__ cand_count, int cap, int height,
                                                       int nms_radius, int max_kept, int kstride,
                                                       rcc_cand* __restrict__ pre, int32_t* __restrict__ npre,
                                                       rcc_frame_corners* __restrict__ fc)
{
  // LDS sized by the launch for this configuration's list capacity and image height (list_lds_bytes): two lists of `cap`
  // entries and, for the bucket path, two 16-bit tables of height + 2 rows.  (Sized for the largest configuration --
  // 4096 entries, 4096 rows: 84 KB -- a CU held one workgroup; the board's 2048 x 1080 needs 37 KB: four.)
  extern __shared__ __attribute__((aligned(16))) uint8_t list_lds[];
  const int rows_al = BUCKETS ? ((height + 2 + 7) & ~7) : 8;
  rcc_cand* const raw = reinterpret_cast<rcc_cand*>(list_lds);
  rcc_cand* const srt = raw + cap;
  unsigned short* const s_start = reinterpret_cast<unsigned short*>(srt + cap);   // first sorted index of each row (after the scan)
  unsigned short* const s_fill = s_start + rows_al;                               // per-row counters
  int* const s_cnt = reinterpret_cast<int*>(s_fill + rows_al);                     // 256
  int* const s_off = s_cnt + 256;                                                  // 257
  const int f = blockIdx.x;
  const int tid = threadIdx.x;
  const int count = cand_count[f];
  rcc_frame_corners* out = fc + f;
  // the frame's table starts from zero (a frame without a target then reads the same on every handle and in every call: the
  // later kernels write the counts and, for a found board, its corners -- nothing behind ncorners)
  static_assert(sizeof(rcc_frame_corners) % 4 == 0, "frame table in dwords");
  for (int i = tid; i < (int)(sizeof(rcc_frame_corners) / 4); i += 256) reinterpret_cast<uint32_t*>(out)[i] = 0u;
  __syncthreads();
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
  if (BUCKETS) {
    for (int y = tid; y <= height; y += 256) { s_fill[y] = 0; }
    __syncthreads();
    for (int i = tid; i < n; i += 256) {
      const rcc_cand e = cand[(size_t)f * cap + i];
      raw[i] = e;
      atomicAdd(reinterpret_cast<unsigned*>(s_fill) + ((unsigned)e.y >> 1), ((unsigned)e.y & 1u) ? 0x10000u : 1u);   // 16-bit counter (n <= 4096: no carry)
    }
    __syncthreads();
    // exclusive scan of the row counts: each thread scans a contiguous chunk of rows, then the chunk totals
    const int rows = height + 1;
    const int chunk = (rows + 255) / 256;
    const int y0 = min(tid * chunk, rows), y1 = min(y0 + chunk, rows);
    int acc = 0;
    for (int y = y0; y < y1; ++y) acc += s_fill[y];
    s_cnt[tid] = acc;
    __syncthreads();
    if (tid == 0) {
      int a = 0;
      for (int t = 0; t < 256; ++t) { s_off[t] = a; a += s_cnt[t]; }
      s_off[256] = a;
    }
    __syncthreads();
    acc = s_off[tid];
    for (int y = y0; y < y1; ++y) { const int c = s_fill[y]; s_start[y] = (unsigned short)acc; acc += c; }
    if (tid == 255) s_start[rows] = (unsigned short)n;
    __syncthreads();
    for (int y = tid; y <= height; y += 256) s_fill[y] = 0;
    __syncthreads();
    // scatter into row buckets (any order inside a bucket) ...
    for (int i = tid; i < n; i += 256) {
      const rcc_cand e = raw[i];
      const unsigned y = (unsigned)e.y;
      const unsigned old = atomicAdd(reinterpret_cast<unsigned*>(s_fill) + (y >> 1), (y & 1u) ? 0x10000u : 1u);
      const unsigned k = (y & 1u) ? (old >> 16) : (old & 0xFFFFu);
      srt[s_start[y] + k] = e;
    }
    __syncthreads();
    // ... then rank inside the bucket by x (unique within a row)
    for (int i = tid; i < n; i += 256) {
      const rcc_cand e = srt[i];
      const int b0 = s_start[e.y], b1 = s_start[e.y + 1];
      int rank = b0;
      for (int j = b0; j < b1; ++j) rank += (srt[j].x < e.x);
      raw[rank] = e;
    }
    __syncthreads();
    for (int i = tid; i < n; i += 256) srt[i] = raw[i];
    __syncthreads();
  } else {
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
  }
  // suppression; each thread owns a contiguous chunk so the output keeps the order
  const int chunk = (n + 255) / 256;
  const int i0 = min(tid * chunk, n), i1 = min(i0 + chunk, n);
  int mykeep = 0;
  // reuse raw[] as the keep flags (x field)
  for (int i = i0; i < i1; ++i) {
    const rcc_cand e = srt[i];
    bool keep = true;
    int j0 = 0, j1 = n;
    if (BUCKETS) {
      j0 = s_start[max((int)e.y - nms_radius, 0)];
      j1 = s_start[min((int)e.y + nms_radius, height) + 1];
    }
    for (int j = j0; j < j1 && keep; ++j) {
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
  // capacity of the list after suppression: tag scenes cfg.max_kept (<= 2048); board scenes the full 2048 -- cfg.max_kept (<= 256)
  // bounds the validated list of a4.3 there (k_validate)
  int max_kept = (c.target_kind == RCC_TARGET_FIDUCIAL && c.max_kept < h->kept_cap) ? c.max_kept : h->kept_cap;
  const bool buckets = c.height <= LIST_MAXH - 1;
  const size_t lds = 2 * (size_t)c.max_candidates * sizeof(rcc_cand) + 2 * (size_t)(buckets ? ((c.height + 2 + 7) & ~7) : 8) * sizeof(unsigned short) +
                     (256 + 257) * sizeof(int);
  if (lds > 64 * 1024) {
    // beyond the default limit of dynamic LDS: raise it for the instantiation about to be launched (idempotent)
    hipError_t ea = buckets ? hipFuncSetAttribute(reinterpret_cast<const void*>(&k_list_sort_nms<1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)
                            : hipFuncSetAttribute(reinterpret_cast<const void*>(&k_list_sort_nms<0>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (ea != hipSuccess) return ea;
  }
  if (buckets)
    hipLaunchKernelGGL(k_list_sort_nms<1>, dim3(nframes), dim3(256), lds, s, d_cand, d_cand_count,
                       c.max_candidates, c.height, c.nms_radius, max_kept, h->kept_cap, h->d_pre, h->d_npre, h->d_fc);
  else
    hipLaunchKernelGGL(k_list_sort_nms<0>, dim3(nframes), dim3(256), lds, s, d_cand, d_cand_count,
                       c.max_candidates, c.height, c.nms_radius, max_kept, h->kept_cap, h->d_pre, h->d_npre, h->d_fc);
  return hipGetLastError();
}
