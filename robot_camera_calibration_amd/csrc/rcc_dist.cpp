// rcc_dist.cpp -- include/rcc_dist.h over RCCL (ncclAllGather of the record tables; backend of xGMI on an MI355X node).
// A library of its own (librcc_dist.so) so that librcc_hip.so does not pull RCCL into single-GPU hosts.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <cstdio>
#include <cstring>
#include <new>
#include "../../include/rcc.h"
#include "../../include/rcc_dist.h"

static_assert(sizeof(ncclUniqueId) <= RCC_DIST_ID_BYTES, "RCC_DIST_ID_BYTES too small for ncclUniqueId");

struct rcc_dist {
  ncclComm_t comm;
  int rank, world, device;
  char err[256];
};

// why the last rcc_dist_unique_id / rcc_dist_create of this thread failed: there is no handle to carry the text yet
static thread_local char g_create_err[256] = "";

extern "C" {

const char* rcc_dist_last_create_error(void) { return g_create_err; }

int rcc_dist_unique_id(void* id)
{
  if (!id) return RCC_ERR_ARG;
  g_create_err[0] = 0;
  ncclUniqueId u;
  ncclResult_t r = ncclGetUniqueId(&u);
  if (r != ncclSuccess) { snprintf(g_create_err, sizeof(g_create_err), "ncclGetUniqueId: %s", ncclGetErrorString(r)); return RCC_ERR_DEVICE; }
  memset(id, 0, RCC_DIST_ID_BYTES);
  memcpy(id, &u, sizeof(u));
  return RCC_OK;
}

int rcc_dist_create(int32_t rank, int32_t world, const void* id, int32_t device, rcc_dist** out)
{
  if (!out) return RCC_ERR_ARG;
  *out = nullptr;
  if (!id || world < 1 || rank < 0 || rank >= world || device < 0) return RCC_ERR_ARG;
  g_create_err[0] = 0;
  int ndev = 0;
  hipError_t he = hipGetDeviceCount(&ndev);
  if (he != hipSuccess || device >= ndev) {
    snprintf(g_create_err, sizeof(g_create_err), "device %d of %d visible (%s)", device, ndev, hipGetErrorString(he));
    return RCC_ERR_DEVICE;
  }
  he = hipSetDevice(device);
  if (he != hipSuccess) { snprintf(g_create_err, sizeof(g_create_err), "hipSetDevice(%d): %s", device, hipGetErrorString(he)); return RCC_ERR_DEVICE; }
  rcc_dist* d = new (std::nothrow) rcc_dist();
  if (!d) return RCC_ERR_NOMEM;
  memset(d, 0, sizeof(*d));
  d->rank = rank; d->world = world; d->device = device;
  ncclUniqueId u;
  memcpy(&u, id, sizeof(u));
  ncclResult_t r = ncclCommInitRank(&d->comm, world, u, rank);
  if (r != ncclSuccess) {
    snprintf(g_create_err, sizeof(g_create_err), "ncclCommInitRank(rank %d of %d): %s", rank, world, ncclGetErrorString(r));
    delete d;
    return RCC_ERR_DEVICE;
  }
  *out = d;
  return RCC_OK;
}

void rcc_dist_destroy(rcc_dist* d)
{
  if (!d) return;
  (void)hipSetDevice(d->device);
  if (d->comm) (void)ncclCommDestroy(d->comm);
  delete d;
}

int rcc_dist_rank(const rcc_dist* d) { return d ? d->rank : RCC_ERR_ARG; }
int rcc_dist_world(const rcc_dist* d) { return d ? d->world : RCC_ERR_ARG; }
const char* rcc_dist_last_error(const rcc_dist* d) { return d ? d->err : ""; }

int rcc_dist_allgather_records(rcc_dist* d, const double* d_table, int32_t nslots, double* d_all, void* stream)
{
  if (!d || !d_table || !d_all || nslots < 0) return RCC_ERR_ARG;
  if (nslots == 0) return RCC_OK;
  if (hipSetDevice(d->device) != hipSuccess) return RCC_ERR_DEVICE;
  const size_t count = (size_t)nslots * RCC_REC_DOUBLES;
  ncclResult_t r = ncclAllGather(d_table, d_all, count, ncclDouble, d->comm, (hipStream_t)stream);
  if (r != ncclSuccess) {
    snprintf(d->err, sizeof(d->err), "ncclAllGather: %s", ncclGetErrorString(r));
    return RCC_ERR_DEVICE;
  }
  return RCC_OK;
}

}  // extern "C"
