/*
 * rcc_dist.h -- C ABI of the one collective of the multi-GPU path (librcc_dist.so): an RCCL all-gather of the per-batch
 * record tables of include/rcc.h (rcc_set_record_tables), for hosts that are not Python -- the C++ ROS node of
 * robot_camera_calibration_amd/host/ runs one process per GPU and uses this where the reference has the
 * "tag_detections" topic between processes (real_preprocessing/src/corner_detections.cpp:78).
 *
 * Frames are independent units: each rank detects its own batch, no collective touches pixels, and the records
 * (19 doubles per slot, ~156 KB per rank for 1024 frames) are exchanged ONCE per batch -- latency-bound on xGMI, so
 * one direct all-gather, never per frame.
 *
 * Bootstrap is the host's business: rank 0 obtains an id with rcc_dist_unique_id and passes its RCC_DIST_ID_BYTES
 * bytes to the other ranks by its own means (rosparam, a file, an environment variable); every rank then calls
 * rcc_dist_create.  Status codes are include/rcc.h's (RCC_OK, RCC_ERR_ARG, RCC_ERR_DEVICE).
 */
#ifndef RCC_DIST_H_
#define RCC_DIST_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RCC_DIST_ID_BYTES 128

typedef struct rcc_dist rcc_dist;

int  rcc_dist_unique_id(void* id /* out: RCC_DIST_ID_BYTES bytes */);
int  rcc_dist_create(int32_t rank, int32_t world, const void* id, int32_t device, rcc_dist** out);
void rcc_dist_destroy(rcc_dist* d);
int  rcc_dist_rank(const rcc_dist* d);
int  rcc_dist_world(const rcc_dist* d);
/* d_table: this rank's nslots x RCC_REC_DOUBLES doubles (device); d_all: world x nslots x RCC_REC_DOUBLES doubles
 * (device), rank r's table at offset r * nslots.  Asynchronous on `stream` (hipStream_t; NULL: the default stream). */
int  rcc_dist_allgather_records(rcc_dist* d, const double* d_table, int32_t nslots, double* d_all, void* stream);
const char* rcc_dist_last_error(const rcc_dist* d);
/* text of the failure of this thread's last rcc_dist_unique_id / rcc_dist_create (they have no handle to carry it); "" after a success */
const char* rcc_dist_last_create_error(void);

#ifdef __cplusplus
}
#endif
#endif /* RCC_DIST_H_ */
