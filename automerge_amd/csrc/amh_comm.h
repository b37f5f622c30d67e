// RCCL behind the C ABI: the one exchange step of the path (all-gather of sync summaries) runs on
// device buffers over xGMI, with no trip through host memory.  librccl is opened on first use (dlopen), so a
// single-GPU process never loads it.
#pragma once
#include <stddef.h>
#include <stdint.h>

struct AmhComm;
// id: 128 bytes (ncclUniqueId) made on rank 0 and handed to every rank out of band
const char* amh_comm_make_id(uint8_t* id128);
AmhComm* amh_comm_open(const uint8_t* id128, int rank, int world, int device, const char** err);
void amh_comm_close(AmhComm*);
int amh_comm_rank(const AmhComm*);
int amh_comm_world(const AmhComm*);
// every rank contributes one u64; out[world] on the device
const char* amh_comm_allgather_u64(AmhComm*, const uint64_t* d_mine, uint64_t* d_all, void* stream);
// all-gather(v): rank r contributes sizes[r] bytes at d_mine; d_all + offs[r] receives them on every rank
const char* amh_comm_allgatherv(AmhComm*, const uint8_t* d_mine, uint8_t* d_all, const uint64_t* sizes, const uint64_t* offs, void* stream);
