// tests/simt/rccl/rccl.h -- TEST INFRASTRUCTURE: the single-rank stand-in for <rccl/rccl.h> of the SIMT emulation build
// (tests/simt/hip/hip_runtime.h).  A communicator of one rank: all-reduce and all-gather are copies; more ranks are refused.
#pragma once
#include <cstring>
#include <hip/hip_runtime.h>
typedef enum { ncclSuccess = 0, ncclUnhandledCudaError = 1, ncclInvalidArgument = 4 } ncclResult_t;
typedef enum { ncclDouble = 8 } ncclDataType_t;
typedef enum { ncclSum = 0 } ncclRedOp_t;
#define NCCL_UNIQUE_ID_BYTES 128
typedef struct { char internal[NCCL_UNIQUE_ID_BYTES]; } ncclUniqueId;
typedef struct simtNcclComm { int nranks; } *ncclComm_t;
inline const char *ncclGetErrorString(ncclResult_t r) { return r == ncclSuccess ? "ncclSuccess" : "simt rccl stub: one rank only"; }
inline ncclResult_t ncclGetUniqueId(ncclUniqueId *id) { memset(id, 0, sizeof(*id)); return ncclSuccess; }
inline ncclResult_t ncclCommInitRank(ncclComm_t *c, int nranks, ncclUniqueId, int rank) {
    if (nranks != 1 || rank != 0) return ncclInvalidArgument;
    *c = new simtNcclComm{1};
    return ncclSuccess;
}
inline ncclResult_t ncclCommDestroy(ncclComm_t c) { delete c; return ncclSuccess; }
inline ncclResult_t ncclAllReduce(const void *s, void *d, size_t n, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) { if (s != d) memmove(d, s, n * 8); return ncclSuccess; }
inline ncclResult_t ncclAllGather(const void *s, void *d, size_t n, ncclDataType_t, ncclComm_t, hipStream_t) { if (s != d) memmove(d, s, n * 8); return ncclSuccess; }
