// Declarations only (csrc/hjgpu_multi.hip binds librccl with dlopen and takes decltype(&ncclX) of these): the CPU test of the
// orchestration (tests/cpp_pipeline_ordering.cpp) drives the loopback transport, RCCL is never called there.
#pragma once
#include <stddef.h>
#include "../hip/hip_runtime.h"
typedef enum { ncclSuccess = 0, ncclUnhandledCudaError = 1, ncclSystemError = 2, ncclInternalError = 3, ncclInvalidArgument = 4,
               ncclInvalidUsage = 5, ncclRemoteError = 6, ncclInProgress = 7 } ncclResult_t;
typedef struct ncclComm *ncclComm_t;
typedef struct { char internal[128]; } ncclUniqueId;
typedef enum { ncclInt8 = 0, ncclUint8 = 1, ncclInt32 = 2, ncclUint32 = 3, ncclInt64 = 4, ncclUint64 = 5 } ncclDataType_t;
typedef enum { ncclSum = 0 } ncclRedOp_t;
ncclResult_t ncclGetVersion(int *);
ncclResult_t ncclGetUniqueId(ncclUniqueId *);
ncclResult_t ncclCommInitAll(ncclComm_t *, int, const int *);
ncclResult_t ncclCommInitRank(ncclComm_t *, int, ncclUniqueId, int);
ncclResult_t ncclCommDestroy(ncclComm_t);
ncclResult_t ncclCommAbort(ncclComm_t);
ncclResult_t ncclCommGetAsyncError(ncclComm_t, ncclResult_t *);
ncclResult_t ncclCommCount(const ncclComm_t, int *);
ncclResult_t ncclCommUserRank(const ncclComm_t, int *);
ncclResult_t ncclCommCuDevice(const ncclComm_t, int *);
ncclResult_t ncclGroupStart();
ncclResult_t ncclGroupEnd();
ncclResult_t ncclSend(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t);
ncclResult_t ncclRecv(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t);
ncclResult_t ncclAllGather(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t);
ncclResult_t ncclBroadcast(const void *, void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t);
ncclResult_t ncclAllReduce(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t);
const char *ncclGetErrorString(ncclResult_t);
