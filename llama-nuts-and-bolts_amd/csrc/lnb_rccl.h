// lnb_rccl.h -- the handful of RCCL entry points the pipeline uses, as a table filled by dlopen (lnb_pipeline.cpp).
// Signatures follow /opt/rocm/include/rccl/rccl.h (ncclUniqueId is a 128-byte struct passed BY VALUE to ncclCommInitRank).
#pragma once
#include <stddef.h>
struct lnb_nccl_id { char internal[128]; };
enum { LNB_NCCL_INT8 = 0, LNB_NCCL_INT32 = 2 };              // ncclInt8 / ncclInt32
struct lnb_rccl_api {
    int (*GetUniqueId)(lnb_nccl_id*);
    int (*CommInitRank)(void** comm, int nranks, lnb_nccl_id id, int rank);
    int (*CommDestroy)(void* comm);
    int (*GroupStart)(void);
    int (*GroupEnd)(void);
    int (*Send)(const void* buf, size_t count, int dtype, int peer, void* comm, void* stream);
    int (*Recv)(void* buf, size_t count, int dtype, int peer, void* comm, void* stream);
    const char* (*GetErrorString)(int);
    int (*GetVersion)(int*);
    int (*CommCount)(const void* comm, int* count);         // ncclCommCount: the communicator's own idea of how many ranks it spans
};
extern "C" const lnb_rccl_api* lnb_rccl_load(void);         // nullptr + lnb_last_error() when the library cannot be loaded
