// dist.h -- collective layer for the row-partitioned solve (private).
// One process per GPU; vectors that the other ranks' SpMV gathers from are exchanged with one
// all-gather per half-step, reduction scalars with one small all-reduce per check.
#pragma once

#include <hip/hip_runtime.h>

#include <cstddef>

namespace hprlp {

struct Comm {
    int rank = 0;
    int size = 1;
    virtual ~Comm() = default;
    // every rank contributes buf[rank*chunk .. (rank+1)*chunk) and receives the whole buf (size*chunk)
    virtual void allgather_inplace(double *buf, size_t chunk, hipStream_t s) = 0;
    virtual void allreduce_sum(double *buf, int count, hipStream_t s) = 0;
};

// RCCL implementation; librccl.so is loaded at run time on first use so that a single-GPU process
// never needs it.  unique_id: the 128-byte ncclUniqueId produced by rank 0 (hprlp_dist_unique_id).
Comm *make_rccl_comm(int rank, int size, const void *unique_id, size_t id_bytes, int device);
void rccl_get_unique_id(void *out, size_t bytes);

}  // namespace hprlp
