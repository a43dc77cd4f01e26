// dist.h -- communication layer for the row-partitioned solve (private).
// One process per GPU.  After each half-step the freshly written slice of the gathered vector has to
// reach the ranks whose SpMV reads it: either as one in-place all-gather, or -- when the shards only
// touch a fraction of the remote entries (banded / block-structured LPs) -- as a neighbour exchange
// of exactly the entries each rank's column indices name (HaloPlan, solver.h).  Reduction scalars
// take one small all-reduce per check.
#pragma once

#include <hip/hip_runtime.h>

#include <cstddef>

namespace hprlp {

// one peer of a grouped point-to-point exchange; either side may be empty
struct P2P {
    int peer;
    const void *send;
    size_t send_bytes;
    void *recv;
    size_t recv_bytes;
};

struct Comm {
    int rank = 0;
    int size = 1;
    virtual ~Comm() = default;
    // every rank contributes buf[rank*chunk .. (rank+1)*chunk) and receives the whole buf (size*chunk)
    virtual void allgather_inplace(double *buf, size_t chunk, hipStream_t s) = 0;
    virtual void allreduce_sum(double *buf, int count, hipStream_t s) = 0;
    // all sends and receives of the list progress together; rank a's send to b pairs with b's receive from a
    virtual void exchange(const P2P *ops, int nops, hipStream_t s) = 0;
    // true if the calls above block the host until the data has moved (the in-process group); RCCL only enqueues
    virtual bool host_blocking() const { return false; }
    // what the transport itself reports (RCCL: ncclCommCount / ncclCommUserRank / ncclCommCuDevice); the in-process group
    // answers from its own fields with device -1
    virtual int reported_ranks() const { return size; }
    virtual int reported_rank() const { return rank; }
    virtual int reported_device() const { return -1; }
};

// RCCL implementation; librccl.so is loaded at run time on first use so that a single-GPU process
// never needs it.  unique_id: the 128-byte ncclUniqueId produced by rank 0 (hprlp_dist_unique_id).
Comm *make_rccl_comm(int rank, int size, const void *unique_id, size_t id_bytes, int device);
void rccl_get_unique_id(void *out, size_t bytes);

// Host-staged group of PROCESSES on one node: a POSIX shared-memory segment named by the 128-byte id (hprlp_dist_unique_id
// under HPRLP_DIST_TRANSPORT=shm) carries a barrier, a table of posted sends and one pinned staging area per rank; a slice
// travels device -> own area -> the reader's device.  Needs neither RCCL nor a device IPC handle: the transport of last
// resort of bench.py's staged fallback, and the one multi-PROCESS form a one-GPU box can run (every rank on device 0).
// device < 0: the collectives' buffers are host memory (the protocol's own test on a box without a GPU).
bool is_shm_unique_id(const void *unique_id, size_t id_bytes);
void shm_make_unique_id(void *out, size_t bytes);
Comm *make_shm_comm(int rank, int size, const void *unique_id, size_t id_bytes, size_t area_bytes, int device);

// In-process implementation: `size` solver instances driven by `size` host threads of ONE process
// (all on the same GPU) exchange through device-to-device copies and host barriers.  It exists so
// that the multi-rank solver path (shards, halo plans, reductions) can be run for real on a one-GPU
// box (tests/test_gpu_dist.py); production multi-GPU runs use RCCL.
struct LocalGroup;
LocalGroup *make_local_group(int size);
void free_local_group(LocalGroup *g);
Comm *make_local_comm(LocalGroup *g, int rank);

}  // namespace hprlp
