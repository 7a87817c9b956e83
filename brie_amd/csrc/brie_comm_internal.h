// brie_comm_internal.h -- what brie_capi.hip needs from brie_comm.hip: the in-library all-reduce of a
// gene-sharded coupled fit runs on the HANDLE's stream, between the two phases of cell_finalize.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

struct brie_comm;

namespace brie {
// sum `n` floats in place over the ranks of `c`, enqueued on `stream` (no host synchronisation)
int comm_allreduce_sum_f32_async(brie_comm *c, float *dev, int64_t n, hipStream_t stream);
int comm_world(const brie_comm *c);
int comm_device(const brie_comm *c);
}  // namespace brie
