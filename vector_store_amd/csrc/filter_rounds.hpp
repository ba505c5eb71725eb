// Filtered search with a lazily evaluated predicate (engine.hip filtered_lazy): the per-round device <-> host exchange.
// A round's block on the device: unknown = [listed, consulted, 62 pad | list of slots] (WalkArgs::unknown_count / unknown_list).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>

namespace vs {

// counters (+ found), answer and list -> the caller's pinned, device-mapped block, written by a kernel: no copy engine on the path
hipError_t launch_export_round(const uint32_t* unknown, uint32_t cap, const uint64_t* d_k, const float* d_d, const uint32_t* d_f, uint32_t k,
                               uint32_t* h_cnt, uint32_t* h_list, uint64_t* h_k, float* h_d, hipStream_t s);

// the host's verdicts (verdict[i] = 0 / 1 for the i-th listed slot; pinned host memory is fine) -> the query's device-resident
// `known` / `allow` bitmaps; then the two counters of the block are zeroed for the next walk
hipError_t launch_apply_verdicts(uint32_t* unknown, const uint8_t* verdict, uint32_t m, uint32_t slots, uint32_t* allow, uint32_t* known,
                                 hipStream_t s);

// verdicts remembered across the queries of one filter (PipeQuery::memo): the slots whose member changed lose theirs
hipError_t launch_memo_forget(uint32_t* memo, uint32_t stride, const uint32_t* slots, uint32_t m, hipStream_t s);

}  // namespace vs
