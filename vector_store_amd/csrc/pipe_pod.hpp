// pipe_pod.hpp -- PODS: the pipelined walk (pipe_device.hpp / kernels_pipe.hip) as a resident kernel that callers hand queries to.
//
// The reference issues one query per FFI call (crates/vector-store/src/vs_index/usearch.rs:212, :236) and puts every filtered query
// on a blocking thread of its own (:937-948): a crowd of callers, each with ONE walk in flight, each walk a workgroup on one CU.  One
// launch per caller and round caps the crowd at the process's hardware queues (16 walks); batched launches hold their stream until the
// slowest walk of the batch ends.  A pod is one launch of `n` workgroups that stay: workgroup b polls slot b of a table in pinned host
// memory, answers the query the host posts there (the same PipeQuery a batched launch reads from its table: the kernel writes the
// answer and the flag into the caller's pinned block) and polls again -- until the host closes the pod.  The callers are then served
// by as many walks at a time as there are callers, and a query costs no launch at all.
//
// A pod is launched with the index's arenas and layouts as kernel arguments, like every other launch: whatever moves an arena (reserve)
// closes the index's pods first.  What adds and removes change -- entry point, top level, "has removed members" -- is read per query
// from PodCtl, so a pod survives them (they never overlap searches: usearch.rs:590-612; the host freezes the pod while it modifies).
// An idle pod closes after a few milliseconds: its workgroups hold their CUs while they poll.
#pragma once
#include <cstddef>

#include "kernels.hpp"

namespace vs {

struct alignas(128) PodSlot {  // pinned host memory: one per workgroup of a pod
    uint32_t posted;           // host: 1, 2, 3, ...: the number of the query in `q`, stored last (release)
    uint32_t ef;               // its beam
    uint32_t left;             // device: 1 = this slot's workgroup has left (the pod was closed, or its host went quiet)
    uint32_t explore;          // a pod of filtered queries: 1 = an exploring round, 0 = the exact walk, 2 = the walk that asks while it runs (round 6)
    PipeQuery q;
};
static_assert(sizeof(PodSlot) == 128, "one slot, one 128-byte line");

struct alignas(64) PodCtl {    // pinned host memory: one per pod
    uint32_t closed;           // host: 1 = workgroups leave as soon as they are idle
    uint32_t heartbeat;        // host: advances while the pod is open; workgroups of a pod whose host has gone quiet for seconds leave by themselves
    // The part of the index's view that adds and removes change (round 5): read by a workgroup for every query it takes, behind the
    // acquire that follows the query's number -- so a pod stays open across modifications (the reference alternates families of adds
    // and searches, usearch.rs:590-612: closing and re-launching 64 workgroups per family was a tenth of a millisecond each way).  The
    // arenas' addresses, the row layout and the workspace layout (laid out for the index's CAPACITY) still travel as kernel arguments:
    // what moves an arena closes the pods.  Written by the host while no query of the pod is in flight.
    uint32_t entry_slot;
    int32_t max_level;
    uint32_t has_removed;
    uint32_t pad[11];
};

// `slots` == nullptr: a plain launch (a.nq queries, one workgroup each).  Else a.nq workgroups that serve slots[blockIdx.x] until ctl->closed.
template <int AR> hipError_t launch_pipe_pod_ar(const WalkArgs& a, uint32_t iters, hipStream_t s, PodSlot* slots, PodCtl* ctl);
hipError_t launch_pipe_pod(const WalkArgs& a, uint32_t iters, hipStream_t s, PodSlot* slots, PodCtl* ctl);
// Round 6: the usearch-order TEAM walk as a pod (kernels_walk.hip; b1 storage): a.nq workgroups of kSearchTeam waves, a.ef = the instance's beam cap.
template <int AR> hipError_t launch_walk_pod_ar(const WalkArgs& a, uint32_t iters, hipStream_t s, PodSlot* slots, PodCtl* ctl);
hipError_t launch_walk_pod(const WalkArgs& a, uint32_t iters, hipStream_t s, PodSlot* slots, PodCtl* ctl);

}  // namespace vs
