// kmc_device.h — gfx950 device code of the model checker: the Next-state relations and
// invariants of the Kafka replication specs lowered onto the bit-packed state vector of
// kmc_layout.h, and the BFS level kernels.  Written for wave64 CDNA4 only.
//
// Specialisation: everything is a template over one (model, N, L, R, E[, K]); the host
// engine instantiates exactly one configuration per code object with KMC_INSTANTIATE (via
// hiprtc), so every bit offset, loop bound and action-instance index below is a compile-time
// constant and a packed state lives in W 64-bit registers per lane.
//
// Kernel shape (k_expand): one wavefront lane per frontier state.  One kernel PER MODE (the search's own,
// the level-step interface's owner bucketing, the enumerator: kmc_expand_body<M, MODE>, three code objects per configuration).
//   * coalesced loads of the SoA frontier planes (plane k, state i at fin[k*stride+i]); every
//     field is extracted once; the invariants of the state being expanded are checked here;
//   * pass 1: the guards of all action instances of `Next` (every binding of the specs' \E over
//     replicas / requests) in one straight-line VALU-domain block -> per-lane "enabled" bitset;
//   * pass 2, two forms.  KIND-MAJOR (Kafka models on the replica-major layout of kmc_layout.h: the headline): a
//     wave-uniform walk over the action KINDS; in every leaf each lane applies ITS OWN next enabled binding of that kind
//     (KmcKafka::apply<K>: replicas / request epoch are per-lane run-time values, a field of replica r is "select word r,
//     extract at a compile-time offset") until no lane has one left — 12 leaves per 64-state tile at the headline.
//     INSTANCE-MAJOR (every other model and layout): a walk over the instances; a scalar binary dispatch jumps to the
//     statically specialised effect of each instance some lane enabled (30 leaves per tile at the headline's constants).
//     Under orbit counting at seven brokers the kind-major walk runs with FULL leaves: a tile's (lane, binding) pairs of a
//     kind dealt out 64 to a leaf, every lane applying one pair to its source lane's state (kmc_pull).
//     Either way lanes never diverge on *which* action they apply, and enabled successors are compacted with
//     __ballot + mbcnt prefix ranks into a per-wave LDS ring (SoA, conflict-free): the write-combining stage;
//   * a flush drains exactly 64 successors, one per lane, so the random HBM probes of the
//     fingerprint table always run with a full wave of independent requests in flight:
//     64-bit mix hash -> open-addressed linear probe -> atomicCAS(0 -> fp) claim;
//   * winners wait in a second LDS ring and are appended to the next frontier 64 at a time:
//     one atomicAdd on a per-segment counter and W fully coalesced 512-byte plane stores.
// The seen-set replaces tlc2.tool.fp.FPSet, the frontier arrays replace
// tlc2.tool.queue.StateQueue and this loop replaces tlc2.tool.Worker.run [TLC-recall; TLC is
// not part of /root/reference].
//
// The source is split into parts (each cites the spec text it lowers): kmc_common.h, kmc_models_small.h, kmc_kafka.h,
// kmc_symm.h, kmc_sink.h, kmc_kernels.h.  gen_sources.py embeds them in this order; kmc_engine_codeobj.cpp concatenates them for hiprtc.
#pragma once
#include "kmc_common.h"
#include "kmc_models_small.h"
#include "kmc_kafka.h"
#include "kmc_symm.h"
#include "kmc_sink.h"
#include "kmc_kernels.h"
