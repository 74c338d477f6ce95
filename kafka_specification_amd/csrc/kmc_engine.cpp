// kmc_engine.cpp — host side of the MI355X model checker behind the C ABI of include/kmc.h.
//
// Owns: specialising the device code for one (model, constants) through hiprtc (with an
// on-disk code-object cache), the HBM fingerprint table / frontier buffers, and the
// level-synchronous BFS loop that stands in for TLC's ModelChecker + Worker threads
// [TLC-recall; TLC is not part of /root/reference].  No CPU fallback exists: without a HIP
// device or compiler every entry point fails with KMC_E_DEVICE / KMC_E_COMPILE.
#include <hip/hip_runtime.h>
#include <hip/hiprtc.h>
#include <rccl/rccl.h>   // types and prototypes only: librccl is bound with dlopen when a communicator is created

#include <dlfcn.h>
#include <sys/stat.h>
#include <unistd.h>

#include <chrono>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/kmc.h"
#include "kmc_device.h"   // host-visible parts: KmcArgs, KmcLevelCtl, layout, fingerprint
#include "kmc_sources.inc"  // generated: KMC_SRC_DEVICE (kmc_layout.h and the parts of kmc_device.h, as text)

// Levels kmc_run queues back to back before it waits (no progress callback): see run_levels.
#define KMC_CHAIN 32
#define KMC_CTL_SLOTS (3 + KMC_CHAIN)   // two alternating levels + one auxiliary + one per chained level

// KMC_VERIFY: both builds carry the fingerprint checksum (KMC_CHECKSUM, kmc_device.h); the second one differs in how it is
// compiled — optimisation level and a quarter of the occupancy target, i.e. another register allocation
#define KMC_VERIFY_PRIMARY_OPTIONS "-DKMC_CHECKSUM=1"
// The second build of the differential self-check: another optimisation level, a quarter of the occupancy target, the
// fingerprint checksum — and, for the Kafka models, ANOTHER LOWERING OF THE GUARDS: KmcKafka::guard<K> looped per kind over
// a run-time binding instead of the straight-line block of every instance's inst<I> (kmc_device.h, RUNTIME_GUARDS).
#define KMC_VERIFY_OPTIONS "-O1 -DKMC_MIN_WAVES=2 -DKMC_CHECKSUM=1 -DKMC_RT_GUARDS_MIN_INSTANCES=0 -DKMC_WITH_DRY=1"

#include "kmc_engine_codeobj.h"    // namespace { validate, get_code_object, ... }
#include "kmc_engine_core.h"       // struct kmc_handle; namespace { launch*, absorb, do_begin, ... }

extern "C" {

#include "kmc_engine_open.h"
#include "kmc_engine_run.h"
#include "kmc_engine_step.h"
#include "kmc_engine_exchange.h"

}  // extern "C"
