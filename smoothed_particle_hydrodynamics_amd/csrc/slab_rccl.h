// Native neighbour exchange of a slab: RCCL point-to-point calls issued by this library itself.
//
// What slab.py does with torch.distributed, without Python in the loop: a C++ host (or a thin
// launcher) creates one context per GPU, hands every rank the same ncclUniqueId, and calls
// sph_hip_slab_comm_run().  Each slab talks to its two neighbours only (rank - 1, rank + 1): one
// xGMI link per pair, ncclSend/ncclRecv of both directions in one group on a high-priority
// exchange stream, overlapped with the interior's acceleration exactly as
// sph_hip_slab_step_begin/_end arrange it.  librccl is opened with dlopen on first use, so a
// single-GPU user of libsph_hip.so carries no dependency on it.
#pragma once

#include <dlfcn.h>
#include <rccl/rccl.h>

#include "sph_device.h"

struct RcclApi {
   void* lib = nullptr;
   ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
   ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
   ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
   ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
   ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
   ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t,
                             hipStream_t) = nullptr;
   ncclResult_t (*GroupStart)() = nullptr;
   ncclResult_t (*GroupEnd)() = nullptr;
   const char* (*GetErrorString)(ncclResult_t) = nullptr;
};

// nullptr + message in *why when librccl or one of its symbols is missing
inline const RcclApi* rccl_api(std::string* why)
{
   static RcclApi api;
   static bool tried = false;
   static std::string failure;
   if (!tried) {
      tried = true;
      // SPH_HIP_RCCL_LIBRARY names the one library to try instead of the usual places (a site
      // with its own RCCL build; the tests use it to reach the not-found branch)
      const char* forced = getenv("SPH_HIP_RCCL_LIBRARY");
      std::string last_error = "?";
      for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
         (void)dlerror();
         api.lib = dlopen(forced ? forced : name, RTLD_NOW | RTLD_GLOBAL);
         if (api.lib) break;
         // dlerror() clears the message it returns: call it once per failure
         const char* m = dlerror();
         if (m) last_error = m;
         if (forced) break;
      }
      if (!api.lib) {
         failure = std::string("cannot open librccl: ") + last_error;
      } else {
#define SPH_RCCL_SYM(field, name)                                                   \
   if (failure.empty()) {                                                          \
      *reinterpret_cast<void**>(&api.field) = dlsym(api.lib, name);                 \
      if (!api.field) failure = std::string("librccl lacks ") + name;               \
   }
         SPH_RCCL_SYM(GetUniqueId, "ncclGetUniqueId")
         SPH_RCCL_SYM(CommInitRank, "ncclCommInitRank")
         SPH_RCCL_SYM(CommDestroy, "ncclCommDestroy")
         SPH_RCCL_SYM(Send, "ncclSend")
         SPH_RCCL_SYM(Recv, "ncclRecv")
         SPH_RCCL_SYM(AllReduce, "ncclAllReduce")
         SPH_RCCL_SYM(GroupStart, "ncclGroupStart")
         SPH_RCCL_SYM(GroupEnd, "ncclGroupEnd")
         SPH_RCCL_SYM(GetErrorString, "ncclGetErrorString")
#undef SPH_RCCL_SYM
      }
   }
   if (!failure.empty()) {
      if (why) *why = failure;
      return nullptr;
   }
   return &api;
}

// One slab's communicator, exchange stream and message buffers.
struct SlabComm {
   ncclComm_t comm = nullptr;
   int rank = 0, nranks = 1;
   hipStream_t stream = nullptr;      // exchange stream (high priority)
   hipEvent_t packed = nullptr;       // main stream -> exchange stream (serial exchange only)
   hipEvent_t arrived = nullptr;      // exchange stream -> main stream
   void* send_left = nullptr;
   void* send_right = nullptr;
   void* recv_left = nullptr;
   void* recv_right = nullptr;
   int capacity_records = 0;          // what the buffers hold
   int active_records = 0;            // what messages are packed for and transferred with (<= capacity)
   size_t bytes = 0;                  // bytes of a message of active_records
   int32_t* trim_word = nullptr;      // device int: this rank's wish, then the maximum over the ranks
   bool primed = false;               // the first ghosts have been delivered
   // trimmed messages grow BEFORE they overflow: every GROW_EVERY steps of sph_hip_slab_comm_run the
   // ranks reduce (max) the record counts of the messages they packed last, the result travels to
   // pinned host memory behind the exchange, and the look at it - GROW_EVERY steps later, at the same
   // step on every rank, because every rank holds the same number - decides for all of them alike
   long long steps_run = 0;           // steps enqueued by sph_hip_slab_comm_run so far
   int32_t* fill_word = nullptr;      // device int: max records of this rank's two messages, then over the ranks
   int32_t* fill_host = nullptr;      // pinned copy of it
   hipEvent_t fill_arrived = nullptr;
   bool fill_pending = false;
   int growths = 0;                   // times the messages went back to capacity_records
};
#define SLAB_GROW_EVERY 16
#define SLAB_GROW_FILL_NUM 4          // grow when a message is more than 4/5 full
#define SLAB_GROW_FILL_DEN 5
