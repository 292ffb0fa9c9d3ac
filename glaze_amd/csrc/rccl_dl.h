// RCCL, loaded at run time.  The library is only needed once a renderer spans several GPUs of one process
// (glz_renderer_set_devices, SURVEY 8(e): one ncclComm per device via ncclCommInitAll, one ncclReduce(sum, float) of the
// RGBA32F accumulator per read-back), so libglaze_hip.so does not link it: a single-GPU host without RCCL still loads the
// library, and a process that already carries an RCCL (PyTorch bundles one under the same soname) shares that copy.
#pragma once
#include <dlfcn.h>
#include <stdlib.h>
#include <rccl/rccl.h>

#include <mutex>
#include <string>

namespace glz {

struct Rccl {
  decltype(&ncclCommInitAll) CommInitAll = nullptr;
  decltype(&ncclCommDestroy) CommDestroy = nullptr;
  decltype(&ncclReduce) Reduce = nullptr;
  decltype(&ncclSend) Send = nullptr;
  decltype(&ncclRecv) Recv = nullptr;
  decltype(&ncclGroupStart) GroupStart = nullptr;
  decltype(&ncclGroupEnd) GroupEnd = nullptr;
  decltype(&ncclGetErrorString) GetErrorString = nullptr;
  decltype(&ncclGetVersion) GetVersion = nullptr;
  std::string error;   // why loading failed

  // nullptr (and `why` filled) when the library or one of its symbols is missing
  static const Rccl* get(std::string& why) {
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
      void* h = nullptr;
      std::string why = "librccl.so.1 not found";
      // GLAZE_RCCL_LIBRARY names the library to load instead (tests/fake_rccl: a recording stand-in that lets the n >= 2 group
      // construction run on a box with one GPU or none)
      const char* forced = getenv("GLAZE_RCCL_LIBRARY");
      for (const char* name : {forced ? forced : "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
        if (forced && name != forced) break;
        h = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
        if (h) break;
        if (const char* e = dlerror()) why = e;   // dlerror() hands its message out once
      }
      if (!h) {
        r.error = "RCCL is not available: " + why;
        return;
      }
      auto sym = [&](const char* n) -> void* {
        void* p = dlsym(h, n);
        if (!p && r.error.empty()) r.error = std::string("RCCL lacks ") + n;
        return p;
      };
      r.CommInitAll = reinterpret_cast<decltype(r.CommInitAll)>(sym("ncclCommInitAll"));
      r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(sym("ncclCommDestroy"));
      r.Reduce = reinterpret_cast<decltype(r.Reduce)>(sym("ncclReduce"));
      r.Send = reinterpret_cast<decltype(r.Send)>(sym("ncclSend"));
      r.Recv = reinterpret_cast<decltype(r.Recv)>(sym("ncclRecv"));
      r.GroupStart = reinterpret_cast<decltype(r.GroupStart)>(sym("ncclGroupStart"));
      r.GroupEnd = reinterpret_cast<decltype(r.GroupEnd)>(sym("ncclGroupEnd"));
      r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(sym("ncclGetErrorString"));
      r.GetVersion = reinterpret_cast<decltype(r.GetVersion)>(sym("ncclGetVersion"));
    });
    if (!r.error.empty()) {
      why = r.error;
      return nullptr;
    }
    return &r;
  }
};

}  // namespace glz
