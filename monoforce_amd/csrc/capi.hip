// Library-wide pieces of the C ABI: error text and version.
#include "mf_common.h"

namespace mf {
static thread_local std::string g_last_error;
void set_error(const std::string& msg) { g_last_error = msg; }
}  // namespace mf

namespace mf {
int device_cus() {
  static thread_local int cached_dev = -2, cached_cus = 256;
  int dev = -1;
  if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); return 256; }
  if (dev != cached_dev) {
    int cus = 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) { (void)hipGetLastError(); cus = 256; }
    cached_dev = dev; cached_cus = cus;
  }
  return cached_cus;
}
}  // namespace mf

namespace mf {
static thread_local const void* g_launch_stub = nullptr;
static thread_local unsigned g_launch_grid = 0, g_launch_block = 0, g_launch_count = 0;
static thread_local std::string g_launch_text;
void note_launch(const void* kernel_stub, unsigned grid, unsigned block) {
  g_launch_stub = kernel_stub; g_launch_grid = grid; g_launch_block = block; ++g_launch_count;
}
}  // namespace mf

#include <cxxabi.h>
// The last rollout kernel THIS thread launched through the library: "<demangled kernel template> grid=G block=B launches=N" (N = rollout-kernel
// launches of this thread so far; "" before the first).  Thread-local like mf_last_error: concurrent streams / threads read their own.
extern "C" const char* mf_last_launch(void) {
  if (!mf::g_launch_stub) return "";
  const char* mangled = hipKernelNameRefByPtr(mf::g_launch_stub, nullptr);
  std::string name = mangled ? mangled : "?";
  if (mangled) {
    int status = 1;
    char* dem = abi::__cxa_demangle(mangled, nullptr, nullptr, &status);
    if (status == 0 && dem) {
      name = dem;
      const size_t paren = name.find('(');      // drop the parameter list
      if (paren != std::string::npos) name.resize(paren);
      if (name.compare(0, 5, "void ") == 0) name.erase(0, 5);
    }
    free(dem);
  }
  mf::g_launch_text = name + " grid=" + std::to_string(mf::g_launch_grid) + " block=" + std::to_string(mf::g_launch_block) +
                      " launches=" + std::to_string(mf::g_launch_count);
  return mf::g_launch_text.c_str();
}

// (thread-local: two host threads driving two streams each read the message of their own failed call)
extern "C" const char* mf_last_error(void) { return mf::g_last_error.c_str(); }
extern "C" const char* mf_version(void) { return "monoforce_hip 0.1 gfx950"; }

// sizeof() of the ABI structs, so language bindings can verify their mirrors (tests/test_capi_cpu.py)
#include <string.h>
extern "C" int mf_sizeof(const char* name) {
  if (!strcmp(name, "MfRolloutDesc")) return (int)sizeof(MfRolloutDesc);
  if (!strcmp(name, "MfRolloutFwdBufs")) return (int)sizeof(MfRolloutFwdBufs);
  if (!strcmp(name, "MfRolloutBwdBufs")) return (int)sizeof(MfRolloutBwdBufs);
  if (!strcmp(name, "MfSplatDesc")) return (int)sizeof(MfSplatDesc);
  if (!strcmp(name, "MfLossDesc")) return (int)sizeof(MfLossDesc);
  if (!strcmp(name, "MfHeightmapDesc")) return (int)sizeof(MfHeightmapDesc);
  if (!strcmp(name, "MfStageDesc")) return (int)sizeof(MfStageDesc);
  if (!strcmp(name, "MfInterpDesc")) return (int)sizeof(MfInterpDesc);
  if (!strcmp(name, "MfRolloutLoss")) return (int)sizeof(MfRolloutLoss);
  return -1;
}
