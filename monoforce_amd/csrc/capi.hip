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
