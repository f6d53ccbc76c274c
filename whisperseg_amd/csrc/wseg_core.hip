// Error reporting, ABI version and device probe for libwseg.
#include "wseg_common.h"
#include <stdarg.h>

namespace wseg {
static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
}  // namespace wseg

using namespace wseg;

extern "C" int wseg_abi_version(void) { return WSEG_ABI_VERSION; }
extern "C" const char* wseg_last_error(void) { return g_err; }

extern "C" int wseg_device_info(char* name_out, size_t name_cap, int* cu_count_out) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) {
    set_error("no HIP device visible");
    return WSEG_ERR_NO_DEVICE;
  }
  int dev = 0;
  WSEG_HIP_CHECK(hipGetDevice(&dev));
  hipDeviceProp_t prop;
  WSEG_HIP_CHECK(hipGetDeviceProperties(&prop, dev));
  if (name_out && name_cap > 0) snprintf(name_out, name_cap, "%s (%s)", prop.name, prop.gcnArchName);
  if (cu_count_out) *cu_count_out = prop.multiProcessorCount;
  if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
    set_error("device %s is not gfx950; libwseg is built for MI355X only", prop.gcnArchName);
    return WSEG_ERR_NO_DEVICE;
  }
  return WSEG_OK;
}
