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
#ifndef WSEG_KNOBS
bool knob_compiled_out(const char* name) {
  if (getenv(name)) fprintf(stderr, "libwseg: %s is set, but this library was built without -DWSEG_KNOBS=1: the variable is ignored and the default is in effect\n", name);
  return false;
}
#endif
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

namespace wseg {
__global__ void lane_xor_probe_kernel(uint32_t* out) {
  const unsigned v = threadIdx.x * 7 + 3;
  out[0 * 64 + threadIdx.x] = lane_xor_u<1>(v);
  out[1 * 64 + threadIdx.x] = lane_xor_u<2>(v);
  out[2 * 64 + threadIdx.x] = lane_xor_u<4>(v);
  out[3 * 64 + threadIdx.x] = lane_xor_u<8>(v);
  out[4 * 64 + threadIdx.x] = lane_xor_u<16>(v);
  out[5 * 64 + threadIdx.x] = lane_xor_u<32>(v);
}
}  // namespace wseg

extern "C" int wseg_debug_lane_xor(uint32_t* out, void* stream) {
  using namespace wseg;
  if (!out) { set_error("wseg_debug_lane_xor: null argument"); return WSEG_ERR_INVALID; }
  hipLaunchKernelGGL(lane_xor_probe_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, out);
  WSEG_LAUNCH_CHECK();
  return WSEG_OK;
}
