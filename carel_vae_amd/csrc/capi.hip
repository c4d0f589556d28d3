// Library-level entry points: ABI version, device check, thread-local error string.
#include "carel_hip_internal.h"
#include <string.h>

namespace carel {

static thread_local char g_err[512] = "";

int set_error(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}

int check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return set_error(CAREL_ERR_HIP, "%s: launch failed: %s", what, hipGetErrorString(e));
  return CAREL_OK;
}

}  // namespace carel

using namespace carel;

extern "C" int carel_abi_version(void) { return CAREL_ABI_VERSION; }

extern "C" const char* carel_last_error(void) { return g_err; }

extern "C" int carel_init(int device) {
  hipDeviceProp_t prop;
  hipError_t e = hipGetDeviceProperties(&prop, device);
  if (e != hipSuccess) return set_error(CAREL_ERR_HIP, "carel_init: hipGetDeviceProperties(%d): %s", device, hipGetErrorString(e));
  if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
    return set_error(CAREL_ERR_HIP, "carel_init: device %d is %s; this library is built for gfx950 (MI355X) only", device, prop.gcnArchName);
  return gemm_pp_init_device(device);
}
