// Library plumbing: ABI version, thread-local error text, device query.
#include "avsd_common.h"
#include <string.h>

static thread_local char g_err[512] = "";

void avsd_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" int avsd_abi_version(void) { return AVSD_ABI_VERSION; }
extern "C" const char* avsd_precision(void) { return AVSD_PRECISION_NAME; }
extern "C" int avsd_sizeof_xattn_desc(void) { return (int)sizeof(avsd_xattn_desc); }

extern "C" const char* avsd_last_error(void) { return g_err; }

extern "C" int avsd_device_info(char* name, int len, int* num_cu) {
  int dev = 0;
  hipDeviceProp_t prop;
  if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) {
    avsd_set_error("device_info: no HIP device visible");
    return AVSD_ENODEV;
  }
  if (name && len > 0) {
    strncpy(name, prop.gcnArchName, (size_t)len - 1);
    name[len - 1] = 0;
  }
  if (num_cu) *num_cu = prop.multiProcessorCount;
  if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
    avsd_set_error("device_info: %s is not gfx950; libavsd_hip.so carries gfx950 code only", prop.gcnArchName);
    return AVSD_ENODEV;
  }
  return AVSD_OK;
}

// lets the ctypes binding verify its mirror of avsd_gemm_desc
extern "C" int avsd_sizeof_gemm_desc(void) { return (int)sizeof(avsd_gemm_desc); }
