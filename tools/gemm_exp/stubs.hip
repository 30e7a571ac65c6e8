// error plumbing of capi_internal.h for the stand-alone experiment library (the product's lives in capi.hip)
#include <cstdarg>
#include <cstdio>
#include <hip/hip_runtime.h>
#include "capi_internal.h"
namespace hstu {
int set_error(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vfprintf(stderr, fmt, ap);
  va_end(ap);
  fputc('\n', stderr);
  return code;
}
int check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return set_error(HSTU_ELAUNCH, "%s: %s", what, hipGetErrorString(e));
  return HSTU_OK;
}
}  // namespace hstu
