// ABI version + error strings.
#include "common.h"
#include "../../include/tsgnn.h"

thread_local char tsgnn_kname_[160] = "";

extern "C" {
const char* tsgnn_last_kernel(void) { return tsgnn_kname_; }
int tsgnn_abi_version(void) { return TSGNN_ABI_VERSION; }
const char* tsgnn_strerror(int code) {
  switch (code) {
    case TSGNN_OK: return "ok";
    case TSGNN_EINVAL: return "invalid argument (null pointer, negative size, bad leading dimension)";
    case TSGNN_ELAUNCH: return "HIP launch failed";
    case TSGNN_EUNSUPPORTED: return "shape not supported by this kernel";
    default: return "unknown tsgnn error";
  }
}
}
