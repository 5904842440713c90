// status strings + ABI version
#include "common.h"

extern "C" const char* aps_status_string(int status) {
  switch (status) {
    case APS_OK: return "ok";
    case APS_ERR_INVALID: return "invalid argument (pointer / size / stride)";
    case APS_ERR_UNSUPPORTED: return "configuration not supported by this build";
    case APS_ERR_LAUNCH: return "HIP launch failure";
    default: return "unknown status";
  }
}

extern "C" int aps_abi_version(void) { return 58; }
