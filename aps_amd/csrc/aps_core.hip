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

extern "C" int aps_abi_version(void) { return 56; }

// A stream whose kernels may only run on the compute units of `mask` (bit i = CU i in the driver's numbering;
// `words` 32-bit words): GraphReplicas' streams when the batches in flight are given disjoint parts of the chip
// (aps_amd/replicas.py, APS_REPLICA_CU_SPLIT).  *stream_out = a hipStream_t the caller wraps (torch.cuda.ExternalStream)
// and never destroys while work may be queued on it; aps_stream_destroy gives it back.
extern "C" int aps_stream_create_masked(const uint32_t* mask, int32_t words, void** stream_out) {
  APS_CHECK_ARG(mask && words > 0 && stream_out);
  hipStream_t st = nullptr;
  const hipError_t e = hipExtStreamCreateWithCUMask(&st, (uint32_t)words, mask);
  if (e != hipSuccess) {
    fprintf(stderr, "[aps_amd] hipExtStreamCreateWithCUMask: %s\n", hipGetErrorString(e));
    return APS_ERR_LAUNCH;
  }
  *stream_out = st;
  return APS_OK;
}

extern "C" int aps_stream_destroy(void* stream) {
  APS_CHECK_ARG(stream);
  return hipStreamDestroy(static_cast<hipStream_t>(stream)) == hipSuccess ? APS_OK : APS_ERR_LAUNCH;
}
