// Library identity and error strings.
#include "spk_common.h"
#include "../../include/spkdiff.h"

extern "C" int spk_version(void) { return SPK_VERSION; }

extern "C" const char* spk_error_string(int code) {
  if (code == SPK_OK) return "ok";
  if (code == SPK_ERR_ARG) return "spkdiff: invalid argument (null pointer, non-positive size or inconsistent shapes)";
  if (code == SPK_ERR_UNSUPPORTED) return "spkdiff: unsupported configuration";
  if (code > 0) return hipGetErrorString((hipError_t)code);
  return "spkdiff: unknown error";
}
