// ABI bookkeeping for libmmt_hip.so
#include "../../include/mmt_hip.h"

extern "C" int mmt_abi_version(void) { return MMT_ABI_VERSION; }
extern "C" const char* mmt_build_info(void) {
  return "libmmt_hip gfx950 (CDNA4) bf16-MFMA " __DATE__ " " __TIME__;
}
