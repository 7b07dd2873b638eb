// ABI version + small device utilities shared by the host side.
#include "common.h"
#include "vilgod_hip.h"

extern "C" int vg_abi_version(void) { return 1; }
