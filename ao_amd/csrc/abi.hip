// ao_amd/csrc/abi.hip -- library identification (host only).
#include "common.h"

extern "C" int ptv2_abi_version(void) { return 1; }

extern "C" const char *ptv2_build_info(void) {
    return "libptv2_hip gfx950 (MI355X) hipcc " __VERSION__ " built " __DATE__;
}
