// CPU-emulated build of the libaae_hip host+kernel sources (TEST INFRASTRUCTURE ONLY;
// see hip_emu.h).  Exports the same C ABI operating on host pointers.
#include "hip_emu.h"

#include "../../augmentedautoencoder_amd/csrc/aae_hip_impl.h"
