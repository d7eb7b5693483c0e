// Experimental build of the conv kernels: the product translation unit with the timing-only hooks switched in.
// Built by tools/build_variant.py into elvis_amd/lib/variants/<name>.so; never part of libelvis_amd.so.
#include "conv_hooks.h"
#include "../../elvis_amd/csrc/conv.hip"
