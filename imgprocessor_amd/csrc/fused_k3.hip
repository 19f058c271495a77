// fused_k3.hip — remap -> 3x3 filter instantiations (see fused_impl.hpp)
#define IPA_FUSED_K 3
#include "fused_impl.hpp"
