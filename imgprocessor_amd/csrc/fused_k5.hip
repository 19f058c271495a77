// fused_k5.hip — remap -> 5x5 filter instantiations (see fused_impl.hpp)
#define IPA_FUSED_K 5
#include "fused_impl.hpp"
