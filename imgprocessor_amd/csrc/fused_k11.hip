// fused_k11.hip — remap -> 11x11 filter instantiations (see fused_impl.hpp)
#define IPA_FUSED_K 11
#include "fused_impl.hpp"
