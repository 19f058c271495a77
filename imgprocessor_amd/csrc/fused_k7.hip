// fused_k7.hip — remap -> 7x7 filter instantiations (see fused_impl.hpp)
#define IPA_FUSED_K 7
#include "fused_impl.hpp"
