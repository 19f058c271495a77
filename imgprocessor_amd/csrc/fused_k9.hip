// fused_k9.hip — remap -> 9x9 filter instantiations (see fused_impl.hpp)
#define IPA_FUSED_K 9
#include "fused_impl.hpp"
