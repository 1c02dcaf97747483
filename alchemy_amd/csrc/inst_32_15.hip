// Instantiations of the LDS-resident NTT kernels for u32 residues, log2(n) in {15}.
#include "kernels_ntt.hpp"
namespace alch {
hipError_t dispatch32_15(int logn, const NttCall<u32>& c) {
    switch (logn) {
    case 15: return run_call<u32, 15>(c);
    default: return hipErrorInvalidValue;
    }
}
}  // namespace alch
