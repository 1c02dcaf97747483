// Instantiations of the LDS-resident NTT kernels for u32 residues, log2(n) in {14}.
#include "kernels_ntt.hpp"
namespace alch {
hipError_t dispatch32_14(int logn, const NttCall<u32>& c) {
    switch (logn) {
    case 14: return run_call<u32, 14>(c);
    default: return hipErrorInvalidValue;
    }
}
}  // namespace alch
