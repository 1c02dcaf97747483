// Instantiations of the LDS-resident NTT kernels for u64 residues, log2(n) in {12 13 14}.
#include "kernels_ntt.hpp"
namespace alch {
hipError_t dispatch64_big(int logn, const NttCall<u64>& c) {
    switch (logn) {
    case 12: return run_call<u64, 12>(c);
    case 13: return run_call<u64, 13>(c);
    case 14: return run_call<u64, 14>(c);
    default: return hipErrorInvalidValue;
    }
}
}  // namespace alch
