// Instantiations of the LDS-resident NTT kernels for u32 residues, log2(n) in {10 11 12 13}.
#include "kernels_ntt.hpp"
namespace alch {
hipError_t dispatch32_mid(int logn, const NttCall<u32>& c) {
    switch (logn) {
    case 10: return run_call<u32, 10>(c);
    case 11: return run_call<u32, 11>(c);
    case 12: return run_call<u32, 12>(c);
    case 13: return run_call<u32, 13>(c);
    default: return hipErrorInvalidValue;
    }
}
}  // namespace alch
