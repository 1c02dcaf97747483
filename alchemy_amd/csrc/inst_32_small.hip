// Instantiations of the LDS-resident NTT kernels for u32 residues, log2(n) in {4 5 6 7 8 9}.
#include "kernels_ntt.hpp"
namespace alch {
hipError_t dispatch32_small(int logn, const NttCall<u32>& c) {
    switch (logn) {
    case 4: return run_call<u32, 4>(c);
    case 5: return run_call<u32, 5>(c);
    case 6: return run_call<u32, 6>(c);
    case 7: return run_call<u32, 7>(c);
    case 8: return run_call<u32, 8>(c);
    case 9: return run_call<u32, 9>(c);
    default: return hipErrorInvalidValue;
    }
}
}  // namespace alch
