// Instantiations of the LDS-resident NTT kernels for u64 residues, log2(n) in {4 5 6 7 8 9 10 11}.
#include "kernels_ntt.hpp"
namespace alch {
hipError_t dispatch64_small(int logn, const NttCall<u64>& c) {
    switch (logn) {
    case 4: return run_call<u64, 4>(c);
    case 5: return run_call<u64, 5>(c);
    case 6: return run_call<u64, 6>(c);
    case 7: return run_call<u64, 7>(c);
    case 8: return run_call<u64, 8>(c);
    case 9: return run_call<u64, 9>(c);
    case 10: return run_call<u64, 10>(c);
    case 11: return run_call<u64, 11>(c);
    default: return hipErrorInvalidValue;
    }
}
}  // namespace alch
