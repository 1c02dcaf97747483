// Instantiations of the split (two sub-transform) crt / crtInv kernels: n = 2^16 (u32), n = 2^15 (u64).
#include "kernels_ntt.hpp"
#include "kernel_crt_split.hpp"
namespace alch {
hipError_t dispatch32_16(int logn, const NttCall<u32>& c) {
    return logn == 16 ? run_call_split<u32, 16>(c) : hipErrorInvalidValue;
}
hipError_t dispatch64_15(int logn, const NttCall<u64>& c) {
    return logn == 15 ? run_call_split<u64, 15>(c) : hipErrorInvalidValue;
}
}  // namespace alch
