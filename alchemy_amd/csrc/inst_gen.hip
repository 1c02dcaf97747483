// Instantiations of the general-index kernels (kernel_gen.hpp) for 32- and 64-bit residues.
#include "kernel_gen.hpp"
namespace alch {
hipError_t gen_dispatch(const GenCall<u32>& c) { return gen_run<u32>(c); }
hipError_t gen_dispatch(const GenCall<u64>& c) { return gen_run<u64>(c); }
}  // namespace alch
