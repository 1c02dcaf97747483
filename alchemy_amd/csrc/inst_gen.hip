// Instantiations of the general-index kernels (kernel_gen.hpp) for 32- and 64-bit residues.
#include "kernel_gen.hpp"
namespace alch {
hipError_t gen_dispatch(const GenCall<u32>& c) { return gen_run<u32>(c); }
hipError_t gen_dispatch(const GenCall<u64>& c) { return gen_run<u64>(c); }
hipError_t gen_ks_dispatch(const DevRing<u32>& R, const GenDev<u32>& G, const GenKsArgs<u32>& A, size_t nct, hipStream_t stream) { return gen_launch_ks<u32>(R, G, A, nct, stream); }
hipError_t gen_ks_dispatch(const DevRing<u64>& R, const GenDev<u64>& G, const GenKsArgs<u64>& A, size_t nct, hipStream_t stream) { return gen_launch_ks<u64>(R, G, A, nct, stream); }
}  // namespace alch
