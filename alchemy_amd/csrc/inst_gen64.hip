// Instantiations of the general-index kernels (kernel_gen.hpp) for 64-bit residues (one translation unit per word size: the
// two compile in parallel).
#include "kernel_gen.hpp"
namespace alch {
hipError_t gen_dispatch(const GenCall<u64>& c) { return gen_run<u64>(c); }
hipError_t gen_ks_dispatch(const DevRing<u64>& R, const GenDev<u64>& G, const GenKsArgs<u64>& A, size_t nct, hipStream_t stream) { return gen_launch_ks<u64>(R, G, A, nct, stream); }
hipError_t gen_rescale_lin_dispatch(const DevRing<u64>& R, const GenDev<u64>& G, const u64* in, u64* res, u64* out, const DropTab<u64>& D, int dec_c0, size_t nelem, hipStream_t stream, bool pow_out) { return gen_launch_rescale_lin<u64>(R, G, in, res, out, D, dec_c0, nelem, stream, pow_out); }
}  // namespace alch
