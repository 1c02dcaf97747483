// Instantiations of the general-index kernels (kernel_gen.hpp) for 32-bit residues (one translation unit per word size: the
// two compile in parallel).
#include "kernel_gen.hpp"
namespace alch {
hipError_t gen_dispatch(const GenCall<u32>& c) { return gen_run<u32>(c); }
hipError_t gen_tunnel_ks_dispatch(const DevRing<u32>& R, const GenDev<u32>& GE, const GenTunArgs<u32>& A, size_t nct, int ng, hipStream_t stream) { return gen_launch_tunnel_ks<u32>(R, GE, A, nct, ng, stream); }
hipError_t gen_ks_dispatch(const DevRing<u32>& R, const GenDev<u32>& G, const GenKsArgs<u32>& A, size_t nct, hipStream_t stream) { return gen_launch_ks<u32>(R, G, A, nct, stream); }
hipError_t gen_rescale_lin_dispatch(const DevRing<u32>& R, const GenDev<u32>& G, const u32* in, u32* res, u32* out, const DropTab<u32>& D, int dec_c0, size_t nelem, hipStream_t stream, bool pow_out) { return gen_launch_rescale_lin<u32>(R, G, in, res, out, D, dec_c0, nelem, stream, pow_out); }
}  // namespace alch
