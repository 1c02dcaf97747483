/*
 * alchemy_hip.h -- C ABI of the MI355X (gfx950) ciphertext-arithmetic backend for ALCHEMY's evaluator.
 *
 * This is the boundary a new Lol `Tensor` instance binds to.  The reference
 * never names a backend: every ALCHEMY signature is polymorphic in the tensor type `t`
 * (Crypto/Alchemy/Language/SHE.hs:21-45, Crypto/Alchemy/Interpreter/PT2CT.hs:78-96) and the backend is
 * chosen only by the tensor type in an example's `PT` alias (examples/Arithmetic.hs:19,23).  The
 * entry points below are what `instance Tensor GT` would `foreign import ccall`; each cites the
 * reference call site (or the Lol Tensor method behind it) that it serves.  INTEGRATION.md shows
 * the Haskell-side stub.
 *
 * Conventions
 *   - plain C: no C++ types, no exceptions across the ABI (every entry point catches: std::bad_alloc -> ALCH_E_NOMEM,
 *     anything else -> ALCH_E_INTERNAL).  Every function returns an int status
 *     (ALCH_OK == 0, negative == error); alch_last_error() gives a thread-local message.
 *   - ring R'_q = Z_q[zeta_m], ANY cyclotomic index m whose odd prime factors are <= 13 (the reference's
 *     ciphertext indices are H0' = F11648 .. H5' = F20475 = 2^a 3^b 5^c 7 13, examples/Common.hs:38-54) with
 *     n = phi(m) <= 40960; for m = 2n a power of two that is Z_q[X]/(X^n+1).  RNS limbs q_0..q_{L-1}, limb 0 =
 *     outermost component of Lol's nested pair (Crypto/Alchemy/Interpreter/PT2CT/Noise.hs:82-89,130).
 *   - HOST buffers use Lol's layout: int64_t, tuple-interleaved ("AoS"): coefficient k of limb j of
 *     one ring element at data[k*L + j], residues in [0, q_j)   (ZqBasic q Int64, examples/Common.hs:35).
 *     Host buffers are caller-owned, modified in place, never retained.
 *   - DEVICE buffers (alch_buf) are library-owned handles holding ring elements limb-major in HBM:
 *     element e, limb j, coefficient k at word (e*L + j)*n + k; 32-bit words when every q_j < 2^31
 *     (all of ALCHEMY's moduli, PT2CT.hs:137-139,283-285), 64-bit words otherwise (q < 2^62).
 *   - bases (toolkit / Lol definitions; m = prod_l p_l^e_l, primes ascending, m_l = p_l^e_l, m'_l = m_l/p_l):
 *       Pow  p_j = prod_l zeta_{m_l}^{j_l}, j_l in [phi(m_l)], zeta_{m_l} = zeta_m^(m/m_l); linear index mixed radix,
 *            first factor outermost (for a two-power index: the coefficient vector)
 *       Dec  d^T = p^T L, L = kron_l (L_{p_l} (x) I_{m'_l}), L_p = lower-triangular all-ones; == Pow for a two-power index
 *       CRT  slot (s_1..s_k) holds sigma_u(x), sigma_u: zeta_m -> omega_m^u, omega_m = gen^((q-1)/m), gen = smallest
 *            generator of Z_q^*, u = i0 + p i1 (mod m_l) where s_l = (i0 - 1) m'_l + digitrev_p(i1);
 *            for a two-power index: slot k holds a(psi^(2*brev(k)+1)), psi = omega_m.
 *       g    = prod_{odd p | m} (1 - zeta_p)   (1 for a two-power index)
 *   - all work is queued on the ring's HIP stream; alch_sync() waits for it.  Entry points are thread-safe: a call that
 *     touches a ring holds the lock of that ring's DEVICE until it returns (calls on one GPU are serialised -- they only queue
 *     work --, calls on different GPUs run side by side), so a host may force tensors from any number of threads (a -threaded
 *     Haskell RTS).  The *_free functions do not take it (finalizer threads): free a handle only when no call is using it.
 *     A ring is bound to the HIP device that was current when it was created; every entry point makes that device current for
 *     the calling thread first, so rings may be used from any OS thread (Haskell `safe` calls).
 *   - there is NO CPU fallback: without a gfx950 device every compute entry point fails with
 *     ALCH_E_NO_DEVICE.
 */
#ifndef ALCHEMY_HIP_H
#define ALCHEMY_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ALCH_OK 0
#define ALCH_E_INVALID (-1)        /* bad argument (null, size, basis) */
#define ALCH_E_NOT_PRIME (-2)      /* alch_host_root only: q is not an odd prime */
#define ALCH_E_NO_CRT (-3)         /* no CRT basis over Z_q: q composite, q = 2, or q != 1 mod m -- Lol's crtFuncs returns Nothing */
#define ALCH_E_UNSUPPORTED (-4)    /* a CRT basis exists but this backend does not serve the ring: index with a prime factor > 13,
                                      limb-polynomial larger than the LDS, q >= 2^62 */
#define ALCH_E_NO_DEVICE (-5)      /* no gfx950 device / HIP runtime error at init */
#define ALCH_E_HIP (-6)            /* HIP runtime error (message in alch_last_error) */
#define ALCH_E_NOMEM (-7)          /* device or host allocation failed */
#define ALCH_E_INTERNAL (-8)       /* a C++ exception was caught at the ABI (never crosses it); message in alch_last_error */
#define ALCH_NOT_DIVISIBLE 1       /* divG family: Lol's Nothing */

/* bases of alch_buf_mulg / alch_buf_divg */
#define ALCH_BASIS_POW 0
#define ALCH_BASIS_DEC 1
#define ALCH_BASIS_CRT 2

typedef struct alch_ring alch_ring;
typedef struct alch_buf alch_buf;
typedef struct alch_hint alch_hint;
typedef struct alch_tunnel alch_tunnel;

/* flags of alch_ct_mul_relin */
#define ALCH_POW_IN 1u             /* inputs are in the Pow basis (crt applied first)          */
#define ALCH_POW_OUT 2u            /* outputs wanted in the Pow basis (crtInv applied last)     */

/* gadgets (Crypto/Alchemy/Interpreter/PT2CT.hs:139-140) */
#define ALCH_GAD_TRIV 0            /* TrivGad: one digit per limb, centred lift                 */
#define ALCH_GAD_BASE2 1           /* BaseBGad 2: sum_i ceil(log2 q_i) digits; unfused device path */

const char *alch_last_error(void);
/* Library/ABI version: (major<<16)|minor. */
uint32_t alch_version(void);

/* ---- ring context ------------------------------------------------------------------------------
 * Replaces the per-call twiddle/modulus arguments lol-cpp receives from Haskell; built once per
 * (index, modulus-list) type, i.e. once per `Cyc t m' zq` instance (PT2CT.hs:251-254).
 * m = cyclotomic index (2n).  Builds device-resident twiddle tables with the root rule above.
 * STATUS ORDER (part of the ABI; haskell/.../GT.hs `ringFor` dispatches on it, tests/test_host_logic.py and
 * tests/test_gpu_example_rings.py pin it):
 *   ALCH_E_INVALID      malformed arguments (null, L out of 1..8, a modulus < 2, repeated moduli)
 *   ALCH_E_NO_CRT       some modulus is composite, 2, or a prime that is not 1 mod m -- exactly the cases in which Lol's
 *                       `crtFuncs` is Nothing for ZqBasic (CRTrans Maybe needs a prime q with m | q - 1; a pair needs both
 *                       components).  Decided BEFORE the index is looked at: it is a property of (m, q), not of this backend.
 *                       The plaintext rings of the reference land here (Z_{2^e}, examples/Common.hs:32; Zq 7 over F4,
 *                       examples/Arithmetic.hs:23); the caller then creates the ring with alch_ring_create_nocrt.
 *   ALCH_E_UNSUPPORTED  a CRT basis exists but the backend does not serve (m, q): a Lol host keeps that (index, element type)
 *                       on lol-cpp as a whole (every basis-order-dependent method).
 *   ALCH_E_NO_DEVICE / ALCH_E_HIP / ALCH_E_NOMEM   run-time failures, checked last.
 * Two-power m: 32 <= m <= 2^17 (n <= 2^16) when every q < 2^31, m <= 2^16 (n <= 2^15) otherwise.  The largest size
 * of each word runs its transforms as two LDS-resident halves and the key switch unfused; the fused
 * kernels of alch_ct_mul_relin / alch_ct_mul_full cover n <= 2^15 (32-bit) / 2^14 (64-bit); at the largest size both
 * entry points run the same operations composed from element-wise kernels and batched transforms.
 * Any other m (composite, or two-power below 32): the pass engine of kernel_gen.hpp -- every transform LDS-resident
 * (phi(m) * word <= 160 KiB), the key switch composed from the element-wise tensor product (with mulG), batched
 * crtInv, digit transforms with decompose + reduce in their loader, and the hint inner product. */
int alch_ring_create(uint32_t m, int L, const uint64_t *q, alch_ring **out);
/* A ring WITHOUT CRT basis: Lol's Tensor over a modulus whose crtFuncs is Nothing -- a plaintext ring Z_p (any
 * 2 <= q < 2^31, e.g. the Z_{2^e} of examples/Common.hs:32) or, with q = 0, the integers (Tensor t m Int64: what
 * Lol lifts to in decrypt, PT2CT.hs:91-99).  Serves the Pow / Dec operations: l, lInv, mulG, divG (Pow, Dec), add,
 * sub, upload / download; everything that needs the CRT basis returns ALCH_E_NO_CRT. */
int alch_ring_create_nocrt(uint32_t m, int L, const uint64_t *q, alch_ring **out);
int alch_ring_destroy(alch_ring *ring);
/* Host-only: PT2CT's limb-count selection (the type-level rules of Crypto/Alchemy/Interpreter/PT2CT/Noise.hs:107-170 and
 * PT2CT.hs:132-140,234-249,281-296).  `moduli` is the circuit's modulus list in the order of the `Zqs` type list
 * (examples/Arithmetic.hs:29-34, examples/HomomRLWR.hs:37-43); a ring uses the shortest PREFIX of it with enough 6.1-bit
 * noise units, nested last-taken-outermost, i.e. the alch_ring of L limbs is created on moduli[L-1], ..., moduli[0].
 * Given the pNoise of an operation's OUTPUT, returns how many moduli its input, its key-switch hint and its output use,
 * and the pNoise its input must have (feed that to the operation before it: PT2CT resolves a circuit backwards).
 *   op ALCH_OP_MUL     mul_        (PT2CT.hs:160-177):  modSwitch . keySwitchQuad hint . modSwitch $ x * y
 *   op ALCH_OP_TUNNEL  linearCyc_  (PT2CT.hs:207-229):  modSwitch . tunnel hint . modSwitch
 * ALCH_E_INVALID when the list does not hold enough units (PT2CT's type error "You need more/bigger moduli!"). */
#define ALCH_OP_MUL 0
#define ALCH_OP_TUNNEL 1
int alch_select_limbs(const uint64_t *moduli, int n_moduli, int op, int gadget, int p_noise_out, int *L_in, int *L_hint,
                      int *L_out, int *p_noise_in);
/* Noise units a modulus holds: floor(log2 q / 6.1)  (mkModulus, Noise.hs:154-170). */
int alch_modulus_units(uint64_t q);
/* Host-only (no GPU needed): the root-rule constants of one modulus, for cross-checking against the
 * oracle: psi (primitive m-th root), and the smallest generator. */
int alch_host_root(uint32_t m, uint64_t q, uint64_t *psi, uint64_t *generator);
int alch_ring_n(const alch_ring *ring, uint32_t *n, int *L, int *word_bytes);
/* Use an externally created hipStream_t (e.g. torch's current stream) for all work of this ring. */
int alch_ring_set_stream(alch_ring *ring, void *hip_stream);
/* Launch-structure options of the fused hot-path kernels (results never depend on them; tests sweep them).  The
 * library reads no environment variable.  Names: "chunk" (ciphertexts per launch group, >= 8, default 1024),
 * "one_stream" (0/1: do not alternate chunks over two streams), "pipe" (1 = tensor kernels on one stream, one chunk ahead of the
 * key-switch kernels on the other; measured slower, default 0), "nstreams" (1 .. 4 independent pipelines the chunks rotate over; default 2,
 * 3 and 4 measured 2-4 % slower), "ks_map" (1 = limb-per-XCD item numbering of the key-switch kernel when L divides 8; no measurable
 * difference, default 0), "ks_rev" (1 = the key-switch kernel walks a chunk's ciphertexts last to first; no measurable difference, default 0),
 * "q30" (0 = rings whose moduli are all below 2^30 use the general kernels), "ks_grid" (persistent workgroups of the key-switch
 * kernel, default 4096), "ti_split" / "ti_grid" (form and grid of the tensor + crtInv kernel), "crt_half" (0 = whole-polynomial crt at 128 KiB), "rs_half" (1 = the closing rescale at n = 2^15 always as two launches of half-size workgroups), "tunnel_ep" (read by alch_tunnel_create on ring_s: 0 = transform
 * the embedded E'-coefficients at dimension phi(s') instead of phi(e')), "tunnel_fused" (alch_ct_tunnel with TrivGad hints: 2 or 4 = digit transforms and hint products in one kernel with that many digits side by side, 0 = through HBM), "tunnel_mac" (0 = the tunnel's hint inner product one Montgomery product at a time behind a separate evalLin pass; default 1: lazy 64-bit groups, k_tunnel_mac_e), "gen_nt" (threads per workgroup of the general-index transform kernels: 128, 256, 512; 0 = by ring size), "rs_lin" / "gen_fused" (0 = the
 * composed forms of the closing rescale / the general-index key switch), "scratch_mib", "rs_slots" (resident
 * workgroups of the closing rescale kernel; for alch_ct_mul_full the options of the hint's ring apply), "stream_dedicated" (1: the ring
 * gets a NEW stream with a hardware queue of its own -- set it before other rings borrow the stream.  Ordinary streams share the HIP
 * runtime's few hardware queues, and two streams that land on one queue run their kernels one after the other; a host that runs
 * independent sub-batches side by side gives each a dedicated stream (at most 32 alive per process: ALCH_E_UNSUPPORTED beyond, the ring
 * keeps its ordinary stream), see alchemy_amd/ringround.py). */
int alch_ring_set_option(alch_ring *ring, const char *name, long value);
int alch_sync(alch_ring *ring);
/* HIP-event timer on the ring's stream (what bench.py brackets the timed region with). */
int alch_timer_start(alch_ring *ring);
int alch_timer_stop(alch_ring *ring, float *elapsed_ms);

/* ---- Tensor methods on host buffers (one ring element, Lol layout, in place) --------------------
 * crt/crtInv: Tensor `crtFuncs` -> crt, crtInv; reached from every Cyc ring product under
 * (*) on CT (Crypto/Alchemy/Interpreter/Eval.hs:65-67) and keySwitchQuadCirc (Eval.hs:133). */
int alch_crt(alch_ring *ring, int64_t *data);
int alch_crtinv(alch_ring *ring, int64_t *data);
/* zipWithT (*), (+), (-) in one basis (CRT for mul): the pointwise half of a Cyc product. */
int alch_mul(alch_ring *ring, int64_t *a, const int64_t *b);
int alch_add(alch_ring *ring, int64_t *a, const int64_t *b);
int alch_sub(alch_ring *ring, int64_t *a, const int64_t *b);
/* Multiply limb j by scalar s[j] (scalarPow/scalarCRT product: toLSD/toMSD of SymmSHE). */
int alch_scale(alch_ring *ring, int64_t *a, const uint64_t *s);
/* mulG and divG families (Tensor mulGPow/mulGDec/mulGCRT, divGPow/divGDec/divGCRT; used by (*) on CT, which
 * applies mulG to every product coefficient, Eval.hs:65-67, and by decrypt, PT2CT.hs:91-99).  g = prod over the odd
 * primes p of m of (1 - zeta_p); identity for a two-power index.  divG returns ALCH_OK (Lol's Just) or
 * ALCH_NOT_DIVISIBLE (Lol's Nothing; the data are left untouched): as in lol-cpp, a Z_q limb fails when the odd
 * radical of m is not a unit mod q, an integer ring (alch_ring_create_nocrt with q = 0) when some coefficient of
 * (rad/g) a is not divisible by the radical.  divGCRT never fails. */
int alch_mulg_pow(alch_ring *ring, int64_t *a);
int alch_mulg_dec(alch_ring *ring, int64_t *a);
int alch_mulg_crt(alch_ring *ring, int64_t *a);
int alch_divg_pow(alch_ring *ring, int64_t *a);
int alch_divg_dec(alch_ring *ring, int64_t *a);
int alch_divg_crt(alch_ring *ring, int64_t *a);
/* Tensor l / lInv: decoding-basis coefficients -> powerful-basis coefficients and back (prefix sums / differences
 * along every odd-prime axis); what Cyc's toPow / toDec run, e.g. under modSwitch's rescaleDec (Eval.hs:130) and
 * decrypt's liftDec.  Identity for a two-power index. */
int alch_l(alch_ring *ring, int64_t *a);
int alch_linv(alch_ring *ring, int64_t *a);
/* Lol `decompose` for TrivGad followed by `reduce` of every digit (the first half of `switch` in
 * keySwitchQuadCirc, Eval.hs:133): c in the Pow basis; digits = L consecutive ring elements (each
 * n*L int64, Pow basis), digit i = centred lift of limb i reduced into every limb. */
int alch_decompose_triv(alch_ring *ring, const int64_t *c_pow, int64_t *digits);
/* The same for BaseBGad 2 (PT2CT.hs:140): per limb i, ceil(log2 q_i) balanced binary digits of the centred lift,
 * least significant first, the top digit absorbing the remainder; digits of limb 0 first.  *n_digits receives
 * their number D = sum_i ceil(log2 q_i); `digits` must hold D ring elements (pass NULL to query D only). */
int alch_decompose_base2(alch_ring *ring, const int64_t *c_pow, int64_t *digits, int *n_digits);

/* ---- device-resident ring-element arrays --------------------------------------------------------
 * What a Haskell `ForeignPtr`-wrapped tensor would hold; upload/download do the AoS <-> limb-major
 * transpose (and the 64 -> 32 bit narrowing when the ring uses 32-bit words). */
int alch_buf_alloc(alch_ring *ring, size_t n_elems, alch_buf **out);
int alch_buf_free(alch_buf *buf);
/* A non-owning alias of elements [first, first + count) of `parent` (same ring): lets the d_rel results of `coeffs` or the L digits
 * of `decompose` be handed on as single-element Tensor values without copies.  alch_buf_free of a view releases only the handle;
 * the parent must outlive its views (the Haskell side keeps the parent's ForeignPtr alive from the view's finalizer). */
int alch_buf_view(const alch_buf *parent, size_t first, size_t count, alch_buf **out);
int alch_buf_elems(const alch_buf *buf, size_t *n_elems);
/* Device address and size of the buffer (limb-major words, see above) for zero-copy hand-off to other device
 * code on the same GPU -- RCCL collectives that gather result batches, a caller's own kernels.
 * Work queued on the ring's stream must be ordered against the consumer by the caller (alch_sync or events). */
int alch_buf_device_ptr(const alch_buf *buf, void **ptr, size_t *bytes);
/* The ring a buffer belongs to, and the HIP device / stream a ring queues its work on: what other device code on the same GPU
 * needs to order itself against the library (include/alchemy_rccl.h queues its collectives on that stream). */
int alch_buf_ring(const alch_buf *buf, alch_ring **ring);
int alch_ring_device(const alch_ring *ring, int *device, void **hip_stream);
int alch_buf_upload(alch_buf *buf, size_t first, size_t count, const int64_t *host);
int alch_buf_download(const alch_buf *buf, size_t first, size_t count, int64_t *host);
/* Synthetic residues: word (e,j,k) = splitmix64(seed + ((e*L + j)*n + k)) mod q_j  (oracle:
 * orc_fill_uniform uses the same rule). */
int alch_buf_fill_uniform(alch_buf *buf, uint64_t seed);
/* Batched Tensor crt / crtInv / pointwise ops on elements [first, first+count). */
int alch_buf_crt(alch_buf *buf, size_t first, size_t count);
int alch_buf_crtinv(alch_buf *buf, size_t first, size_t count);
int alch_buf_mul(alch_buf *dst, const alch_buf *a, const alch_buf *b, size_t count);
int alch_buf_add(alch_buf *dst, const alch_buf *a, const alch_buf *b, size_t count);
int alch_buf_sub(alch_buf *dst, const alch_buf *a, const alch_buf *b, size_t count);
/* dst = src * s_j per limb (toLSD / toMSD scalars of SymmSHE, any basis); dst may equal src. */
int alch_buf_scale(alch_buf *dst, const alch_buf *src, size_t count, const uint64_t *s);
/* Batched l / lInv / mulG / divG on elements [first, first+count); basis = ALCH_BASIS_*.  alch_buf_divg returns
 * ALCH_NOT_DIVISIBLE when any element of the range is not divisible (the range is then unspecified). */
int alch_buf_l(alch_buf *buf, size_t first, size_t count);
int alch_buf_linv(alch_buf *buf, size_t first, size_t count);
int alch_buf_mulg(alch_buf *buf, size_t first, size_t count, int basis);
int alch_buf_divg(alch_buf *buf, size_t first, size_t count, int basis);
/* SymmSHE mulPublic (Crypto/Alchemy/Interpreter/Eval.hs:132; the first op of homomRLWR, examples/HomomRLWR.hs:52-59):
 * dst[e] = src[e] * pub[pub_index] for e < count, CRT basis (every component of every ciphertext times one public
 * ring element, already embedded / reduced into this ring by the host). */
int alch_buf_mul_public(alch_buf *dst, const alch_buf *src, const alch_buf *pub, size_t pub_index, size_t count);
/* SymmSHE addPublic (Eval.hs:131; PT2CT.hs:114-118 uses it for constants): cts[2b] += pub[pub_index] for b < batch
 * (c0 of every linear ciphertext; `pub` = the public element times l^-1 g^k, embedded, in the basis of cts). */
int alch_buf_add_public(alch_buf *cts, const alch_buf *pub, size_t pub_index, size_t batch);
/* SymmSHE addPublic with its encoding change as ONE pass (Eval.hs:131: addPublic works on the LSD form, i.e. toLSD's per-limb scalar
 * comes first): dst[e] = src[e] * s_j for every element e < 2*batch, then c0 components (even e) += pub[pub_index].  s = NULL: no
 * scalar.  dst may equal src.  Same results as alch_buf_scale followed by alch_buf_add_public. */
int alch_ct_add_public(alch_buf *dst, const alch_buf *src, size_t batch, const uint64_t *s, const alch_buf *pub, size_t pub_index);
/* Device-resident Lol `decompose` (TrivGad) + `reduce`: element `src_index` of src (Pow basis) -> L digit
 * elements written to dst[dst_first .. dst_first+L) (Pow basis). */
int alch_buf_decompose_triv(const alch_buf *src, size_t src_index, alch_buf *dst, size_t dst_first);
/* 64-bit order-independent checksum of elements [first, first+count): sum over words of
 * splitmix64(position ^ value<<20) -- used by the full-size parity tests. */
int alch_buf_checksum(const alch_buf *buf, size_t first, size_t count, uint64_t *sum);
/* The same sum with the words counted from element `position` of a larger array: the checksum of a batch that lives in several
 * buffers (sub-batches on their own streams, shards on their own GPUs) is the sum of its parts' checksums, each taken at the
 * position of the part's first element in the whole batch. */
int alch_buf_checksum_at(const alch_buf *buf, size_t first, size_t count, uint64_t position, uint64_t *sum);

/* ---- device-resident Tensor values -------------------------------------------------
 * E issues one Lol call per op (Crypto/Alchemy/Interpreter/Eval.hs:120-134) and Lol one Tensor call per basis change, so through
 * `instance Tensor GT` a ciphertext operation arrives as a chain of single-element calls.  A `GT` value may therefore hold an
 * alch_buf of ONE ring element instead of a host vector (haskell/.../GT.hs: constructor GTDev); these entry points are what its
 * methods call.  alch_buf_alloc / alch_buf_free of small buffers are served from a per-ring free list (no hipMalloc / hipFree, no
 * synchronisation; reuse is ordered on the ring's stream), and uploads / downloads of a few elements go through pinned staging
 * (no synchronisation on upload, one on download).
 *   alch_ring_share_stream   `ring` queues its work on the stream of `with` from now on: operations between the two rings
 *                            (embed / twace / coeffs, modSwitch, tunnel) then need no events.  A host that issues one Tensor call
 *                            at a time shares one stream between all its rings.
 *   alch_buf_copy            dst[dst_first + i] = src[src_first + i]
 *   alch_buf_tensor_op       dst[dst_first + i] = op(src[src_first + i]) for the unary Tensor methods, out of place (the ranges
 *                            may coincide: in place): crt / crtInv read src and write dst in one kernel, l / lInv / mulG / divG on
 *                            Pow and Dec likewise.  ALCH_T_DIVG_* return ALCH_NOT_DIVISIBLE for Lol's Nothing (dst unspecified). */
#define ALCH_T_CRT 0
#define ALCH_T_CRTINV 1
#define ALCH_T_L 2
#define ALCH_T_LINV 3
#define ALCH_T_MULG_POW 4
#define ALCH_T_MULG_DEC 5
#define ALCH_T_MULG_CRT 6
#define ALCH_T_DIVG_POW 7
#define ALCH_T_DIVG_DEC 8
#define ALCH_T_DIVG_CRT 9
int alch_ring_share_stream(alch_ring *ring, alch_ring *with);
int alch_buf_copy(alch_buf *dst, size_t dst_first, const alch_buf *src, size_t src_first, size_t count);
int alch_buf_tensor_op(alch_buf *dst, size_t dst_first, const alch_buf *src, size_t src_first, size_t count, int op);

/* ---- Tensor methods between two indices m | m'  ----------------------------------------------------
 * Lol's embedPow / embedDec / twacePowDec / coeffs and crtExtFuncs = (twaceCRT, embedCRT): what Cyc `embed`, `twace` and `coeffsCyc`
 * run, e.g. under SymmSHE encrypt / decrypt (Crypto/Alchemy/Interpreter/PT2CT.hs:84-99), mulPublic / addPublic (Eval.hs:131-132) and
 * tunnel (Eval.hs:134).  twaceCRT and embedCRT act on CRT SLOTS, so they must come from the same instance as crt / crtInv /
 * mulGCRT: they are computed from the slot rule at the top of this file (embedCRT: slot s of the big ring <- the slot of the small
 * ring whose unit is the unit of s reduced mod m; twaceCRT[t] = (mhat/mhat') g(t)^-1 sum over that fibre of g'(s) y[s], the
 * tweaked trace Tw(y) = (mhat/mhat') Tr(y g'/g)).  `small` has index m, `big` index m' (m | m'), same moduli (or both rings
 * without CRT over the same moduli: Pow / Dec forms only).
 *   embed   basis Pow: coefficient j of the small ring lands at its position in the big ring's powerful basis, zeros elsewhere;
 *           Dec: lInv_big . embedPow . l_small;  CRT: slot replication
 *   twace   basis Pow or Dec (Tensor twacePowDec: the same positions read back on either basis), or CRT
 *   coeffs  Tensor `coeffs`: the d_rel = phi(m')/phi(m) coefficient vectors over the small ring w.r.t. the relative powerful (or
 *           decoding: same positions) basis, relative indices in mixed radix, first prime outermost; dst holds d_rel * count elements,
 *           element e's coefficients at [e * d_rel, (e + 1) * d_rel).
 * Work is queued on the destination ring's stream, ordered after the source ring's. */
int alch_buf_embed(alch_buf *dst_big, const alch_buf *src_small, size_t count, int basis);
int alch_buf_twace(alch_buf *dst_small, const alch_buf *src_big, size_t count, int basis);
int alch_buf_coeffs(alch_buf *dst_small, const alch_buf *src_big, size_t count);
/* The same on host buffers (one ring element, Lol layout; `out` is distinct from `in`). */
int alch_embed_pow(alch_ring *small, alch_ring *big, const int64_t *in_small, int64_t *out_big);
int alch_embed_dec(alch_ring *small, alch_ring *big, const int64_t *in_small, int64_t *out_big);
int alch_embed_crt(alch_ring *small, alch_ring *big, const int64_t *in_small, int64_t *out_big);
int alch_twace_pow_dec(alch_ring *small, alch_ring *big, const int64_t *in_big, int64_t *out_small);
int alch_twace_crt(alch_ring *small, alch_ring *big, const int64_t *in_big, int64_t *out_small);
int alch_coeffs(alch_ring *small, alch_ring *big, const int64_t *in_big, int64_t *out_small_d_rel);
/* Host-only index tables of an index pair (no GPU needed; Lol computes the corresponding `baseIndicesPow` / `extIndicesCoeffs` in
 * Haskell): *len in = capacity of out (entries), out = entries written; out == NULL queries the length.
 *   ALCH_EXT_POW_POS   phi(m) entries: position in the big ring of the small ring's powerful-basis element j (Tensor powBasisPow:
 *                      the relative powerful basis element i is the unit vector at entry [i][0] of ALCH_EXT_COEFFS)
 *   ALCH_EXT_COEFFS    d_rel * phi(m) entries, [i][j]: position in the big ring of coefficient j of E-coefficient i
 *   ALCH_EXT_CRT_SLOT  phi(m') entries: the small ring's CRT slot behind every CRT slot of the big ring */
#define ALCH_EXT_POW_POS 0
#define ALCH_EXT_COEFFS 1
#define ALCH_EXT_CRT_SLOT 2
int alch_ext_table(uint32_t m_small, uint32_t m_big, int which, int32_t *out, size_t *len);
/* Tensor crtSetDec (host-only table construction, like the twiddle tables): the relative mod-p CRT set of O_m' / O_m for a prime p
 * not dividing m', as coefficient vectors over F_p on the decoding basis of index m' -- what Cyc's `crtSet` lifts to Z_{p^e} and
 * embeds, and `decToCRT` (examples/Common.hs:65-75) turns into the linear functions of the ring switches.  out: *count vectors of
 * phi(m') residues each, consecutive; *count in = capacity (vectors); out == NULL queries the number of CRT-set elements
 * (#<p>-cosets of Z_m'^* / #<p>-cosets of Z_m^*).  c_k is the idempotent that is 1 exactly at the primes above p indexed by I_k;
 * the I_k take one <p>-coset of Z_m'^* above every <p>-coset of Z_m^* (groups ordered by the smallest element of the coset below,
 * cosets inside a group by their smallest element, I_k = every group's k-th coset).  The set is canonical; its ORDER is this
 * library's rule (Lol's is not observable in the reference). */
int alch_crt_set_dec(uint32_t m_small, uint32_t m_big, uint32_t p, int64_t *out, size_t *count);

/* ---- key-switch hint ----------------------------------------------------------------------------
 * KSQuadCircHint gad (Cyc t m' zq) as produced by ksQuadCircHint
 * (Crypto/Alchemy/Interpreter/KeysHints.hs:101-113): one degree-1 polynomial (h0_i, h1_i) per gadget
 * digit.  host: 2*D consecutive ring elements in the CRT basis, order h0_0, h1_0, h0_1, h1_1, ... ;
 * D = L for TrivGad, sum_i ceil(log2 q_i) for BaseBGad 2 (digits of limb 0 first, least significant first;
 * alch_decompose_base2 reports D).
 * Kept device-resident (Montgomery form) for every later call. */
int alch_hint_load(alch_ring *ring, int gadget, const int64_t *host_crt, alch_hint **out);
/* Same, taking 2*L CRT-basis elements already on the device (e.g. synthetic). */
int alch_hint_from_buf(alch_ring *ring, int gadget, const alch_buf *src, alch_hint **out);
int alch_hint_free(alch_hint *hint);

/* ---- the hot path -------------------------------------------------------------------------------
 * out[b] = keySwitchQuadCirc hint (a[b] * b[b])  for b < batch: SymmSHE (*) (Eval.hs:65-67) followed
 * by keySwitchQuadCirc (Eval.hs:133) as PT2CT's mul_ sequences them (PT2CT.hs:172-177), on linear
 * ciphertexts.  a, b, out hold 2*batch ring elements: ciphertext b = elements (2b, 2b+1) = (c0, c1).
 * s_pre[j] = per-limb scalar folded into the tensor product: the product of both operands' toLSD
 * scalars and of keySwitchQuadCirc's toMSD scalar (p^-1 mod q_j when both operands are LSD); NULL = 1.
 * The (enc, k, l) metadata of SymmSHE's CT stays on the host (k1+k2+1, l1*l2 -- see
 * alchemy_amd/host/symmshe.hpp).  flags: ALCH_POW_IN / ALCH_POW_OUT; default is CRT basis in and out,
 * the representation Lol's Cyc keeps fresh ciphertexts and ring products in. */
int alch_ct_mul_relin(alch_ring *ring, const alch_hint *hint, const alch_buf *a, const alch_buf *b,
                      alch_buf *out, size_t batch, const uint64_t *s_pre, unsigned flags);

/* The complete PT2CT mul_ (PT2CT.hs:160-177):  out[b] = modSwitch (keySwitchQuadCirc hint (modSwitch (a[b] * b[b])))
 * -- SymmSHE (*) (Eval.hs:65-67), modSwitch to the hint's modulus (Eval.hs:130; PT2CTMulCtx'' "ModSwitchCtx_ ctin ->
 * hintzq", PT2CT.hs:142-158), keySwitchQuadCirc (Eval.hs:133), modSwitch to the output modulus.  Three rings with
 * the limb nesting of Noise.hs:82-89 (outermost = first): `hint` belongs to ring_h = (q_0 .. q_{Lh-1}); a and b
 * to ring_in = its last L_in limbs (L_in < Lh: KSPNoise gives a TrivGad hint one extra limb, PT2CT.hs:139);
 * out to ring_out = its last L_out limbs (L_out < Lh, at most 3 limbs dropped).  The rings are taken from the
 * handles; work is queued on ring_h's stream after everything queued so far on the other two, which in turn wait
 * for it.  s_pre[j] (j < L_in): toLSD scalars of both operands times the first modSwitch's toMSD scalar
 * (p^-1 mod q when both operands are LSD), NULL = 1; the library itself multiplies in the added moduli
 * (Rescale b -> (a,b): x -> (0, q_a x)) and performs both later toMSD as the identities they are.
 * flags: ALCH_POW_OUT leaves the result in the Pow basis (what Lol's rescale produces); operands are CRT basis.
 * BaseBGad 2 hints (composed from the entry points' own kernels): when the hint's ring has at least as many limbs as the operands',
 * both operands are switched up, then the BaseBGad key switch of alch_ct_mul_relin and the closing modSwitch.  When KSPNoise
 * (BaseBGad 2) leaves the hint on FEWER limbs than the product (PT2CT.hs:140 against :164; ring_h = the last limbs of ring_in) the
 * leading modSwitch goes down on the QUADRATIC ciphertext: tensor product on ring_in, c0 rescaled on the decoding basis, c1 and c2 on
 * the powerful basis, BaseBGad 2 digits of c2 on ring_h, hint products, closing modSwitch to ring_out. */
int alch_ct_mul_full(const alch_hint *hint, const alch_buf *a, const alch_buf *b, alch_buf *out, size_t batch,
                     const uint64_t *s_pre, unsigned flags);

/* ---- ring tunnelling ----------------------------------------------------------------
 * SymmSHE `tunnel hint` as E runs it (Eval.hs:134) between PT2CT's two modSwitch_ (PT2CT.hs:224-229), with the hint of
 * `tunnelHint f skout skin` (Crypto/Alchemy/Interpreter/KeysHints.hs:120-129); the linear functions of the reference are
 * `linearDec` lists (examples/Common.hs:65-75), the hops switch1..5 (examples/Common.hs:78-95).
 * ring_r = R'_q (input ciphertexts), ring_s = S'_q (hint and output), same moduli; E' = R' cap S' (index gcd), which must
 * satisfy Lol's tunnel conditions (alch_tunnel_info reports E' and d_rel = dim R'/E', or ALCH_E_INVALID).
 *   lin_crt : d_rel elements of ring_s, CRT basis: the values f'(d_i) of the E'-linear function on the relative decoding
 *             basis of R'/E' (Lol: `extendLin (lift f)` reduced mod q), i in the order of Tensor `coeffs`
 *   gadget  : ALCH_GAD_TRIV (examples/HomomRLWR.hs:46) or ALCH_GAD_BASE2 (examples/Tunnel.hs:24); D digits (alch_decompose_base2
 *             reports D for BaseBGad 2; D = L for TrivGad)
 *   ks_crt  : 2 * d_rel * D elements of ring_s, CRT basis: for relative index i and gadget digit t the linear hint
 *             (b, a) with b + a s_out = g_t f'(s_in p_i) + e   (p_i = relative powerful basis of R'/E'); order i, t, (b, a)
 * alch_ct_tunnel: out[b] = (f'(c0), 0) + sum_i switch(hint_i, embed(coeffsPow(c1)_i)) for linear ciphertexts with k = 0
 * (elements (2b, 2b+1)); s_pre = toMSD's per-limb scalar (NULL = 1).  CRT basis in and out unless ALCH_POW_IN /
 * ALCH_POW_OUT.  Runs on ring_s's stream; the input is not modified.
 * k > 0 (a ciphertext that has been multiplied): SymmSHE's `tunnel` runs absorbGFactors first -- every component times the element
 * reduce(liftPow(g^-k in R'_p)), a public multiplication (alch_buf_mul_public) that leaves k = 0 -- and then this call; the host mirror
 * (alchemy_amd/host/symmshe_gen.hpp, `tunnel`) and tests/test_gpu_tunnel.py do exactly that.
 * `in` may also belong to a ring holding only the LAST limbs of ring_r (same index): PT2CT emits
 * modSwitch_ .: tunnel_ hint .: modSwitch_ (PT2CT.hs:224-229) and the leading modSwitch up, x -> (0, q_a x), is then part of
 * this call -- the added limbs are zero, so their transforms, digits and hint products are skipped (same results as
 * alch_ct_mod_switch followed by alch_ct_tunnel; s_pre is indexed by ring_s's limbs either way). */
int alch_tunnel_info(const alch_ring *ring_r, const alch_ring *ring_s, uint32_t *e_prime, uint32_t *d_rel);
int alch_tunnel_create(alch_ring *ring_r, alch_ring *ring_s, int gadget, const alch_buf *lin_crt, const alch_buf *ks_crt,
                       alch_tunnel **out);
int alch_tunnel_free(alch_tunnel *t);
int alch_ct_tunnel(const alch_tunnel *t, const alch_buf *in, alch_buf *out, size_t batch, const uint64_t *s_pre, unsigned flags);

/* ---- SymmSHE modSwitch on batches of linear ciphertexts (Eval.hs:130; the two modSwitch_ of PT2CT.hs:177 and :224-229) ----
 * The rings come from the handles; the smaller ring's moduli are the LAST limbs of the bigger ring's (Noise.hs:82-89).  The
 * ciphertexts are in MSD form already (apply toMSD's per-limb scalar with alch_buf_scale first).
 *   out has more limbs: Rescale b -> (a, b) -- added limbs 0, the others times the added moduli; any basis, flags ignored.
 *   out has fewer limbs: Rescale (a, b) -> b per dropped limb (any number), c0 rescaled on the Dec basis and c1 on the Pow basis
 *   (Lol: rescaleDec / rescalePow); CRT basis in and out unless ALCH_POW_IN / ALCH_POW_OUT.  The input is not modified. */
int alch_ct_mod_switch(const alch_buf *in, alch_buf *out, size_t batch, unsigned flags);

/* ---- modSwitch building block (Eval.hs:130) ---------------------------------------
 * Rescale (a,b) -> b on Pow-basis elements: src lives in ring_src (L limbs), dst in ring_dst whose
 * limbs are ring_src's limbs 1..L-1:  dst_j = q_0^-1 * (src_j - reduce(lift src_0)). */
int alch_buf_rescale_drop0(const alch_buf *src, alch_buf *dst, size_t count);
/* Rescale b -> (a,b) (modSwitch up, PT2CT.hs:177 first modSwitch_): dst lives in a ring whose limbs 1..L are
 * src's limbs and whose limb 0 is the added modulus q_a:  dst_0 = 0,  dst_{j+1} = q_a * src_j.  Any basis. */
int alch_buf_rescale_add0(const alch_buf *src, alch_buf *dst, size_t count);

#ifdef __cplusplus
}
#endif
#endif /* ALCHEMY_HIP_H */
