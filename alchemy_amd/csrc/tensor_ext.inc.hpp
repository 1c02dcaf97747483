// Tensor methods between two cyclotomic indices m | m' (included at the end of alchemy_hip.hip: uses its handles and helpers).
//
// Lol's class Tensor: embedPow, embedDec, twacePowDec, coeffs, powBasisPow, crtExtFuncs = (twaceCRT, embedCRT) and
// crtSetDec.  They are off ALCHEMY's hot path, but twaceCRT / embedCRT act on CRT SLOTS: an `instance Tensor` is only sound when
// they follow the same slot rule as its crt / crtInv / mulGCRT -- so they belong to this library, computed from the rule documented
// in include/alchemy_hip.h (gen_ext_tables), not to lol-cpp.  Reference call sites: Cyc `embed` / `twace` under SymmSHE's encrypt /
// decrypt (Crypto/Alchemy/Interpreter/PT2CT.hs:84-99), mulPublic / addPublic (Eval.hs:131-132), tunnel (Eval.hs:134), and `crtSet`
// under decToCRT (examples/Common.hs:65-75).
//
//   Pow / Dec:  index gathers (twacePowDec reads the positions embedPow writes; coeffs reads d_rel such position sets);
//               embedDec = lInv_big . embedPow . l_small (the decoding basis of the small ring is not a subset of the big one's)
//   CRT:        embedCRT[s] = x[slot_small[s]];   twaceCRT[t] = (mhat/mhat') g(t)^-1 sum_{s in fibre(t)} g'(s) y[s]
#include "crtset_host.hpp"

// generic gather: element e, row i (of `rows`), limb j, position k:  out[((e*rows + i)*L + j)*n_out + k] = in[(e*L + j)*n_in + tab[i*n_out + k]]
// (0 where the table holds -1)
template <typename W>
__global__ void k_ext_gather_rows(const W* __restrict__ in, W* __restrict__ out, const int32_t* __restrict__ tab, u32 n_in, u32 n_out,
                                  u32 rows, u32 L, size_t elems) {
    const size_t total = elems * rows * L * (size_t)n_out;
    for (size_t w = blockIdx.x * (size_t)blockDim.x + threadIdx.x; w < total; w += (size_t)gridDim.x * blockDim.x) {
        const u32 k = (u32)(w % n_out);
        size_t t = w / n_out;
        const u32 j = (u32)(t % L); t /= L;
        const u32 i = (u32)(t % rows);
        const size_t e = t / rows;
        const int32_t s = tab[(size_t)i * n_out + k];
        out[w] = s < 0 ? (W)0 : in[(e * L + j) * (size_t)n_in + (u32)s];
    }
}

// twaceCRT: out[(e*L + j)*n_s + t] = scale_j * ginv_small[j][t] * sum_f gbig[j][fib[t*F + f]] * in[(e*L + j)*n_b + fib[t*F + f]]
// (tables in Montgomery form; null g tables = 1: two-power index)
template <typename W>
__global__ void k_ext_twace_crt(DevRing<W> R, const W* __restrict__ in, W* __restrict__ out, const int32_t* __restrict__ fib, u32 F,
                                u32 n_b, u32 n_s, GTab<W> gbig, GTab<W> ginv_small, Scal<W> scale_m, size_t elems) {
    const u32 L = (u32)R.L;
    const size_t total = elems * L * (size_t)n_s;
    for (size_t w = blockIdx.x * (size_t)blockDim.x + threadIdx.x; w < total; w += (size_t)gridDim.x * blockDim.x) {
        const u32 t = (u32)(w % n_s);
        const size_t pj = w / n_s;
        const u32 j = (u32)(pj % L);
        const W q = R.mod[j].q, qni = R.mod[j].qni;
        const W* src = in + pj * (size_t)n_b;
        const W* gb = gbig.p[j];
        W acc = 0;
        for (u32 f = 0; f < F; ++f) {
            const u32 s = (u32)fib[(size_t)t * F + f];
            W v = src[s];
            if (gb) v = csub(mont_mul_lazy(v, gb[s], q, qni), q);
            acc = csub((W)(acc + v), q);
        }
        if (ginv_small.p[j]) acc = csub(mont_mul_lazy(acc, ginv_small.p[j][t], q, qni), q);
        out[w] = csub(mont_mul_lazy(acc, scale_m.v[j], q, qni), q);
    }
}

static void ext_free(ExtTab& t) {
    if (t.pow_gather) (void)hipFree(t.pow_gather);
    if (t.coeffs) (void)hipFree(t.coeffs);
    if (t.slot_small) (void)hipFree(t.slot_small);
    if (t.fibres) (void)hipFree(t.fibres);
    t = ExtTab();
}

// Host tables of one index pair (no device needed): shared by alch_ext_table and the device cache.
struct ExtHost {
    u32 d_rel = 0, fibre = 0, n_small = 0, n_big = 0;
    std::vector<int32_t> pow_gather, coeffs, slot_small, fibres;
};

static int ext_host_tables(uint32_t m_small, uint32_t m_big, ExtHost& h) {
    GenHost gs, gb;
    if (m_small < 1 || m_big < 1 || m_big % m_small) return fail(ALCH_E_INVALID, "the first index must divide the second");
    if (!gen_factor(m_small, gs) || !gen_factor(m_big, gb)) return fail(ALCH_E_UNSUPPORTED, "index not served: " + (gs.error.empty() ? gb.error : gs.error));
    std::vector<int32_t> pow_pos;
    std::vector<u32> slot;
    if (!gen_ext_tables(gs, gb, h.d_rel, pow_pos, h.coeffs, slot)) return fail(ALCH_E_INVALID, "indices do not form an extension");
    h.n_small = gs.n; h.n_big = gb.n;
    h.fibre = gb.n / gs.n;
    h.pow_gather.assign(gb.n, -1);
    for (u32 j = 0; j < gs.n; ++j) h.pow_gather[(size_t)pow_pos[j]] = (int32_t)j;
    h.slot_small.assign(slot.begin(), slot.end());
    h.fibres.assign((size_t)gs.n * h.fibre, -1);
    std::vector<u32> fill(gs.n, 0);
    for (u32 s = 0; s < gb.n; ++s) {
        const u32 t = slot[s];
        if (t >= gs.n || fill[t] >= h.fibre) return fail(ALCH_E_INVALID, "internal: uneven CRT fibres");
        h.fibres[(size_t)t * h.fibre + fill[t]++] = (int32_t)s;
    }
    return ALCH_OK;
}

static int ext_tables(alch_ring* small, alch_ring* big, const ExtTab** out) {
    if (!small || !big) return fail(ALCH_E_INVALID, "null ring");
    if (big->m % small->m) return fail(ALCH_E_INVALID, "the small ring's index must divide the big ring's");
    if (small->L != big->L || small->word != big->word || small->zdom != big->zdom) return fail(ALCH_E_INVALID, "both rings must have the same moduli");
    for (int j = 0; j < small->L; ++j) if (small->q[j] != big->q[j]) return fail(ALCH_E_INVALID, "both rings must have the same moduli");
    if (small->device != big->device) return fail(ALCH_E_INVALID, "both rings must live on the same device");
    for (auto& e : big->ext) if (e.first == small->m) { *out = &e.second; return ALCH_OK; }
    ExtHost h;
    int rc = ext_host_tables(small->m, big->m, h);
    if (rc != ALCH_OK) return rc;
    BIND(big);
    ExtTab t;
    t.d_rel = h.d_rel; t.fibre = h.fibre;
    auto up = [&](int32_t** dst, const std::vector<int32_t>& v) {
        return hipMalloc((void**)dst, v.size() * sizeof(int32_t)) == hipSuccess &&
               hipMemcpy(*dst, v.data(), v.size() * sizeof(int32_t), hipMemcpyHostToDevice) == hipSuccess;
    };
    if (!up(&t.pow_gather, h.pow_gather) || !up(&t.coeffs, h.coeffs) || !up(&t.slot_small, h.slot_small) || !up(&t.fibres, h.fibres)) {
        ext_free(t);
        return fail(ALCH_E_NOMEM, "hipMalloc(extension tables) failed");
    }
    big->ext.emplace_back(small->m, t);
    *out = &big->ext.back().second;
    return ALCH_OK;
}

// order dst-ring work after everything queued on src's stream, and hand back
static int ext_order(alch_ring* work, alch_ring* other, bool before) {
    if (work->stream == other->stream) return ALCH_OK;
    if (!work->ev_x) HIP_TRY(hipEventCreateWithFlags(&work->ev_x, hipEventDisableTiming));
    if (before) { HIP_TRY(hipEventRecord(work->ev_x, other->stream)); HIP_TRY(hipStreamWaitEvent(work->stream, work->ev_x, 0)); }
    else { HIP_TRY(hipEventRecord(work->ev_x, work->stream)); HIP_TRY(hipStreamWaitEvent(other->stream, work->ev_x, 0)); }
    return ALCH_OK;
}

template <typename W>
static int ext_gather(alch_ring* work, const void* in, void* out, const int32_t* tab, u32 n_in, u32 n_out, u32 rows, size_t elems) {
    const size_t total = elems * rows * (size_t)work->L * n_out;
    hipLaunchKernelGGL((k_ext_gather_rows<W>), dim3(ew_grid(total)), dim3(256), 0, work->stream, (const W*)in, (W*)out, tab, n_in, n_out, rows,
                       (u32)work->L, elems);
    HIP_TRY(hipGetLastError());
    return ALCH_OK;
}

extern "C" int alch_buf_embed(alch_buf* dst, const alch_buf* src, size_t count, int basis) try {
    if (!dst || !src) return fail(ALCH_E_INVALID, "null buffer");
    alch_ring* big = dst->ring;
    alch_ring* small = src->ring;
    if (basis != ALCH_BASIS_POW && basis != ALCH_BASIS_DEC && basis != ALCH_BASIS_CRT) return fail(ALCH_E_INVALID, "unknown basis");
    if (count > dst->n_elems || count > src->n_elems) return fail(ALCH_E_INVALID, "count out of bounds");
    if (basis == ALCH_BASIS_CRT && (!big->has_crt || !small->has_crt)) return fail(ALCH_E_NO_CRT, "embedCRT needs the CRT basis of both rings");
    const ExtTab* t = nullptr;
    int rc = ext_tables(small, big, &t);
    if (rc != ALCH_OK || count == 0) return rc;
    BIND(big);
    if ((rc = ext_order(big, small, true)) != ALCH_OK) return rc;
    const alch_buf* from = src;
    if (basis == ALCH_BASIS_DEC && small->gen && small->gh.rad > 1) {         // Dec -> Pow in the small ring, on a scratch copy
        alch_buf* s = nullptr;
        if ((rc = scratch_get(small, count, &s)) != ALCH_OK) return rc;
        BIND(big);
        HIP_TRY(hipMemcpyAsync(s->dptr, src->dptr, count * elem_bytes(small), hipMemcpyDeviceToDevice, big->stream));
        if ((rc = columns(small, GEN_L, s->dptr, 0, count, 1, big->stream)) != ALCH_OK) return rc;
        from = s;
    }
    const int32_t* tab = basis == ALCH_BASIS_CRT ? t->slot_small : t->pow_gather;
    rc = big->word == 4 ? ext_gather<u32>(big, from->dptr, dst->dptr, tab, small->n, big->n, 1, count)
                        : ext_gather<u64>(big, from->dptr, dst->dptr, tab, small->n, big->n, 1, count);
    if (rc != ALCH_OK) return rc;
    if (basis == ALCH_BASIS_DEC && big->gen && big->gh.rad > 1 && (rc = columns(big, GEN_LINV, dst->dptr, 0, count, 1)) != ALCH_OK) return rc;
    return ext_order(big, small, false);
} catch (...) { return abi_catch(); }

extern "C" int alch_buf_twace(alch_buf* dst, const alch_buf* src, size_t count, int basis) try {
    if (!dst || !src) return fail(ALCH_E_INVALID, "null buffer");
    alch_ring* small = dst->ring;
    alch_ring* big = src->ring;
    if (basis != ALCH_BASIS_POW && basis != ALCH_BASIS_DEC && basis != ALCH_BASIS_CRT) return fail(ALCH_E_INVALID, "unknown basis");
    if (count > dst->n_elems || count > src->n_elems) return fail(ALCH_E_INVALID, "count out of bounds");
    if (basis == ALCH_BASIS_CRT && (!big->has_crt || !small->has_crt)) return fail(ALCH_E_NO_CRT, "twaceCRT needs the CRT basis of both rings");
    const ExtTab* t = nullptr;
    int rc = ext_tables(small, big, &t);
    if (rc != ALCH_OK || count == 0) return rc;
    BIND(small);
    if ((rc = ext_order(small, big, true)) != ALCH_OK) return rc;
    if (basis != ALCH_BASIS_CRT) {                                             // twacePowDec: the same positions on either basis
        rc = small->word == 4 ? ext_gather<u32>(small, src->dptr, dst->dptr, t->coeffs, big->n, small->n, 1, count)
                              : ext_gather<u64>(small, src->dptr, dst->dptr, t->coeffs, big->n, small->n, 1, count);
        if (rc != ALCH_OK) return rc;
        return ext_order(small, big, false);
    }
    const u32 mhs = small->m % 2 ? small->m : small->m / 2, mhb = big->m % 2 ? big->m : big->m / 2;
    uint64_t sc[MAXL] = {0};
    for (int j = 0; j < small->L; ++j) sc[j] = h_invmod((mhb / mhs) % small->q[j], small->q[j]);
    const size_t total = count * elem_words(small);
    if (small->word == 4) {
        GTab<u32> gb{}, gi{};
        for (int j = 0; j < small->L; ++j) {
            gb.p[j] = (big->gen && big->gh.rad > 1) ? big->g32.gcrt[j] : nullptr;
            gi.p[j] = (small->gen && small->gh.rad > 1) ? small->g32.gcrt_inv[j] : nullptr;
        }
        Scal<u32> sm; scal_to_mont<u32>(small, sc, 1, sm);
        hipLaunchKernelGGL((k_ext_twace_crt<u32>), dim3(ew_grid(total)), dim3(256), 0, small->stream, small->d32, (const u32*)src->dptr,
                           (u32*)dst->dptr, t->fibres, t->fibre, big->n, small->n, gb, gi, sm, count);
    } else {
        GTab<u64> gb{}, gi{};
        for (int j = 0; j < small->L; ++j) {
            gb.p[j] = (big->gen && big->gh.rad > 1) ? big->g64.gcrt[j] : nullptr;
            gi.p[j] = (small->gen && small->gh.rad > 1) ? small->g64.gcrt_inv[j] : nullptr;
        }
        Scal<u64> sm; scal_to_mont<u64>(small, sc, 1, sm);
        hipLaunchKernelGGL((k_ext_twace_crt<u64>), dim3(ew_grid(total)), dim3(256), 0, small->stream, small->d64, (const u64*)src->dptr,
                           (u64*)dst->dptr, t->fibres, t->fibre, big->n, small->n, gb, gi, sm, count);
    }
    HIP_TRY(hipGetLastError());
    return ext_order(small, big, false);
} catch (...) { return abi_catch(); }

extern "C" int alch_buf_coeffs(alch_buf* dst, const alch_buf* src, size_t count) try {
    if (!dst || !src) return fail(ALCH_E_INVALID, "null buffer");
    alch_ring* small = dst->ring;
    alch_ring* big = src->ring;
    const ExtTab* t = nullptr;
    int rc = ext_tables(small, big, &t);
    if (rc != ALCH_OK) return rc;
    if (count > src->n_elems || count > dst->n_elems / (size_t)t->d_rel) return fail(ALCH_E_INVALID, "dst must hold d_rel * count elements");
    if (count == 0) return ALCH_OK;
    BIND(small);
    if ((rc = ext_order(small, big, true)) != ALCH_OK) return rc;
    rc = small->word == 4 ? ext_gather<u32>(small, src->dptr, dst->dptr, t->coeffs, big->n, small->n, t->d_rel, count)
                          : ext_gather<u64>(small, src->dptr, dst->dptr, t->coeffs, big->n, small->n, t->d_rel, count);
    if (rc != ALCH_OK) return rc;
    return ext_order(small, big, false);
} catch (...) { return abi_catch(); }

// ---- host-buffer forms (one ring element, Lol layout), staged through fresh device buffers of both rings ----
static int ext_host(alch_ring* small, alch_ring* big, const int64_t* in, int64_t* out, int which /* 0 embed, 1 twace, 2 coeffs */, int basis) {
    if (!small || !big || !in || !out) return fail(ALCH_E_INVALID, "null argument");
    const ExtTab* t = nullptr;
    int rc = ext_tables(small, big, &t);
    if (rc != ALCH_OK) return rc;
    alch_buf *bs = nullptr, *bb = nullptr;
    const size_t ns = which == 2 ? t->d_rel : 1;
    if ((rc = alch_buf_alloc(small, ns, &bs)) != ALCH_OK) return rc;
    if ((rc = alch_buf_alloc(big, 1, &bb)) != ALCH_OK) { alch_buf_free(bs); return rc; }
    if (which == 0) {
        rc = alch_buf_upload(bs, 0, 1, in);
        if (rc == ALCH_OK) rc = alch_buf_embed(bb, bs, 1, basis);
        if (rc == ALCH_OK) rc = alch_buf_download(bb, 0, 1, out);
    } else {
        rc = alch_buf_upload(bb, 0, 1, in);
        if (rc == ALCH_OK) rc = which == 1 ? alch_buf_twace(bs, bb, 1, basis) : alch_buf_coeffs(bs, bb, 1);
        if (rc == ALCH_OK) rc = alch_buf_download(bs, 0, ns, out);
    }
    alch_buf_free(bs);
    alch_buf_free(bb);
    return rc;
}

extern "C" int alch_embed_pow(alch_ring* s, alch_ring* b, const int64_t* in, int64_t* out) try { return ext_host(s, b, in, out, 0, ALCH_BASIS_POW); } catch (...) { return abi_catch(); }
extern "C" int alch_embed_dec(alch_ring* s, alch_ring* b, const int64_t* in, int64_t* out) try { return ext_host(s, b, in, out, 0, ALCH_BASIS_DEC); } catch (...) { return abi_catch(); }
extern "C" int alch_embed_crt(alch_ring* s, alch_ring* b, const int64_t* in, int64_t* out) try { return ext_host(s, b, in, out, 0, ALCH_BASIS_CRT); } catch (...) { return abi_catch(); }
extern "C" int alch_twace_pow_dec(alch_ring* s, alch_ring* b, const int64_t* in, int64_t* out) try { return ext_host(s, b, in, out, 1, ALCH_BASIS_POW); } catch (...) { return abi_catch(); }
extern "C" int alch_twace_crt(alch_ring* s, alch_ring* b, const int64_t* in, int64_t* out) try { return ext_host(s, b, in, out, 1, ALCH_BASIS_CRT); } catch (...) { return abi_catch(); }
extern "C" int alch_coeffs(alch_ring* s, alch_ring* b, const int64_t* in, int64_t* out) try { return ext_host(s, b, in, out, 2, ALCH_BASIS_POW); } catch (...) { return abi_catch(); }

// ---- host-only tables --------------------------------------------------------------------------------------
extern "C" int alch_ext_table(uint32_t m_small, uint32_t m_big, int which, int32_t* out, size_t* len) try {
    if (!len) return fail(ALCH_E_INVALID, "null argument");
    ExtHost h;
    int rc = ext_host_tables(m_small, m_big, h);
    if (rc != ALCH_OK) return rc;
    const std::vector<int32_t>* v = nullptr;
    std::vector<int32_t> pow_pos;
    switch (which) {
    case ALCH_EXT_POW_POS: pow_pos.assign(h.coeffs.begin(), h.coeffs.begin() + h.n_small); v = &pow_pos; break;
    case ALCH_EXT_COEFFS: v = &h.coeffs; break;
    case ALCH_EXT_CRT_SLOT: v = &h.slot_small; break;
    default: return fail(ALCH_E_INVALID, "unknown table");
    }
    if (out) {
        if (*len < v->size()) return fail(ALCH_E_INVALID, "output too small: need " + std::to_string(v->size()) + " entries");
        std::copy(v->begin(), v->end(), out);
    }
    *len = v->size();
    return ALCH_OK;
} catch (...) { return abi_catch(); }

extern "C" int alch_crt_set_dec(uint32_t m_small, uint32_t m_big, uint32_t p, int64_t* out, size_t* count) try {
    if (!count) return fail(ALCH_E_INVALID, "null argument");
    std::vector<int64_t> v;
    size_t c = 0;
    std::string err;
    if (m_small < 1 || m_big < 1 || m_big % m_small) return fail(ALCH_E_INVALID, "the first index must divide the second");
    if (!out) {                                       // query: the number of CRT-set elements only (cosets, no field arithmetic)
        std::vector<std::vector<std::vector<u32>>> sets;
        if (!crt_set_cosets(m_small, m_big, p, sets, err)) return fail(ALCH_E_INVALID, err);
        *count = sets.size();
        return ALCH_OK;
    }
    if (!crt_set_dec(m_small, m_big, p, v, c, err)) return fail(ALCH_E_INVALID, err);
    if (*count < c) return fail(ALCH_E_INVALID, "output too small: the CRT set has " + std::to_string(c) + " elements");
    std::copy(v.begin(), v.end(), out);
    *count = c;
    return ALCH_OK;
} catch (...) { return abi_catch(); }
