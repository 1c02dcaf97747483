"""The ciphertext-op sequence of the reference's HomomRLWR example (BASELINE config 4) on resident batches.

`eval (pt2ct ringRound)` in examples/HomomRLWR.hs:45-59 runs: mulPublic, the five ring tunnels switch1..5 over
H0' .. H5' (examples/Common.hs:49-54,78-95), then rescaleTreePow2 (Language/RescaleTree.hs:64-87): x (1 + x), eight
leaves (addPublic + div2), 4 + 2 + 1 pairwise mul_ each followed by div2.  This module issues exactly those library
calls on a batch of ciphertexts that stays in HBM, with the limb counts PT2CT's type-level rules pick
(alch_select_limbs, SURVEY 3.3: tunnels 5/6/5 .. 5/5/4, products 4/5/3, 3/4/2, 2/3/1, 1/2/1) and the HomomRLWR moduli
(examples/HomomRLWR.hs:37-43).  Residues and hints are synthetic: throughput does not need valid encryptions (examples/homomrlwr_replay.cpp runs the same
sequence with real keys, hints and encryptions to the example's PASS); every bit of the result batch is checked against the C
restatement's replay of the op-by-op sequence (tests/ringround_oracle.py, bench.py's homomrlwr.batch_checksum).
Used by bench.py (extra field `homomrlwr`) and tools/bench_homomrlwr.py."""
import time

from . import capi
from .capi import Ring, Tunnel

QS = [1543651201, 689270401, 718099201, 720720001, 1556755201, 1567238401]          # Zqs order
HP = [11648, 29120, 43680, 54600, 27300, 20475]                                     # H0' .. H5'
P = 32                                                                              # plaintext modulus 2^5 (K = P5)


def moduli(L):
    return list(reversed(QS[:L]))                                                   # last-taken modulus outermost


class RingRound:
    def __init__(self, batch, ring_opts=(), pow_handoff=True, one_stream=True, first=0, dedicated_stream=False):
        self.B = batch
        self.dedicated_stream = dedicated_stream   # the chain's one stream gets a hardware queue of its own (RingRoundLanes)
        self.first = first                 # index of this batch's first ciphertext in the whole (seeded) batch: RingRoundLanes
        self.pow_handoff = pow_handoff
        # one_stream: every ring of the pipeline queues on the first ring's HIP stream (alch_ring_share_stream) -- the op sequence is one
        # dependency chain, so per-ring streams only add an event record + wait at each of its ~100 ring-to-ring hand-offs
        self.one_stream = one_stream
        self.rings, self.pool, self.cursor, self.pubs, self.stages = {}, [], 0, {}, {}
        self.ring_opts = tuple(ring_opts)
        # PT2CT's limb counts, resolved backwards from the output pNoise 0
        p, muls, tuns = 0, [], []
        for _ in range(4):
            lin, lh, lout, p = capi.select_limbs(QS, p, capi.ALCH_OP_MUL)
            muls.append((lin, lh, lout))
        for _ in range(5):
            lin, lh, lout, p = capi.select_limbs(QS, p, capi.ALCH_OP_TUNNEL)
            tuns.append((lin, lh, lout))
        muls.reverse(); tuns.reverse()                 # execution order: switch1..5, x(1+x), tree levels 1..3
        self.muls, self.tuns = muls, tuns
        # resident hints
        # (the seeded source buffers stay referenced so that a test can download them and replay the pass on the oracle)
        self.tunnels, self.tunnel_src = [], []
        for k in range(5):
            _, lh_, _ = tuns[k]
            rr, rs = self.ring(HP[k], lh_), self.ring(HP[k + 1], lh_)
            _, d_rel = Tunnel.info(rr, rs)
            self.tunnel_src.append((self.seeded(rs, d_rel, 100 + k), self.seeded(rs, 2 * d_rel * lh_, 200 + k)))
            self.tunnels.append(Tunnel(rr, rs, *self.tunnel_src[k]))
        self.quads, self.quad_src = [], []
        for _, lh_, _ in muls:
            rh = self.ring(HP[5], lh_)
            self.quad_src.append(self.seeded(rh, 2 * lh_, 300 + lh_))
            self.quads.append(rh.hint_from_buf(self.quad_src[-1]))

    def ring(self, m, L):
        if (m, L) not in self.rings:
            r = Ring(m, moduli(L))
            for k, v in self.ring_opts:
                r.set_option(k, v)
            if self.one_stream and self.rings:
                r.share_stream(next(iter(self.rings.values())))
            elif self.dedicated_stream:
                try:
                    r.set_option("stream_dedicated", 1)
                except capi.AlchemyError as e:       # the process holds its 32 dedicated streams already: an ordinary stream is still correct
                    if e.code != capi.ALCH_E_UNSUPPORTED:
                        raise
            self.rings[(m, L)] = r
        return self.rings[(m, L)]

    @staticmethod
    def seeded(r, count, seed):
        b = r.alloc(count); b.fill_uniform(seed); return b

    # buffer pool: the first pass allocates, later passes replay the same sequence of requests without any hipMalloc
    def scratch(self, r, count):
        i = self.cursor; self.cursor += 1
        if i == len(self.pool):
            self.pool.append(r.alloc(count))
        assert self.pool[i].ring is r and self.pool[i].n_elems == count
        return self.pool[i]

    def public(self, r, seed):
        if (id(r), seed) not in self.pubs:
            self.pubs[(id(r), seed)] = self.seeded(r, 1, seed)
        return self.pubs[(id(r), seed)]

    def sync(self):
        for r in self.rings.values():
            r.sync()

    def run(self, stage_times=False):
        """One pass over the batch; returns the result buffer (ciphertexts over H5' on one limb).  stage_times: synchronise
        around every stage and accumulate seconds per stage in self.stages."""
        for _ in self.steps(stage_times):
            pass
        return self.result

    def steps(self, stage_times=False):
        """run() as a generator that yields after every stage (14 of them): RingRoundLanes issues the stages of its sub-batches in
        turn, so that the chains start and end together on the device whatever the host's launch rate.  The result is self.result."""
        B, muls, tuns = self.B, self.muls, self.tuns
        ring, scratch, public = self.ring, self.scratch, self.public

        def timed(name, r, fn):
            if not stage_times:
                return fn()
            r.sync(); t0 = time.perf_counter(); v = fn(); r.sync()
            self.stages[name] = self.stages.get(name, 0.0) + time.perf_counter() - t0
            return v

        # Per-limb scalars (toLSD / toMSD of SymmSHE, div2's 2^-1) are not run as passes of their own: a ciphertext batch carries
        # a pending scalar `pend` that the next kernel taking a per-limb scalar absorbs (the tensor product's s_pre, addPublic's
        # fused scalar, the public element of mulPublic).  Arithmetic in Z_q is exact, so the results are bit for bit those of the
        # op-by-op sequence (tests/ringround_oracle.py replays that one).
        def mulv(a, b, L):
            return [x * y % q for x, y, q in zip(a, b, moduli(L))]

        # fresh ciphertexts over H0', mulPublic a (times toMSD's scalar: folded into the one public element)
        L_0 = tuns[0][0]
        r0 = ring(HP[0], L_0)
        self.cursor = 0
        if "x" not in self.pubs:
            # word (e, j, k) of the whole batch is splitmix64(1 + ((e L + j) n + k)) mod q_j (alch_buf_fill_uniform): a lane that starts
            # at ciphertext `first` = element 2 first shifts the seed by the words in front of it
            self.pubs["x"] = self.seeded(r0, 2 * B, 1 + 2 * self.first * L_0 * r0.n)
        if "pub_msd" not in self.pubs:
            ps = r0.alloc(1)
            ps.scale(public(r0, 2), 1, [pow(P, -1, q) for q in moduli(L_0)])
            self.pubs["pub_msd"] = ps
        x, x1 = self.pubs["x"], scratch(r0, 2 * B)
        timed("mulPublic", r0, lambda: x1.mul_public(x, self.pubs["pub_msd"], 0, 2 * B))
        yield
        cur = x1
        for k in range(5):
            lin_, lh_, lout_ = tuns[k]
            rr, rs, ro = ring(HP[k], lh_), ring(HP[k + 1], lh_), ring(HP[k + 1], lout_)

            def hop(cur=cur, rr=rr, rs=rs, ro=ro, k=k, lin_=lin_, lh_=lh_, lout_=lout_):
                # modSwitch_ (up) .: tunnel_ hint as one call: the ciphertexts stay on their lin_ limbs.  Between two hops the
                # ciphertexts are handed over in the Pow basis (what Lol's rescale leaves them in and its tunnel reads): the
                # closing modSwitch then needs no forward transforms and the next tunnel no inverse ones.
                pow_in = capi.ALCH_POW_IN if (k > 0 and self.pow_handoff) else 0
                pow_out = capi.ALCH_POW_OUT if (k < 4 and self.pow_handoff) else 0
                mid = scratch(rs, 2 * B)
                if lout_ < lh_:
                    self.tunnels[k].apply(cur, mid, B, flags=pow_in)
                    dn = scratch(ro, 2 * B); capi.ct_mod_switch(mid, dn, B, flags=pow_out); mid = dn
                else:
                    self.tunnels[k].apply(cur, mid, B, flags=pow_in | pow_out)
                return mid
            cur = timed(f"tunnel{k + 1}", rs, hop)
            yield
        # rescale tree on H5'
        m5 = HP[5]

        def product(level, a, pa, b, pb):
            """mul_ of (a, pending pa) and (b, pending pb): the product's own toMSD scalar P^-1 and both pending scalars ride on s_pre."""
            lin_, _, lout_ = muls[level]
            o = scratch(ring(m5, lout_), 2 * B)
            capi.ct_mul_full(self.quads[level], a, b, o, B, s_pre=mulv(mulv(pa, pb, lin_), [pow(P, -1, q) for q in moduli(lin_)], lin_))
            return o

        def plus_public(src, ps, L, seed):             # toLSD, addPublic (div2_'s modSwitchPT is metadata) -- one fused pass
            r = ring(m5, L)
            o = scratch(r, 2 * B)
            o.ct_add_public(src, B, mulv(ps, [P % q for q in moduli(L)], L), public(r, seed), 0)
            return o

        L0 = muls[0][0]
        one0 = [1] * L0

        def level0():                                   # x_lsd = P x stays pending on x itself
            return product(0, cur, [P % q for q in moduli(L0)], plus_public(cur, one0, L0, 50), one0)
        y = timed("x(1+x)", ring(m5, muls[0][1]), level0)
        yield
        L1 = muls[1][0]
        one1 = [1] * L1
        t = timed("leaves(addPublic,div2)", ring(m5, L1), lambda: [plus_public(y, one1, L1, 60 + i) for i in range(8)])
        yield
        pend = one1
        for level in (1, 2, 3):
            def tree_level(t=t, level=level, pend=pend):
                return [product(level, t[2 * i], pend, t[2 * i + 1], pend) for i in range(len(t) // 2)]
            t = timed(f"tree level {level} ({8 >> level} mul_, div2)", ring(m5, muls[level][1]), tree_level)
            pend = [pow(2, -1, q) for q in moduli(muls[level][2])]              # div2_: toMSD scalar, plaintext modulus halves (metadata)
            if level < 3:
                yield
        res = t[0]
        res.scale(res, 2 * B, pend)                     # the last div2's scalar: nothing follows that could absorb it
        self.result = res
        yield

    def measure(self, passes=1):
        """Warm-up pass (allocations), then `passes` timed passes; wall-clock seconds per pass."""
        self.run(); self.sync()
        t0 = time.perf_counter()
        for _ in range(passes):
            out = self.run()
        self.sync()
        return (time.perf_counter() - t0) / passes, out


def lane_split(batch, lanes):
    """Contiguous sub-batches of a batch: (sizes, first ciphertext of each); sizes differ by at most one, no lane is empty."""
    lanes = max(1, min(lanes, batch))
    sizes = [batch // lanes + (1 if i < batch % lanes else 0) for i in range(lanes)]
    return sizes, [sum(sizes[:i]) for i in range(lanes)]


class RingRoundLanes:
    """The same batch as `lanes` contiguous sub-batches, each a RingRound of its own on its own HIP stream (its own rings, hints and
    scratch; same seeds, so the union of the results is word for word RingRound(batch)'s).  The op sequence of one sub-batch is a
    single dependency chain of ~130 kernels, half of them memory-bound passes or the thin last wave of a transform grid; a second
    chain fills those with its own VALU-bound transforms.  Measured on MI355X at 1024 ciphertexts (tools/bench_homomrlwr_dual.py):
    1 lane 46.6 k pipelines/s, 2 lanes 51.5 k, 3 lanes 52.2 k, 4 lanes 50.7 k, 6 lanes 51.5 k (each lane on a hardware queue of its own) -- flat
    beyond two, which is the default, as for the headline's two chunk pipelines."""

    def __init__(self, batch, lanes=2, ring_opts=(), pow_handoff=True):
        sizes, firsts = lane_split(batch, lanes)
        self.B, self.sizes, self.firsts = batch, sizes, firsts
        # every lane's stream gets a hardware queue of its own ("stream_dedicated", include/alchemy_hip.h): two ordinary streams share a
        # queue every other time, and two chains on one queue run one after the other (44.7 k instead of 50.5 k pipelines/s)
        self.lanes = [RingRound(b, ring_opts, pow_handoff, True, f, True) for b, f in zip(sizes, firsts)]
        self.tuns, self.muls = self.lanes[0].tuns, self.lanes[0].muls

    def run(self):
        """One pass over the whole batch; the result buffers of the lanes in batch order (lane i holds ciphertexts firsts[i] ...).
        The lanes' stages are issued in turn (RingRound.steps): both chains are on the device from the first stage to the last."""
        gens, done = [rr.steps() for rr in self.lanes], object()
        while gens:
            gens = [g for g in gens if next(g, done) is not done]      # one stage of every lane that still has one
        return [rr.result for rr in self.lanes]

    def sync(self):
        for rr in self.lanes:
            rr.sync()

    def checksum(self, outs, count=None):
        """Checksum of the first `count` result ciphertexts (all by default), additive over ciphertexts like Buf.checksum."""
        count, total = (self.B if count is None else count), 0
        for o, b, f in zip(outs, self.sizes, self.firsts):
            take = max(0, min(b, count - f))
            if take:
                total = (total + o.checksum(0, 2 * take, 2 * f)) & ((1 << 64) - 1)
        return total

    def download(self, outs):
        import numpy as np
        return np.concatenate([o.download(0, 2 * b) for o, b in zip(outs, self.sizes)])

    def measure(self, passes=1):
        self.run(); self.sync()
        t0 = time.perf_counter()
        for _ in range(passes):
            outs = self.run()
        self.sync()
        return (time.perf_counter() - t0) / passes, outs
