"""TEST-ONLY shard backend: the stage calls of the sharded filter restated with the CPU oracle and
exact Python integers, so that the orchestration in composablestatespacemodels_amd/sharded.py
(split sizes, send ranges, candidate order, collectives) can run on CPU under gloo.

It mirrors GpuShard's interface; it is never imported by the product package.
"""
from __future__ import annotations

import numpy as np
import torch

from composablestatespacemodels_amd.sharded import shard_bounds
from oracle import oracle

FRAC = 96


def fix(w: float) -> int:
    """floor(w * 2^96) exactly (floats are dyadic rationals)."""
    if not (w > 0.0) or w == float("inf"):
        return 0
    m, e = np.frexp(w)            # w = m * 2^e, 0.5 <= m < 1
    mi = int(m * 2**53)
    sh = int(e) - 53 + FRAC
    return mi << sh if sh >= 0 else mi >> (-sh)


def _split64(v: int):
    lo = v & (2**64 - 1); hi = v >> 64
    f = lambda x: x - 2**64 if x >= 2**63 else x   # store as int64 bit pattern
    return f(lo), f(hi)


def _join64(lo: int, hi: int) -> int:
    return (lo & (2**64 - 1)) | ((hi & (2**64 - 1)) << 64)


class OracleShard:
    def __init__(self, model, n_global, rank, world, seed, lgcp_precision=0, resampler=0):
        self.rank, self.world, self.n_global = rank, world, n_global
        self.resampler = resampler        # 0 systematic, 1 stratified (CSSM_RESAMPLE_*): the grid the slot counts follow
        self.first, self.n = shard_bounds(n_global, world, rank)
        self.o = oracle.OraclePf(model.descriptor(lgcp_precision), self.n, seed)
        self.o.set_shard(self.first, n_global)
        self.d = self.o.d
        self.seed = seed
        self.sums5 = torch.zeros(5, dtype=torch.int64)
        self.all_sums = torch.zeros(5 * world, dtype=torch.int64)
        self.meta = torch.zeros(3 * world + 1, dtype=torch.int64)
        self.send_first, self.send_count = self.meta[:world], self.meta[world:2 * world]
        self.recv_count, self.redo_flag = self.meta[2 * world:3 * world], self.meta[3 * world:]
        self.redone = 0
        self.bits, self.fail_step = 0, None
        self.ll, self.ess, self.step_idx = 0.0, n_global, 0

    def buffer(self, name, n):
        return torch.zeros(max(n, 1), dtype=torch.float64)

    def init(self, t0):
        self.o.init(t0)
        self.ll, self.ess, self.step_idx = 0.0, self.n_global, 0
        self.next_ref = float("nan")     # LGCP (contract v8): the level predicted from the previous weighted observation's max

    def propagate(self, t, y, has_obs):
        self.o.propagate_only(t, y, bool(has_obs))
        self.x1 = self.o.proposed()
        if has_obs or self.o_is_lgcp():
            self.logw = self.o.logw()
            # known without any exchange: a function of the observation -- or, LGCP, predicted from the GLOBAL max of the weighted
            # observation before (every rank holds it after that observation's exchange); NaN for the first event of a series
            self.c = self.next_ref if self.o_is_lgcp() else self.o.ref_level(y)
            self.optimistic = not np.isnan(self.c)
            key = int(oracle.lib().oracle_c_order_key(float(self.logw.max())))
            if self.optimistic:
                self._local_sums(self.c, clamp=True)
            else:
                self.sums5[:4] = 0
            self.sums5[4] = key - 2**64 if key >= 2**63 else key
        else:
            self.o.set_particles(self.x1)
        self.step_idx += 1

    def _cnt(self, C, u):
        """#{ slots whose grid point is <= C }: systematic (one uniform u) or stratified (one per slot, keyed by GLOBAL slot)"""
        if self.resampler == 1:
            return int(oracle.lib().oracle_c_strat_count(C, self.seed, self.step_idx - 1, self.n_global))
        return int(oracle.lib().oracle_c_sys_count(C, u, self.n_global))

    def o_is_lgcp(self):
        return self.o._d.desc.obs_kind == 2

    def _local_sums(self, level, clamp):
        a = self.logw - level
        self.w1 = oracle.c_exp(np.minimum(a, 2.0**-20) if clamp else a)   # k_propagate clamps at CSSM_REF_BELOW: beyond it the step is redone anyway
        self.q = [fix(float(w)) for w in self.w1]
        S = sum(self.q); S2 = sum(fix(float(w * w)) for w in self.w1)
        a, b = _split64(S); c, d = _split64(S2)
        self.sums5[:4] = torch.tensor([a, b, c, d], dtype=torch.int64)

    def _global_max(self):
        v = [int(x) for x in self.all_sums.tolist()]
        key = max(v[5 * r + 4] & (2**64 - 1) for r in range(self.world))
        return float(oracle.lib().oracle_c_order_unkey(key))

    def sums(self):
        """Second attempt: sums relative to the level chosen with the gathered max."""
        self.optimistic = False
        self.redone += 1
        self._local_sums(float(oracle.lib().oracle_c_ref_choose(self.c, self._global_max())), clamp=False)

    def offspring(self):
        gmax = self._global_max()
        self.level = float(oracle.lib().oracle_c_ref_choose(self.c, gmax))
        if self.optimistic and not (self.level == self.c):
            self.redo_flag[0] = 1
            return
        self.redo_flag[0] = 0
        if self.o_is_lgcp():
            self.next_ref = float(oracle.lib().oracle_c_ref_predict(gmax))
        v = [int(x) for x in self.all_sums.tolist()]
        S_off = S_tot = S2_tot = 0
        for r in range(self.world):
            S = _join64(v[5 * r], v[5 * r + 1]); S2 = _join64(v[5 * r + 2], v[5 * r + 3])
            if r < self.rank:
                S_off += S
            S_tot += S; S2_tot += S2
        tot = float(S_tot) * 2.0**-96; tot2 = float(S2_tot) * 2.0**-96
        self.ll = self.ll + self.level + float(oracle.c_log(np.array([tot / self.n_global]))[0])
        self.ess = int(np.floor(1.0 / (tot2 / (tot * tot))))
        u = oracle.lib().oracle_c_u(self.seed, self.step_idx - 1)
        totd = float(S_tot)
        run = S_off
        E = np.zeros(self.n, dtype=np.int64)
        for j, q in enumerate(self.q):
            run += q
            E[j] = self._cnt(float(run) / totd, u)
        self.E = E
        e_before = self._cnt(float(S_off) / totd, u) if self.rank > 0 else 0
        per = (self.n_global + self.world - 1) // self.world
        for qd in range(self.world):
            b_lo = min(qd * per, self.n_global); b_hi = min(b_lo + per, self.n_global)
            j_lo = int(np.searchsorted(E, b_lo, side="right"))
            start = e_before if j_lo == 0 else int(E[j_lo - 1])
            if b_lo >= b_hi or j_lo >= self.n or start >= b_hi:
                self.send_first[qd] = 0; self.send_count[qd] = 0
                continue
            j_last = min(int(np.searchsorted(E, b_hi, side="left")), self.n - 1)
            self.send_first[qd] = j_lo; self.send_count[qd] = j_last - j_lo + 1

    def pack(self, first, count, buf):
        rows = []
        for q, (f, c) in enumerate(zip(first, count)):
            if q == self.rank:
                continue               # the own range never travels
            for j in range(int(f), int(f) + int(c)):
                rows.append(np.concatenate([self.x1[:, j], [float(self.E[j])]]))
        if rows:
            flat = np.concatenate(rows)
            buf[: flat.size] = torch.from_numpy(flat)

    def adopt(self, buf, n_low, n_high, self_first, self_count):
        n_rem = n_low + n_high
        rows = buf[: n_rem * (self.d + 1)].numpy().reshape(n_rem, self.d + 1)
        own = np.concatenate([self.x1[:, self_first:self_first + self_count].T,
                              self.E[self_first:self_first + self_count, None].astype(np.float64)], axis=1)
        rows = np.concatenate([rows[:n_low], own, rows[n_low:]], axis=0)   # global particle order
        n_recv = rows.shape[0]
        cand, cend = rows[:, : self.d], rows[:, self.d].astype(np.int64)
        slots = np.arange(self.first, self.first + self.n)
        anc = np.searchsorted(cend, slots, side="right")      # first candidate whose end slot exceeds s
        assert anc.max() < n_recv
        self.o.set_particles(np.ascontiguousarray(cand[anc].T))
        if hasattr(self, "_t"):
            self._record(self.step_idx - self._base, self.step_idx)

    # ---- the series API of GpuShard (records resident, observations by index), restated
    def begin(self, t, y, has):
        self._t = np.asarray(t, dtype=np.float64); self._y = np.asarray(y, dtype=np.float64)
        self._has = np.ones(len(self._t), dtype=np.uint8) if has is None else np.asarray(has)
        self.bits, self.need, self.fail_step = 0, np.zeros(len(self._t), dtype=np.uint32), None
        self.init(float(self._t.min()))
        self._base = 0
        self._path = np.zeros((len(self._t) + 1, self.d))
        self._record(0, 0)

    def want_path(self, on):
        self._want_path = bool(on)

    # -- getIntervals over the shards: the mirror of cssm_pf_shard_summary_* (include/cssm_pf.h), histogram layout included
    def summary_begin(self, interval):
        x = self.o.particles()                                             # (d, n) resampled local cloud
        rows = np.vstack([x, self.o.eta()[None, :]])
        u = np.ascontiguousarray(rows).view(np.uint64)
        self._sm_keys = np.where(u >> np.uint64(63) != 0, ~u, u | np.uint64(1 << 63))   # cssm_order_key
        self.sm_sums = torch.from_numpy(x.sum(axis=1))
        self.sm_hist = torch.zeros((self.d + 1) * 512, dtype=torch.int32)
        ng = self.n_global
        idx = int(np.floor(interval * ng))
        cl = lambda r: min(max(r, 0), ng - 1)
        self._sm_rank = [[cl(ng - idx - 1), cl(idx - 1)] for _ in range(self.d)] + [[cl(ng - idx), cl(idx)]]
        self._sm_prefix = np.zeros((self.d + 1, 2), dtype=np.uint64)

    def summary_hist(self, shift):
        h = self.sm_hist.numpy().reshape(self.d + 1, 2, 256)
        hm = np.uint64(0) if shift >= 56 else np.uint64((~0 << (shift + 8)) & (2**64 - 1))
        for row in range(self.d + 1):
            k = self._sm_keys[row]
            b = ((k >> np.uint64(shift)) & np.uint64(255)).astype(np.int64)
            for j in range(2):
                sel = (k & hm) == (self._sm_prefix[row, j] & hm)
                h[row, j] += np.bincount(b[sel], minlength=256).astype(np.int32)

    def summary_pick(self, shift):
        h = self.sm_hist.numpy().reshape(self.d + 1, 2, 256)
        for row in range(self.d + 1):
            for j in range(2):
                r, cum, b = self._sm_rank[row][j], 0, 0
                while b < 255 and cum + int(h[row, j, b]) <= r:
                    cum += int(h[row, j, b]); b += 1
                self._sm_prefix[row, j] |= np.uint64(b << shift)
                self._sm_rank[row][j] = r - cum
        h[:] = 0

    def summary_finish(self):
        k = self._sm_prefix
        val = np.where(k >> np.uint64(63) != 0, k & np.uint64((1 << 63) - 1), ~k).view(np.float64)   # cssm_order_unkey
        mean = self.sm_sums.numpy() / float(self.n_global)
        return dict(state_mean=mean, state_lower=val[:self.d, 0].copy(), state_upper=val[:self.d, 1].copy(),
                    eta_of_mean=self.o.eta_of(mean), eta_lower=float(val[self.d, 0]), eta_upper=float(val[self.d, 1]))

    def get_path(self, T):
        return self._path[: T + 1].copy()

    def _record(self, row, step_for_pick):
        """row of the path <- the current cloud's particle at the global slot sampleOne picks (if this rank owns it)"""
        if not getattr(self, "_want_path", False):
            return
        slot = int(oracle.lib().oracle_c_pick(self.seed, step_for_pick, self.n_global))
        if self.first <= slot < self.first + self.n:
            self._path[row] = self.o.particles()[:, slot - self.first]

    def begin_more(self, t, y, has):
        """T more observations of the running filter: the clock, ll, ess and the observation count go on."""
        self._t = np.asarray(t, dtype=np.float64); self._y = np.asarray(y, dtype=np.float64)
        self._has = np.ones(len(self._t), dtype=np.uint8) if has is None else np.asarray(has)
        self.bits, self.need, self.fail_step = 0, np.zeros(len(self._t), dtype=np.uint32), None
        self._base = self.step_idx

    def propagate_at(self, k, with_sums=True):
        if self.bits & 12:
            return                                   # on hold after a capacity miss (8) or void (4: level ruled out): nothing may change
        self.propagate(float(self._t[k]), float(self._y[k]), bool(self._has[k]) or self.o_is_lgcp())
        if not (bool(self._has[k]) or self.o_is_lgcp()):
            self._record(k + 1, self.step_idx)        # (a weighted observation: behind its resampling)

    def resume(self):
        """The observation whose exchange did not fit (its propagate is done); the hold is lifted."""
        assert self.bits & 8 and self.fail_step is not None
        self.bits &= ~8
        k, self.fail_step = self.fail_step, None
        return k - self._base                        # (the record's index in the resident series)

    # ---- the single-collective exchange of GpuShard (k_boundary_pack / k_offspring_expand_spec), restated
    def spec_segment(self, cap):
        R = self.d + 1
        return R * ((12 + self.d) // R) + cap * R + R * ((cap + self.d) // R)

    def boundary_pack(self, cap, send_buf):
        if self.bits & 12:
            return
        R, seg = self.d + 1, self.spec_segment(cap)
        HD = R * ((12 + self.d) // R)
        cnt_all = min(self.n, cap)
        plow, phigh = sum(self.q[:cnt_all]), sum(self.q[self.n - cnt_all:])
        buf = send_buf.numpy()
        bits = buf.view(np.uint64)
        buf[: self.world * seg] = 0.0
        S = sum(self.q)
        words = [int(x) & (2**64 - 1) for x in self.sums5.tolist()]
        for qd in range(self.world):
            o = qd * seg
            cnt = min(self.n, cap) if abs(qd - self.rank) == 1 else 0      # rows travel to the two adjacent ranks only
            first = 0 if qd < self.rank else self.n - cnt
            P, run = [], 0
            for j in range(first, first + cnt):
                run += self.q[j]
                P.append(run)
            base = 0 if qd <= self.rank else S - phigh        # a LAST block starts at S_local - P_total
            buf[o] = float(cnt)
            for i, w in enumerate(words):
                bits[o + 1 + i] = w
            bits[o + 6] = base & (2**64 - 1); bits[o + 7] = base >> 64
            bits[o + 8] = plow & (2**64 - 1); bits[o + 9] = plow >> 64
            bits[o + 10] = phigh & (2**64 - 1); bits[o + 11] = phigh >> 64
            for i in range(cnt):
                buf[o + HD + i * R: o + HD + i * R + self.d] = self.x1[:, first + i]
                bits[o + HD + i * R + self.d] = P[i] & (2**64 - 1)
                bits[o + HD + cap * R + i] = P[i] >> 64

    def adopt_spec(self, recv_buf, cap):
        if self.bits & 12:
            return
        R, seg = self.d + 1, self.spec_segment(cap)
        HD = R * ((12 + self.d) // R)
        buf = recv_buf.numpy()
        bits = buf.view(np.uint64)
        to_i64 = lambda x: x - 2**64 if x >= 2**63 else x
        for r in range(self.world):                       # every rank's 5 words are in the segment headers
            for i in range(5):
                self.all_sums[5 * r + i] = to_i64(int(bits[r * seg + 1 + i]))
        if self.optimistic and not (float(oracle.lib().oracle_c_ref_choose(self.c, self._global_max())) == self.c):
            self.bits |= 4                                  # the level first: the max rules the reference level out, the series is void
            return
        v = [int(x) for x in self.all_sums.tolist()]
        S = [_join64(v[5 * r], v[5 * r + 1]) for r in range(self.world)]
        off = [sum(S[:r]) for r in range(self.world)]
        totd = float(sum(S))
        u = oracle.lib().oracle_c_u(self.seed, self.step_idx - 1)
        cnt_of = lambda G: 0 if G == 0 else self._cnt(float(G) / totd, u)
        # the verdict every rank reaches from the headers alone: are EVERY rank's slots covered?
        plow = [int(bits[r * seg + 8]) | (int(bits[r * seg + 9]) << 64) for r in range(self.world)]
        phigh = [int(bits[r * seg + 10]) | (int(bits[r * seg + 11]) << 64) for r in range(self.world)]
        per = (self.n_global + self.world - 1) // self.world
        all_ok = True
        for r in range(self.world):
            lo = min(r * per, self.n_global); hi = min(lo + per, self.n_global)
            if lo >= hi:
                continue
            own_begin, own_end = cnt_of(off[r]), cnt_of(off[r] + S[r])
            if lo < own_begin:
                all_ok &= r > 0 and phigh[r - 1] > 0 and cnt_of(off[r - 1] + S[r - 1] - phigh[r - 1]) <= lo
            if own_end < hi:
                all_ok &= r < self.world - 1 and cnt_of(off[r + 1] + plow[r + 1]) >= hi
        if not all_ok:
            self.bits |= 8                                  # nothing is touched: the host resumes this observation with more capacity
            self.fail_step = self.step_idx - 1 if self.fail_step is None else min(self.fail_step, self.step_idx - 1)
            return
        self.offspring()                                   # own particles: level, ll, ess, end slots E
        if int(self.redo_flag[0]):
            self.bits |= 4
            return
        cands = {}
        for sdr in range(self.world):
            if sdr == self.rank:
                continue
            o = sdr * seg
            c = min(max(int(buf[o]), 0), cap)
            base = int(bits[o + 6]) | (int(bits[o + 7]) << 64)
            rows = []
            for i in range(c):
                P = int(bits[o + HD + i * R + self.d]) | (int(bits[o + HD + cap * R + i]) << 64)
                rows.append((buf[o + HD + i * R: o + HD + i * R + self.d].copy(), cnt_of(off[sdr] + base + P)))
            begin = 0 if (off[sdr] + base) == 0 else cnt_of(off[sdr] + base)
            cands[sdr] = (begin, rows)
        slot_lo, slot_hi = self.first, self.first + self.n
        low = [row for sdr in range(self.rank) for row in cands[sdr][1]]
        high = [row for sdr in range(self.rank + 1, self.world) for row in cands[sdr][1]]
        own = [(self.x1[:, j], int(self.E[j])) for j in range(self.n)]
        allc = low + own + high                            # global particle order: end slots are non-decreasing
        cend = np.array([e for _, e in allc], dtype=np.int64)
        slots = np.arange(slot_lo, slot_hi)
        anc = np.searchsorted(cend, slots, side="right")
        assert anc.max() < len(allc)
        self.o.set_particles(np.ascontiguousarray(np.stack([allc[a][0] for a in anc], axis=1)))
        self._record(self.step_idx - self._base, self.step_idx)

    def status(self, T):
        return self.ll, self.ess, self.bits, self.need[:T]

    def result(self):
        return self.ll, self.ess

    def particles(self):
        return self.o.particles()
