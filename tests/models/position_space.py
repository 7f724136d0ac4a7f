"""Executable model of the device-side formulation of ArraysComp and flatten (DESIGN.md sections 4.4/4.5).

Pure Python, small inputs only.  It mirrors the data flow of the HIP kernels (position-space arrays, the
time-ordered push pool, MIS rounds among live entries, the final resolve pass) so that the *algorithm* can
be checked against the oracle's direct restatement of the reference on the CPU, independently of the GPU.
"""
import numpy as np


def factorize_position_space(n, isa, phi, plcp, maxlcp, threshold):
    """Returns (factors in emission order [(pos, src, len)], total_rounds, stats)."""
    if maxlcp + 1 <= threshold:
        return [], 0, {}
    cur = [int(x) for x in plcp]
    cur[n - 1] = 0
    # originals: positions with plcp >= threshold, ordered by (plcp, isa)  [isa >= 1 always: isa==0 is the sentinel]
    orig = {}
    order = sorted((p for p in range(n) if cur[p] >= threshold and isa[p] >= 1), key=lambda p: int(isa[p]))
    for p in order:
        orig.setdefault(cur[p], []).append(p)
    resid = [cur[p] if (cur[p] >= threshold and isa[p] >= 1) else 0 for p in range(n)]
    pool_p, pool_t = [], []
    factors = []
    rounds_total = 0
    for L in range(maxlcp, threshold - 1, -1):
        lst = list(orig.get(L, ())) + [p for p, t in zip(pool_p, pool_t) if t == L]
        m = len(lst)
        if m == 0:
            continue
        lidx = {p: k for k, p in enumerate(lst)}
        assert len(lidx) == m
        for p in lst:
            assert resid[p] == L
        vcur = [cur[p] for p in lst]
        # state: 0 undecided (live only), 1 selected, 2 not selected
        state = [0 if vcur[k] == L else 2 for k in range(m)]

        def neighbours(p):
            for q in range(max(0, p - L + 1), min(n, p + L)):
                if q != p and resid[q] == L:
                    yield q

        while any(s == 0 for s in state):
            rounds_total += 1
            new_state = list(state)
            for k in range(m):
                if state[k] != 0:
                    continue
                p = lst[k]
                blocked, hit = False, False
                for q in neighbours(p):
                    kq = lidx[q]
                    if kq < k:
                        if state[kq] == 0:
                            blocked = True
                            break
                        if state[kq] == 1:
                            hit = True
                if not blocked:
                    new_state[k] = 2 if hit else 1
            state = new_state
        # resolve pass for everything not selected
        pushes = []
        for k in range(m):
            p = lst[k]
            if state[k] == 1:
                continue
            v = vcur[k]
            if v >= threshold:
                for q in neighbours(p):
                    kq = lidx[q]
                    if kq < k and state[kq] == 1:
                        v = 0 if q < p else min(v, q - p)
            if v >= threshold:
                pushes.append((p, v))       # resid is only rewritten by pushes, after the whole resolve pass
        # apply selected
        for k in range(m):
            if state[k] != 1:
                continue
            p = lst[k]
            factors.append((p, int(phi[p]), L))
            for j in range(L):
                cur[p + j] = 0
            for j in range(min(L, p)):
                q = p - 1 - j
                cur[q] = min(cur[q], j + 1)
        for p, v in pushes:
            resid[p] = v
            pool_p.append(p)
            pool_t.append(v)
    return factors, rounds_total, {"pushes": len(pool_p)}


def flatten_rounds(factors):
    """factors: list of (pos, src, len) sorted by pos.  Round-based equivalent of FactorBuffer::flatten.
    Returns (new factor list, num_flattened, max_depth_lb, rounds)."""
    z = len(factors)
    if z == 0:
        return [], 0, 0, 0
    end = factors[-1][0] + factors[-1][2]
    owner = [-1] * end
    for i, (pos, src, ln) in enumerate(factors):
        for j in range(ln):
            owner[pos + j] = i
    orig_src = [f[1] for f in factors]
    final_src = list(orig_src)
    done = [False] * z
    cur_src = list(orig_src)
    depth = [0] * z
    rounds = 0
    while not all(done):
        rounds += 1
        snapshot = list(done)
        for i in range(z):
            if done[i]:
                continue
            pos, _, ln = factors[i]
            src = cur_src[i]
            while True:
                if src >= end or owner[src] < 0:
                    done[i] = True
                    break
                s = owner[src]
                spos, _, slen = factors[s]
                d = src - spos
                if d + ln > slen:
                    done[i] = True
                    break
                if s < i:
                    if not snapshot[s]:
                        break           # wait for s
                    ssrc = final_src[s]
                else:
                    ssrc = orig_src[s]  # s >= i: the sequential pass would still see the original value
                src = ssrc + d
                depth[i] += 1
            cur_src[i] = src
            if done[i]:
                final_src[i] = src if depth[i] else orig_src[i]
    out = [(f[0], final_src[i], f[2]) for i, f in enumerate(factors)]
    nf = sum(1 for d in depth if d)
    return out, nf, max(depth), rounds


def factorize_explicit_priority(n, isa, phi, plcp, maxlcp, threshold, rng=None):
    """Second device formulation (factorize.hip): lists are UNORDERED sets, the list order of the reference lives in
    an explicit priority array: prio[p] = ISA[p] for original candidates; entries pushed from level L get
    prio = base + rank, where rank is their index after sorting the level's pushes by (target, old prio) and base
    grows monotonically -- so inside every target list pushed entries follow the originals, later pushes follow
    earlier ones, and pushes of one level keep their relative order.  `rng` shuffles every list to prove that no
    result depends on list order."""
    if maxlcp + 1 <= threshold:
        return []
    cur = [int(x) for x in plcp]
    cur[n - 1] = 0
    prio = [int(x) for x in isa]
    resid = [cur[p] if cur[p] >= threshold else 0 for p in range(n)]
    orig = {}
    for p in range(n):                               # position order
        if cur[p] >= threshold:
            orig.setdefault(cur[p], []).append(p)
    segs = {}                                        # target level -> list of pushed positions
    prio_base = n
    factors = []
    UNDECIDED, SELECTED, NOTLIVE, REJECTED = 0, 1, 2, 3
    pst = [NOTLIVE] * n
    for L in range(maxlcp, threshold - 1, -1):
        ent = list(orig.get(L, ())) + list(segs.get(L, ()))
        if not ent:
            continue
        if rng:
            rng.shuffle(ent)
        live, stale = [], []
        for p in ent:
            assert resid[p] == L
            v = cur[p]
            if v == L:
                live.append(p); pst[p] = UNDECIDED
            else:
                pst[p] = NOTLIVE
                if v >= threshold:
                    stale.append(p)

        def window(p):
            return range(max(0, p - L + 1), min(n, p + L))

        while any(pst[p] == UNDECIDED for p in live):
            snap = list(pst)
            for p in live:
                if snap[p] != UNDECIDED:
                    continue
                hit = blocked = False
                for q in window(p):
                    if q == p or resid[q] != L:
                        continue
                    if snap[q] == SELECTED:
                        hit = True
                        break
                    if snap[q] == UNDECIDED and prio[q] < prio[p]:
                        blocked = True
                if hit:
                    pst[p] = REJECTED
                elif not blocked:
                    pst[p] = SELECTED
        pushes = []
        for p in stale + [q for q in live if pst[q] == REJECTED]:
            v = cur[p]
            for q in window(p):
                if q == p or resid[q] != L or pst[q] != SELECTED or prio[q] >= prio[p]:
                    continue
                if q < p:
                    v = 0
                    break
                v = min(v, q - p)
            if v >= threshold:
                pushes.append((v, prio[p], p))
        sel = sorted((p for p in live if pst[p] == SELECTED), key=lambda p: prio[p])
        for p in sel:
            factors.append((p, int(phi[p]), L))
            for j in range(L):
                cur[p + j] = 0
            for j in range(min(L, p)):
                cur[p - 1 - j] = min(cur[p - 1 - j], j + 1)
        pushes.sort()
        for i, (v, _, p) in enumerate(pushes):
            resid[p] = v
            prio[p] = prio_base + i
            segs.setdefault(v, []).append(p)
        prio_base += len(pushes)
    return factors


def _global_levels(n, isa, plcp, maxlcp, threshold, lcut):
    """Levels maxlcp .. lcut+1 of factorize_explicit_priority; returns the state handed to the tile pass:
    (factors, cur, prio, resid).  resid[p] = level whose list holds p's entry (0: none)."""
    cur = [int(x) for x in plcp]
    cur[n - 1] = 0
    prio = [int(x) for x in isa]
    resid = [cur[p] if cur[p] >= threshold else 0 for p in range(n)]
    lists = {}
    for p in range(n):
        if cur[p] >= threshold:
            lists.setdefault(cur[p], []).append(p)
    prio_base = n
    factors = []
    for L in range(maxlcp, max(lcut, threshold - 1), -1):
        ent = lists.pop(L, [])
        if not ent:
            continue
        live = [p for p in ent if cur[p] == L]
        stale = [p for p in ent if threshold <= cur[p] < L]
        und = set(live)
        sel = set()
        rej = set()
        while und:
            new_sel, new_rej = [], []
            for p in und:
                hit = blocked = False
                for q in range(max(0, p - L + 1), min(n, p + L)):
                    if q == p:
                        continue
                    if q in sel:
                        hit = True
                        break
                    if q in und and prio[q] < prio[p]:
                        blocked = True
                if hit:
                    new_rej.append(p)
                elif not blocked:
                    new_sel.append(p)
            for p in new_sel:
                sel.add(p); und.discard(p)
            for p in new_rej:
                rej.add(p); und.discard(p)
        pushes = []
        for p in stale + sorted(rej):
            v = cur[p]
            for q in range(max(0, p - L + 1), min(n, p + L)):
                if q in sel and prio[q] < prio[p]:
                    if q < p:
                        v = 0
                        break
                    v = min(v, q - p)
            if v >= threshold:
                pushes.append((v, prio[p], p))
        for p in ent:
            resid[p] = 0
        for p in sel:
            factors.append((p, L))
            for j in range(L):
                cur[p + j] = 0
            for j in range(min(L, p)):
                cur[p - 1 - j] = min(cur[p - 1 - j], j + 1)
        pushes.sort()
        for i, (v, _, p) in enumerate(pushes):
            resid[p] = v
            prio[p] = prio_base + i
            lists.setdefault(v, []).append(p)
        prio_base += len(pushes)
    return factors, cur, prio, resid


def factorize_tile(n, w0, w1, a, b, cur_g, prio_g, resid_g, threshold, lcut):
    """Levels lcut .. threshold for the window [w0, w1) of the text, computed from the window's slice of the global
    state alone (factorize_tiles.hip).  Everything outside the window is unknown; [fl, fr) is the range of positions
    whose state is still exactly known.  Returns (factors with a <= pos < b, valid)."""
    INF = 1 << 60
    cur = {p: cur_g[p] for p in range(w0, w1)}
    res = {p: (resid_g[p] if resid_g[p] <= lcut else 0) for p in range(w0, w1)}
    prio = {p: (0, prio_g[p]) for p in range(w0, w1)}          # (0, global) < (1, local)
    # the outermost lcut - 1 positions on either side only serve as upper bounds of what unknown factors can reach (see below)
    strip = max(lcut - 1, 0)
    fl = w0 + strip if w0 > 0 else -INF
    fr = w1 - strip if w1 < n else INF
    mid = (w0 + w1) // 2
    local_base = 0
    out = []
    UND, SEL, REJ, UNC = 0, 1, 2, 3
    for L in range(lcut, threshold - 1, -1):
        lo, hi = max(w0, fl), min(w1, fr)
        ent = [p for p in range(lo, hi) if res[p] == L]
        # An unknown factor of this level starts at an unknown position q whose working value -- which only ever decreases, and
        # of which the window holds an upper bound (only effects of certain factors were applied) -- is still >= L.  On the left
        # it covers up to q + L - 1, on the right it truncates down to q - (L - 1): the borders only move as far as such a q exists.
        dfl, dfr = fl, fr
        if fl > -INF:
            qs = [q for q in range(max(fl - L + 1, w0), min(fl, w1)) if cur[q] >= L]
            if qs:
                dfl = max(qs) + L
        if fr < INF:
            qs = [q for q in range(max(fr, w0), min(fr + L - 1, w1)) if cur[q] >= L]
            if qs:
                dfr = min(qs) - (L - 1)
        exposed = lambda p: (p < dfl) or (p >= dfr)
        st = {}
        stale = []
        for p in ent:
            if cur[p] == L:
                st[p] = UND
            elif cur[p] >= threshold:
                stale.append(p)
        while any(s == UND for s in st.values()):
            snap = dict(st)
            progressed = False
            for p, s in snap.items():
                if s != UND:
                    continue
                hit = blocked = unc = False
                for q in range(p - L + 1, p + L):
                    if q == p or q not in snap:
                        continue
                    sq = snap[q]
                    if sq == SEL:
                        hit = True
                        break
                    if prio[q] < prio[p]:
                        if sq == UND:
                            blocked = True
                        elif sq == UNC:
                            unc = True
                if hit:
                    st[p] = REJ
                elif not blocked:
                    st[p] = UNC if (unc or exposed(p)) else SEL
                if st[p] != UND:
                    progressed = True
            assert progressed
        tainted = []
        pushes = []
        for p in stale + [q for q in ent if st.get(q) == REJ]:
            v = cur[p]
            uncertain = exposed(p)
            for q in range(p - L + 1, p + L):
                if q == p or q not in st or prio[q] >= prio[p]:
                    continue
                if st[q] == UNC:
                    uncertain = True
                elif st[q] == SEL:
                    v = 0 if q < p else min(v, q - p)
            if uncertain:
                tainted.append((p, 1))
            elif v >= threshold:
                pushes.append((prio[p], v, p))
        for p in ent:
            if st.get(p) == UNC:
                tainted.append((p, L))
            res[p] = 0
        for p in ent:
            if st.get(p) == SEL:
                if a <= p < b:
                    out.append((p, L))
                for j in range(L):
                    if p + j in cur:
                        cur[p + j] = 0
                for j in range(L):
                    q = p - 1 - j
                    if q in cur:
                        cur[q] = min(cur[q], j + 1)
        pushes.sort()
        for i, (_, v, p) in enumerate(pushes):
            res[p] = v
            prio[p] = (1, local_base + i)
        local_base += len(pushes)
        nfl = dfl
        nfr = dfr
        for p, reach in tainted:
            if p < mid:
                nfl = max(nfl, p + reach)
            else:
                nfr = min(nfr, p - reach + 1) if reach == 1 else min(nfr, p - (L - 1))
        fl, fr = nfl, nfr
    return out, (fl <= a and fr >= b)


def factorize_hybrid_tiles(n, isa, phi, plcp, maxlcp, threshold, lcut, interior, halo):
    """Global levels above lcut, tile-local levels below.  Returns (factor set {(pos, len)}, tiles, invalid tiles)."""
    if maxlcp + 1 <= threshold:
        return set(), 0, 0
    factors, cur, prio, resid = _global_levels(n, isa, plcp, maxlcp, threshold, lcut)
    out = set(factors)
    tiles = invalid = 0
    for a in range(0, n, interior):
        b = min(n, a + interior)
        w0, w1 = max(0, a - halo), min(n, b + halo)
        fs, ok = factorize_tile(n, w0, w1, a, b, cur, prio, resid, threshold, min(lcut, maxlcp))
        tiles += 1
        if ok:
            out.update(fs)
        else:
            invalid += 1
            out.add(("invalid", a, b))
    return out, tiles, invalid


def max_lcp_position_space(n, isa, phi, plcp, maxlcp, threshold):
    """Device formulation of lcpcomp::MaxLCPStrategy (MaxLCPStrategy.hpp:36-100 over MaxLCPSuffixList.hpp).

    The reference's list is a stack per LCP level: an insert puts the entry in FRONT of its level (:86-124), the head of the
    highest level is taken.  Key decreases happen at once, when the truncating factor is selected, and only into lower levels,
    so when level L is reached its stack is final: the entries truncated to L, most recently truncated first, then the
    original entries by DESCENDING suffix-array index.  A factor F truncates at most one entry to L (the one at
    F.pos - L), so "most recently truncated" is the selection time of that factor.

    Position space: cur[p] = current key, prio[p] = position in its level's stack (smaller first):
      truncated by the t-th selected factor -> 2^31 - 1 - t,   never truncated -> 2^31 + (n - 1 - isa[p]).
    Level L = all p with cur[p] == L; selected = lexicographically first maximal independent set under prio (conflict =
    text distance < L); selected factors are numbered in prio order; kills, truncations (nearest start wins).
    Returns the factors in emission order."""
    if maxlcp < threshold:
        return []
    cur = [int(x) for x in plcp]
    cur[n - 1] = 0
    prio = [(1 << 31) + (n - 1 - int(isa[p])) for p in range(n)]
    orig = {}
    for p in range(n):
        if cur[p] >= threshold and isa[p] >= 1:
            orig.setdefault(cur[p], []).append(p)
    pool = {}                        # level -> positions pushed into it (stale copies are filtered by cur[p] == L)
    factors = []
    t = 0
    for L in range(maxlcp, threshold - 1, -1):
        lst = [p for p in orig.get(L, ()) if cur[p] == L] + [p for p in pool.get(L, ()) if cur[p] == L]
        assert len(set(lst)) == len(lst)
        if not lst:
            continue
        live = sorted(lst, key=lambda p: prio[p])
        sel = []
        taken = set()
        for p in live:               # sequential statement of the MIS (the kernels compute the same set by rounds)
            if all(abs(p - q) >= L for q in taken):
                taken.add(p)
                sel.append(p)
        lowered = {}
        for j, p in enumerate(sel):
            factors.append((p, int(phi[p]), L))
        for p in sel:                # kills
            for k in range(L):
                if p + k < n:
                    cur[p + k] = 0
        for j, p in enumerate(sel):  # truncations: the nearest factor start in front of s wins
            for k in range(min(L, p)):
                s, d = p - 1 - k, k + 1
                if cur[s] > d:
                    cur[s] = d
                    lowered[s] = True
        for j, p in enumerate(sel):  # one event per (factor, level): the entry whose final key is its distance to p
            for k in range(min(L, p)):
                s, d = p - 1 - k, k + 1
                if lowered.get(s) and cur[s] == d and d >= threshold:
                    prio[s] = (1 << 31) - 1 - (t + j)
                    pool.setdefault(d, []).append(s)
        t += len(sel)
    return factors


def factorize_eager(n, isa, plcp, maxlcp, threshold, rng=None):
    """Round 5: ArraysComp as a function of the factor SET (LCPCompressor sorts the factors by position before anything
    reads them, LZSSFactors.hpp:69-76, so the emission order of ArraysComp.hpp:82-110 is not observable).

    An entry is NATURAL while its working value is still its PLCP value and TRUNCATED once a selected factor at
    p = x + cur[x] cut it.  A truncated entry x with cur[x] = v has no entry of value v in (x, x + v) (everything there
    was cut to < v by the same factor) and no truncated one in (x - v, x) (its cutting factor would have cut x further),
    so truncated entries of one level never conflict with each other and only ever lose against NATURAL entries to
    their left -- which are originals of list v and therefore precede every pushed entry (ArraysComp.hpp:85-89 appends).
    Hence: the order among pushed entries is irrelevant, and neither is the list an entry waits in: a cut entry can move
    to list cur[x] at once (eager push-down).  Level L = all x with cur[x] == L; naturals in ISA order, then the truncated
    ones in any order.  Returns the factor set {(pos, len)} and the number of entry visits."""
    if maxlcp + 1 <= threshold:
        return set(), 0
    cur = [int(x) for x in plcp]
    cur[n - 1] = 0
    for p in range(n):
        if isa[p] == 0:
            cur[p] = 0                               # the candidate loop starts at SA index 1 (:54)
    trunc = [False] * n
    out = set()
    visits = 0
    for L in range(maxlcp, threshold - 1, -1):
        ent = [p for p in range(n) if cur[p] == L]
        nat = sorted((p for p in ent if not trunc[p]), key=lambda p: int(isa[p]))
        tr = [p for p in ent if trunc[p]]
        if rng:
            rng.shuffle(tr)
        for p in nat + tr:
            visits += 1
            if cur[p] != L:
                continue
            out.add((p, L))
            for j in range(L):
                cur[p + j] = 0
            for j in range(min(L, p)):
                q = p - 1 - j
                if cur[q] > j + 1:
                    cur[q] = j + 1
                    trunc[q] = True
    return out, visits


def _global_levels_eager(n, isa, plcp, maxlcp, threshold, lcut):
    """Levels maxlcp .. lcut+1 of factorize_eager; returns (factors, cur, trunc): the state handed to the eager tile pass."""
    cur = [int(x) for x in plcp]
    cur[n - 1] = 0
    for p in range(n):
        if isa[p] == 0:
            cur[p] = 0
    trunc = [False] * n
    factors = []
    for L in range(maxlcp, max(lcut, threshold - 1), -1):
        ent = [p for p in range(n) if cur[p] == L]
        order = sorted((p for p in ent if not trunc[p]), key=lambda p: int(isa[p])) + [p for p in ent if trunc[p]]
        for p in order:
            if cur[p] != L:
                continue
            factors.append((p, L))
            for j in range(L):
                cur[p + j] = 0
            for j in range(min(L, p)):
                q = p - 1 - j
                if cur[q] > j + 1:
                    cur[q] = j + 1
                    trunc[q] = True
    return factors, cur, trunc


def factorize_tile_eager(n, w0, w1, a, b, cur_g, trunc_g, isa, threshold, lcut):
    """Round 5 window pass (factorize_tiles.hip, window_eager_kernel): levels lcut .. threshold of the window [w0, w1) from ONE state
    byte per position -- value (the working LCP, which is also the list the entry waits in: cut entries move at once), a
    TRUNCATED flag and a factor-start mark.  Priorities (= ISA) are only ever compared between NATURAL entries; a natural entry
    precedes a truncated one; truncated entries never meet.  The known range [fl, fr) works as in factorize_tile.
    Returns (factors with a <= pos < b, valid)."""
    INF = 1 << 60
    val = {p: (cur_g[p] if cur_g[p] >= threshold else 0) for p in range(w0, w1)}
    tr = {p: bool(trunc_g[p]) for p in range(w0, w1)}
    sel_mark = {}
    strip = max(lcut - 1, 0)
    fl = w0 + strip if w0 > 0 else -INF
    fr = w1 - strip if w1 < n else INF
    mid = (w0 + w1) // 2
    UND, SEL, REJ, UNC = 0, 1, 2, 3
    for L in range(lcut, threshold - 1, -1):
        lo, hi = max(w0, fl), min(w1, fr)
        ub = lambda q: 0 if q in sel_mark else val[q]          # upper bound of the working value (a certain factor start is exact: 0)
        dfl, dfr = fl, fr
        if fl > -INF:
            qs = [q for q in range(max(fl - L + 1, w0), min(fl, w1)) if ub(q) >= L]
            if qs:
                dfl = max(qs) + L
        if fr < INF:
            qs = [q for q in range(max(fr, w0), min(fr + L - 1, w1)) if ub(q) >= L]
            if qs:
                dfr = min(qs) - (L - 1)
        ent = [p for p in range(lo, hi) if p not in sel_mark and val[p] == L]
        exposed = lambda p: (p < dfl) or (p >= dfr)

        def before(q, p):                      # q precedes p in list order
            if tr[q] != tr[p]:
                return not tr[q]
            if tr[q]:
                return False                   # (never meet)
            return int(isa[q]) < int(isa[p])

        st = {p: UND for p in ent}
        while any(s == UND for s in st.values()):
            snap = dict(st)
            progressed = False
            for p, s in snap.items():
                if s != UND:
                    continue
                hit = blocked = unc = False
                for q in range(p - L + 1, p + L):
                    if q == p or q not in snap:
                        continue
                    sq = snap[q]
                    if sq == SEL:
                        hit = True
                        break
                    if (sq == UND or sq == UNC) and before(q, p):
                        if sq == UND:
                            blocked = True
                        else:
                            unc = True
                if hit:
                    st[p] = REJ
                elif not blocked:
                    st[p] = UNC if (unc or exposed(p)) else SEL
                if st[p] != UND:
                    progressed = True
            assert progressed
        tl, trr = -INF, INF
        sel = [p for p in ent if st[p] == SEL]
        for p in ent:
            if st[p] == UNC:
                if p < mid:
                    tl = max(tl, p + L)
                else:
                    trr = min(trr, p - (L - 1))
        for p in sel:                          # kills first ...
            for j in range(L):
                if p + j in val:
                    val[p + j] = 0
            sel_mark[p] = L
        for p in sel:                          # ... then the cuts (a killed position stays dead)
            for k in range(1, L):
                q = p - k
                if q in val and q not in sel_mark and val[q] > k:
                    val[q] = k if k >= threshold else 0
                    tr[q] = True
        fl, fr = max(dfl, tl), min(dfr, trr)
    out = [(p, L) for p, L in sel_mark.items() if a <= p < b]
    return out, (fl <= a and fr >= b)


def factorize_hybrid_tiles_eager(n, isa, plcp, maxlcp, threshold, lcut, interior, halo):
    if maxlcp + 1 <= threshold:
        return set(), 0, 0
    factors, cur, trunc = _global_levels_eager(n, isa, plcp, maxlcp, threshold, lcut)
    out = set(factors)
    tiles = invalid = 0
    for a in range(0, n, interior):
        b = min(n, a + interior)
        w0, w1 = max(0, a - halo), min(n, b + halo)
        fs, ok = factorize_tile_eager(n, w0, w1, a, b, cur, trunc, isa, threshold, min(lcut, maxlcp))
        tiles += 1
        if ok:
            out.update(fs)
        else:
            invalid += 1
            out.add(("invalid", a, b))
    return out, tiles, invalid


def factorize_heads(n, isa, plcp, maxlcp, threshold, rng=None):
    """Round 5, global levels (factorize.hip, eager_levels_kernel): the eager formulation with explicit lists, in which a cut only ever
    inserts ONE entry.  A factor at p cuts a contiguous run [q_h, p) of alive positions (q + PLCP[q] is non-decreasing in q, and no
    selected factor can lie inside the run); the run's values p - q fall towards p, so its head q_h is decided first, at level
    p - q_h, and if it is selected it covers the whole run.  The tail can only matter if the head is covered by a factor that
    ends INSIDE the run -- at a position e whose predecessor e - 1 was alive with cur[e - 1] = cur[e] + 1 when that factor was
    selected: then e is the new head and is inserted by the factor that covered its predecessor (right-head rule).
    Lists: originals (positions with PLCP = L; skipped unless still natural, cur = L) + inserted entries (skipped unless cur = L;
    duplicates are harmless).  Returns (factor set, number of list insertions, number of cuts)."""
    if maxlcp + 1 <= threshold:
        return set(), 0, 0
    cur = [int(x) for x in plcp]
    cur[n - 1] = 0
    for p in range(n):
        if isa[p] == 0:
            cur[p] = 0
    orig = {}
    for p in range(n):
        if cur[p] >= threshold:
            orig.setdefault(cur[p], []).append(p)
    ins = {}
    out = set()
    inserts = cuts = 0
    for L in range(maxlcp, threshold - 1, -1):
        nat = sorted((p for p in orig.get(L, ()) if cur[p] == L), key=lambda p: int(isa[p]))
        natset = set(nat)
        tr = [p for p in dict.fromkeys(ins.get(L, ())) if cur[p] == L and p not in natset]      # (deduplicated)
        if rng:
            rng.shuffle(tr)
        for p in nat + tr:
            if cur[p] != L:
                continue
            out.add((p, L))
            last = cur[p + L - 1]                     # value of the last covered position before it is killed
            if L == 1:
                last = L
            for j in range(L):
                cur[p + j] = 0
            head = None
            for j in range(min(L - 1, p)):            # distances 1 .. L - 1 (distance L never lowers anything)
                q = p - 1 - j
                if cur[q] > j + 1:
                    cur[q] = j + 1
                    cuts += 1
                    head = q                          # the leftmost cut position
            if head is not None and cur[head] >= threshold:
                ins.setdefault(cur[head], []).append(head)
                inserts += 1
            r = p + L                                 # right-head rule
            if r < n and cur[r] >= threshold and last == cur[r] + 1:
                ins.setdefault(cur[r], []).append(r)
                inserts += 1
    return out, inserts, cuts


def factorize_heads_only(n, isa, plcp, maxlcp, threshold, rng=None):
    """factorize_heads with the NATURAL lists restricted to run heads as well (factorize_eager.hip as built): a position whose
    predecessor is alive with cur[q - 1] = cur[q] + 1 -- the body of a PLCP ramp, or of a cut run -- is listed nowhere; its predecessor is
    decided one level earlier and either covers it or is itself covered by a factor that ends right in front of it, and then the
    right-head rule inserts it (with its class: natural iff its value is still its PLCP value).  Returns (factor set, listed, inserted)."""
    if maxlcp + 1 <= threshold:
        return set(), 0, 0
    orig_plcp = [int(x) for x in plcp]
    cur = list(orig_plcp)
    cur[n - 1] = 0
    for p in range(n):
        if isa[p] == 0:
            cur[p] = 0
    lists = {}
    listed = 0
    for p in range(n):
        if cur[p] >= threshold and not (p > 0 and cur[p - 1] == cur[p] + 1):
            lists.setdefault(cur[p], []).append(p)
            listed += 1
    out = set()
    inserted = 0
    for L in range(maxlcp, threshold - 1, -1):
        ent = [p for p in dict.fromkeys(lists.get(L, ())) if cur[p] == L]
        nat = sorted((p for p in ent if cur[p] == orig_plcp[p]), key=lambda p: int(isa[p]))
        tr = [p for p in ent if cur[p] != orig_plcp[p]]
        if rng:
            rng.shuffle(tr)
        for p in nat + tr:
            if cur[p] != L:
                continue
            out.add((p, L))
            last = cur[p + L - 1]
            for j in range(L):
                cur[p + j] = 0
            head = None
            for j in range(min(L - 1, p)):
                q = p - 1 - j
                if cur[q] > j + 1:
                    cur[q] = j + 1
                    head = q
                else:
                    break                          # the cut positions are one contiguous run that ends in front of the factor
            if head is not None and cur[head] >= threshold:
                lists.setdefault(cur[head], []).append(head)
                inserted += 1
            r = p + L
            if r < n and cur[r] >= threshold and last == cur[r] + 1:
                lists.setdefault(cur[r], []).append(r)
                inserted += 1
    return out, listed, inserted
