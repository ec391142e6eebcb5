"""Exact multi-string BWT of a synthetic error-free read set -- WITHOUT suffix-sorting the reads.

TEST / BENCH INFRASTRUCTURE (workload generation), not part of the product.  torch is used for the bulk
array work (sorts, searches, scans); it runs on the GPU for the human-scale index of bench.py and on the
CPU in the tests.

Why: a 30x human-scale read set is 9e10 suffixes -- no suffix sorter fits that into this pipeline.  But the
reads are substrings of ONE random genome G, and for a random genome almost every suffix of a read is
placed by the genome suffix it starts: only the LAST few symbols of a read (fewer than the longest repeat of
G, ~30) need individual care.  With n reads of length L starting at s_r (cnt[s] reads start at s):

  rows       one per suffix of read$: (p, m) = the suffix that starts at genome position p and has m symbols
             left before its '$' (p = s + L - m, m = 0..L), cnt[p + m - L] copies; its BWT symbol is G[p-1]
             ('$' for m = L: the row of a whole read);
  order      by the string G[p..p+m)$ with '$' smallest; identical strings in genome-suffix order of p
             (any fixed rule that commutes with prepending a symbol gives a valid FM-index; the reference
             orders ties by read rank -- counts of '$'-free k-mers do not depend on it);
  long rows  m > M = 28: G[p..p+28) is unique in a random genome (a few dozen pairs of a 3e9-bp genome
             excepted), so the rows of p sit together, in genome-suffix order of p, shortest first:
             one run  G[p-1] x (reads covering p with more than M symbols left)  then  '$' x cnt[p];
  short rows m <= M: sorted explicitly by (first m symbols padded with 'A', m) -- the shorter string first --
             and merged in front of the genome positions that continue their prefix.

The work is one sort of the genome positions by their 31-mer plus sorts of the (few) short rows, both in 64
groups by leading 3-mer; the three shallowest levels ('$', 'x$', 'xy$') are emitted in front of the groups
they precede.  Output: RLE bytes in the reference's encoding (msbwt_core.rs:3-14), symbol totals.

tests/test_synth.py checks the result against the suffix-sorting builder (synth_build_msbwt) on small
genomes: same symbol totals, same count for every k-mer tried, counts = number of reads covering a window.
"""
import numpy as np

M_SHORT = 28          # rows with at most this many symbols left are sorted explicitly
KEY_BASES = 31        # genome positions are ordered by their first 31 bases (62 bits), then by position
DOLLAR = 0
CODE_OF_BASE = (1, 2, 3, 5)  # A C G T as symbol codes


def _encode_runs(torch, sym, length):
    """runs (symbol code, length > 0, neighbours differ) -> RLE bytes: sym | digit << 3, base-32 digits, least
    significant first, zero digits included (bwt_converter.rs:26-80)."""
    nb = torch.ones_like(length)
    for d in range(1, 9):
        nb += (length >= (1 << (5 * d))).to(length.dtype)
    off = torch.cumsum(nb, 0) - nb
    out = torch.zeros(int(nb.sum().item()), dtype=torch.uint8, device=sym.device)
    sym64 = sym.to(torch.int64)
    for d in range(0, 9):
        sel = nb > d
        if d and not bool(sel.any()):
            break
        out[off[sel] + d] = (sym64[sel] | (((length[sel] >> (5 * d)) & 31) << 3)).to(torch.uint8)
    return out


def _nonzero(torch, x, chunk=1 << 30):
    """indices of the non-zero elements of a 1-D tensor (torch.nonzero is limited to 2^31 - 1 elements per call)"""
    if x.numel() <= chunk:
        return torch.nonzero(x).squeeze(1)
    return torch.cat([torch.nonzero(x[lo:lo + chunk]).squeeze(1) + lo for lo in range(0, x.numel(), chunk)])


class _RunWriter:
    """Takes runs part by part (in final order), merges equal neighbours across part borders, keeps bytes on the host."""

    def __init__(self, torch):
        self.torch = torch
        self.parts = []
        self.pending = None  # (symbol, length) of the last run seen: its successor may still extend it
        self.totals = np.zeros(6, dtype=np.int64)

    def add(self, sym, length):
        torch = self.torch
        keep = length > 0
        sym, length = sym[keep], length[keep]
        if sym.numel() == 0:
            return
        if self.pending is not None:
            sym = torch.cat([torch.tensor([self.pending[0]], dtype=sym.dtype, device=sym.device), sym])
            length = torch.cat([torch.tensor([self.pending[1]], dtype=length.dtype, device=length.device), length])
        first = torch.ones(sym.numel(), dtype=torch.bool, device=sym.device)
        first[1:] = sym[1:] != sym[:-1]
        gid = torch.cumsum(first.to(torch.int64), 0) - 1
        msym = sym[first]
        mlen = torch.zeros(msym.numel(), dtype=torch.int64, device=sym.device).index_add_(0, gid, length)
        self.pending = (int(msym[-1].item()), int(mlen[-1].item()))
        msym, mlen = msym[:-1], mlen[:-1]
        if msym.numel():
            self.totals += torch.zeros(6, dtype=torch.int64, device=sym.device).index_add_(0, msym.to(torch.int64), mlen).cpu().numpy()
            self.parts.append(_encode_runs(torch, msym, mlen).cpu().numpy())

    def finish(self):
        torch = self.torch
        if self.pending is not None:
            s, n = self.pending
            self.totals[s] += n
            self.parts.append(_encode_runs(torch, torch.tensor([s], dtype=torch.uint8), torch.tensor([n], dtype=torch.int64)).numpy())
            self.pending = None
        return (np.concatenate(self.parts) if self.parts else np.zeros(0, dtype=np.uint8)), self.totals


def read_set(genome_len, read_len, coverage, seed, device="cpu"):
    """A random genome (bases 0..3; read_len + 64 spare bases at the end so that every read and every sort key is
    defined) and cnt[s] = reads starting at s (Poisson, mean coverage / read_len) for s < genome_len."""
    import torch
    dev = torch.device(device)
    gen = torch.Generator(device=dev)
    gen.manual_seed(seed)
    n = genome_len + read_len + 64
    genome = torch.empty(n, dtype=torch.uint8, device=dev)
    cnt = torch.zeros(genome_len, dtype=torch.uint8, device=dev)
    step = 1 << 28
    for lo in range(0, n, step):
        genome[lo:lo + step] = torch.randint(0, 4, (min(step, n - lo),), generator=gen, device=dev, dtype=torch.uint8)
    lam = coverage / read_len
    for lo in range(0, genome_len, step):
        m = min(step, genome_len - lo)
        cnt[lo:lo + m] = torch.poisson(torch.full((m,), lam, device=dev, dtype=torch.float32), generator=gen).clamp_(max=255).to(torch.uint8)
    return genome, cnt


def reads_of(genome, cnt, read_len):
    """The read set as an (n_reads, read_len) array of symbol codes (small scales: tests)."""
    import torch
    starts = torch.repeat_interleave(torch.arange(cnt.numel(), device=cnt.device), cnt.to(torch.int64))
    idx = starts[:, None] + torch.arange(read_len, device=cnt.device)[None, :]
    codes = torch.tensor(CODE_OF_BASE, dtype=torch.uint8, device=cnt.device)
    return codes[genome[idx].to(torch.int64)].cpu().numpy()


def msbwt_rle(genome, cnt, read_len, log=None):
    """RLE bytes (numpy uint8) of the multi-string BWT of the read set {genome[s : s + read_len] : cnt[s] times},
    the symbol totals [$ A C G N T], and the number of reads."""
    import torch
    dev = genome.device
    L = read_len
    g = cnt.numel()
    npos = g + L + 1                      # suffix start positions p = 0 .. g + L
    assert genome.numel() >= npos + KEY_BASES + 1 and L > M_SHORT + 1
    i64 = torch.int64
    say = log or (lambda msg: None)

    # cntp[p + m] = copies of row (p, m): reads starting at p + m - L
    cntp = torch.zeros(npos + L + 1, dtype=torch.uint8, device=dev)
    cntp[L:L + g] = cnt
    n_reads = int(cnt.sum(dtype=i64).item())
    cp = torch.zeros(cntp.numel() + 1, dtype=torch.int32 if n_reads < 2**31 else i64, device=dev)  # cp[i] = sum(cntp[:i])
    torch.cumsum(cntp, 0, dtype=cp.dtype, out=cp[1:])
    # prev[p] = symbol code of genome[p - 1] (prev[0] is never used with a non-zero count)
    codes = torch.tensor(CODE_OF_BASE, dtype=torch.uint8, device=dev)
    prev = torch.empty(npos, dtype=torch.uint8, device=dev)
    prev[0] = codes[0]
    prev[1:] = codes[genome[:npos - 1].to(i64)]
    # key[p] = the 31 bases at p, first base most significant (62 bits): built by doubling
    key = genome[:npos + KEY_BASES].to(i64)
    have = 1
    while have < KEY_BASES:            # key holds `have` bases per position; append the next min(have, rest)
        take = min(have, KEY_BASES - have)
        nxt = key[have:have + (key.numel() - have)]
        if take < have:
            nxt = nxt >> (2 * (have - take))
        key = (key[:nxt.numel()] << (2 * take)) | nxt
        have += take
    key = key[:npos].contiguous()
    say("keys of %d positions ready (%d reads)" % (npos, n_reads))

    out = _RunWriter(torch)
    shift_m = lambda m: 2 * (KEY_BASES - m)

    def level_rows(m):
        """rows of one shallow level over ALL positions, in genome-suffix order: (sorted keys, symbols, copies)"""
        p = _nonzero(torch, cntp[m:m + npos])
        k, order = torch.sort(key[p], stable=True)
        p = p[order]
        return k, prev[p], cntp[p + m].to(i64)

    low = [level_rows(m) for m in range(3)]           # '$', 'x$', 'xy$'
    say("levels 0-2 sorted")

    def low_segment(m, prefix):                       # rows of level m whose first m bases are `prefix`
        k, sym, mult = low[m]
        lo = int(torch.searchsorted(k, torch.tensor([prefix << shift_m(m)], dtype=i64, device=dev))[0].item()) if m else 0
        hi = int(torch.searchsorted(k, torch.tensor([(prefix + 1) << shift_m(m)], dtype=i64, device=dev))[0].item()) if m else k.numel()
        return sym[lo:hi], mult[lo:hi]

    out.add(*low_segment(0, 0))
    top3 = (key >> shift_m(3)).to(torch.uint8)    # leading 3-mer of every position: the 64 groups
    for grp in range(64):
        if grp % 16 == 0:
            out.add(*low_segment(1, grp // 16))
        if grp % 4 == 0:
            out.add(*low_segment(2, grp // 4))
        idx = _nonzero(torch, top3 == grp)
        if idx.numel() == 0:
            continue
        ks, order = torch.sort(key[idx], stable=True)
        sa = idx[order]                                # genome positions of this group in suffix order
        del idx, order
        n = sa.numel()
        wlong = (cp[sa + L] - cp[sa + (M_SHORT + 1)]).to(i64)   # rows with M_SHORT < m < L
        dollars = cntp[sa + L].to(i64)                          # m = L: reads that start here
        keys, jl, mu = [], [], []
        k28 = ks >> shift_m(M_SHORT)
        for m in range(3, M_SHORT + 1):
            c = cntp[sa + m]
            nz = torch.nonzero(c).squeeze(1)
            if nz.numel() == 0:
                continue
            pre = (k28[nz] >> (2 * (M_SHORT - m))) << (2 * (M_SHORT - m))   # first m bases, padded with 'A'
            keys.append((pre << 6) | m)
            jl.append(nz)
            mu.append(c[nz].to(i64))
        if keys:
            keys, perm = torch.sort(torch.cat(keys), stable=True)   # ties (same string): genome-suffix order, as generated
            jl, mu = torch.cat(jl)[perm], torch.cat(mu)[perm]
            ins = torch.searchsorted(k28, keys >> 6, right=False)   # in front of the positions that continue the prefix
            r = keys.numel()
        else:
            jl = mu = ins = torch.zeros(0, dtype=i64, device=dev)
            r = 0
        arange_n = torch.arange(n, dtype=i64, device=dev)
        pile_at = 2 * arange_n + torch.searchsorted(ins, arange_n, right=True)
        short_at = torch.arange(r, dtype=i64, device=dev) + 2 * ins
        sym = torch.empty(r + 2 * n, dtype=torch.uint8, device=dev)
        length = torch.empty(r + 2 * n, dtype=i64, device=dev)
        sym[short_at] = prev[sa[jl]]
        length[short_at] = mu
        sym[pile_at] = prev[sa]
        length[pile_at] = wlong
        sym[pile_at + 1] = DOLLAR
        length[pile_at + 1] = dollars
        out.add(sym, length)
        if grp % 8 == 7:
            say("group %d of 64 done" % (grp + 1))
    rle, totals = out.finish()
    assert int(totals.sum()) == n_reads * (L + 1) and int(totals[DOLLAR]) == n_reads
    return rle, totals, n_reads


# ---- the same for a genome WITH repeats ----------------------------------------------------------------------------------------
# msbwt_rle above leans on the genome being random: 28 bases place a suffix.  A genome with repeat families, segmental duplications,
# satellite arrays and microsatellites (synth.repeat_genome) shares stretches far longer than that, up to and beyond the read length.
# What changes: (1) the genome positions are ordered by their first >= read_len bases (prefix doubling over the 31-mer ranks: 31 ->
# 62 -> 124 -> 248 bases), not by 31; (2) with a[r] = the number of leading bases rank r shares with rank r - 1 (capped at read_len)
# and c[r] = max(a[r], a[r + 1]), the rows (p, m) of rank r with m > max(c[r], 28) are still one run at r -- their strings are unique
# -- while every row with m <= c[r] shares its string's bases with a neighbour and is placed EXPLICITLY, like the short rows: in front
# of rank i = the first rank of the interval of positions that continue its m bases (the last rank <= r with a[i] < m), ordered there
# by (m, r).  [Why (i, m, r) is the order: rows with different i compare like the genome suffixes at i; a row of m bases is a proper
# prefix of everything in its interval that is longer, and '$' is the smallest symbol; equal strings may stand in any fixed order.]


def repeat_read_set(genome_len, read_len, coverage, seed, device="cpu"):
    """Like read_set, over synth.repeat_genome (human repeat classes at their own proportions, copy numbers scaled to the length):
    bases 0..3 with read_len + 256 random spare bases at the end, and cnt[s] = reads starting at s."""
    import torch
    import synth
    dev = torch.device(device)
    codes = synth.repeat_genome(genome_len, seed)                      # symbol codes 1 2 3 5
    base_of_code = np.array([0, 0, 1, 2, 0, 3], dtype=np.uint8)
    n = genome_len + read_len + 256
    genome = torch.empty(n, dtype=torch.uint8, device=dev)
    step = 1 << 28
    for lo in range(0, genome_len, step):
        genome[lo:min(lo + step, genome_len)] = torch.from_numpy(base_of_code[codes[lo:lo + step]]).to(dev)
    del codes
    gen = torch.Generator(device=dev)
    gen.manual_seed(seed)
    genome[genome_len:] = torch.randint(0, 4, (n - genome_len,), generator=gen, device=dev, dtype=torch.uint8)
    cnt = torch.zeros(genome_len, dtype=torch.uint8, device=dev)
    lam = coverage / read_len
    for lo in range(0, genome_len, step):
        m = min(step, genome_len - lo)
        cnt[lo:lo + m] = torch.poisson(torch.full((m,), lam, device=dev, dtype=torch.float32), generator=gen).clamp_(max=255).to(torch.uint8)
    return genome, cnt


PAD_POSITIONS = 192   # positions behind the last suffix that are ordered along with the rest (they anchor nothing: no read reaches them)


def _run_starts(torch, first):
    """first[j]: element j opens a run (first[0] is set) -> for every element the index of its run's first element"""
    starts = torch.nonzero(first).squeeze(1)
    return starts[torch.cumsum(first.to(torch.int64), 0) - 1]


def _common_bases(torch, x, y):
    """leading bases two 31-mer keys share (0..31)"""
    d = x ^ y
    hi, lo = d >> 31, d & 0x7FFFFFFF
    top = torch.where(hi > 0, hi, lo).to(torch.float64)
    msb = torch.frexp(torch.clamp(top, min=1.0))[1].to(torch.int64) - 1 + torch.where(hi > 0, 31, 0)
    return torch.where(d == 0, torch.full_like(msb, KEY_BASES), (2 * KEY_BASES - 1 - msb) >> 1)


def msbwt_rle_repeats(genome, cnt, read_len, log=None):
    """msbwt_rle for ANY genome: RLE bytes, symbol totals, number of reads.  genome: bases 0..3, at least
    cnt.numel() + read_len + 1 + PAD_POSITIONS + 31 of them (repeat_read_set)."""
    import torch
    dev = genome.device
    L = read_len
    g = cnt.numel()
    npos = g + L + 1                      # suffix start positions p = 0 .. g + L
    N = npos + PAD_POSITIONS              # positions that are ordered
    assert genome.numel() >= N + KEY_BASES and L > M_SHORT + 1 and L <= 5 * KEY_BASES
    i64 = torch.int64
    say = log or (lambda msg: None)

    cntp = torch.zeros(N + L + 1, dtype=torch.uint8, device=dev)
    cntp[L:L + g] = cnt
    n_reads = int(cnt.sum(dtype=i64).item())
    cp = torch.zeros(cntp.numel() + 1, dtype=torch.int32 if n_reads < 2**31 else i64, device=dev)
    torch.cumsum(cntp, 0, dtype=cp.dtype, out=cp[1:])
    codes = torch.tensor(CODE_OF_BASE, dtype=torch.uint8, device=dev)
    prev = torch.empty(N, dtype=torch.uint8, device=dev)
    prev[0] = codes[0]
    prev[1:] = codes[genome[:N - 1].to(i64)]
    key = genome[:N + KEY_BASES].to(i64)
    have = 1
    while have < KEY_BASES:
        take = min(have, KEY_BASES - have)
        nxt = key[have:have + (key.numel() - have)]
        if take < have:
            nxt = nxt >> (2 * (have - take))
        key = (key[:nxt.numel()] << (2 * take)) | nxt
        have += take
    key = key[:N].contiguous()
    say("keys of %d positions ready (%d reads)" % (N, n_reads))
    shift_m = lambda m: 2 * (KEY_BASES - m)

    # ---- the positions in the order of their first 31 bases, group by group (leading 3-mer) ----
    top3 = (key >> shift_m(3)).to(torch.uint8)
    sa_all = torch.empty(N, dtype=i64, device=dev)
    rank = torch.empty(N, dtype=i64, device=dev)      # rank[p] = index in sa_all of the first position that equals p as far as compared
    tied = torch.zeros(N, dtype=torch.bool, device=dev)  # per index of sa_all: its position still equals a neighbour
    bounds = [0]
    for grp in range(64):
        idx = _nonzero(torch, top3 == grp)
        n = idx.numel()
        base = bounds[-1]
        bounds.append(base + n)
        if n == 0:
            continue
        ks, order = torch.sort(key[idx], stable=True)
        sa = idx[order]
        del idx, order
        first = torch.ones(n, dtype=torch.bool, device=dev)
        first[1:] = ks[1:] != ks[:-1]
        start = _run_starts(torch, first)
        last = torch.ones(n, dtype=torch.bool, device=dev)
        last[:-1] = first[1:]
        sa_all[base:base + n] = sa
        rank[sa] = base + start
        tied[base:base + n] = ~(first & last)
        del ks, sa, first, last, start
    del top3
    say("31-mer order ready: %d positions equal a neighbour" % int(tied.sum().item()))

    # ---- prefix doubling: ties are ordered by the rank of what follows (31 -> 62 -> 124 -> 248 bases >= read_len) ----
    off = KEY_BASES
    while off < L:
        left = 0
        for grp in range(64):
            base, n = bounds[grp], bounds[grp + 1] - bounds[grp]
            if n == 0:
                continue
            t = torch.nonzero(tied[base:base + n]).squeeze(1)
            if t.numel() == 0:
                continue
            p = sa_all[base + t]
            assert int(p.max().item()) + off < N, "a position in the padding equals another one: another seed, please"
            k2 = ((rank[p] - base) << 32) | rank[p + off]
            k2, perm = torch.sort(k2, stable=True)
            p = p[perm]
            sa_all[base + t] = p
            m = t.numel()
            first = torch.ones(m, dtype=torch.bool, device=dev)
            first[1:] = k2[1:] != k2[:-1]
            start = t[_run_starts(torch, first)]
            last = torch.ones(m, dtype=torch.bool, device=dev)
            last[:-1] = first[1:]
            rank[p] = base + start
            still = ~(first & last)
            tied[base + t] = still
            left += int(still.sum().item())
            del t, p, k2, perm, first, last, start, still
        off *= 2
        say("order by %d bases ready: %d positions still equal a neighbour" % (min(off, 8 * KEY_BASES), left))
    del rank, tied

    out = _RunWriter(torch)

    def level_rows(m):
        p = _nonzero(torch, cntp[m:m + npos])
        k, order = torch.sort(key[p], stable=True)
        p = p[order]
        return k, prev[p], cntp[p + m].to(i64)

    low = [level_rows(m) for m in range(3)]
    say("levels 0-2 sorted")

    def low_segment(m, prefix):
        k, sym, mult = low[m]
        lo = int(torch.searchsorted(k, torch.tensor([prefix << shift_m(m)], dtype=i64, device=dev))[0].item()) if m else 0
        hi = int(torch.searchsorted(k, torch.tensor([(prefix + 1) << shift_m(m)], dtype=i64, device=dev))[0].item()) if m else k.numel()
        return sym[lo:hi], mult[lo:hi]

    out.add(*low_segment(0, 0))
    explicit_long = 0
    for grp in range(64):
        if grp % 16 == 0:
            out.add(*low_segment(1, grp // 16))
        if grp % 4 == 0:
            out.add(*low_segment(2, grp // 4))
        base, n = bounds[grp], bounds[grp + 1] - bounds[grp]
        if n == 0:
            continue
        sa = sa_all[base:base + n]
        ks = key[sa]
        # a[r]: leading bases shared with the rank before (0 for the first of a group: another leading 3-mer), capped at L
        a = torch.zeros(n, dtype=i64, device=dev)
        a[1:] = _common_bases(torch, ks[1:], ks[:-1])
        deep = torch.nonzero(a == KEY_BASES).squeeze(1)
        for t in range(KEY_BASES, L, KEY_BASES):
            if deep.numel() == 0:
                break
            more = _common_bases(torch, key[sa[deep] + t], key[sa[deep - 1] + t])
            a[deep] += more
            deep = deep[more == KEY_BASES]
        a.clamp_(max=L)
        c = a.clone()
        c[:-1] = torch.maximum(c[:-1], a[1:])
        lo_m = c.clamp(min=M_SHORT)                              # rows with m <= lo_m are placed explicitly
        wlong = (cp[sa + L] - cp[sa + torch.clamp(lo_m + 1, max=L)]).to(i64)   # rows with lo_m < m < L
        dollars = torch.where(c >= L, torch.zeros_like(c), cntp[sa + L].to(i64))
        keys, jl, mu, sy = [], [], [], []
        k28 = ks >> shift_m(M_SHORT)
        for m in range(3, M_SHORT + 1):
            cm = cntp[sa + m]
            nz = torch.nonzero(cm).squeeze(1)
            if nz.numel() == 0:
                continue
            pre = (k28[nz] >> (2 * (M_SHORT - m))) << (2 * (M_SHORT - m))   # first m bases, padded with 'A'
            ins = torch.searchsorted(k28, pre, right=False)                 # the first rank that continues them
            keys.append((ins << 8) | m)
            jl.append(nz)
            mu.append(cm[nz].to(i64))
            sy.append(prev[sa[nz]])
        rep = torch.nonzero(c > M_SHORT).squeeze(1)              # ranks inside repeats: more levels are explicit
        if rep.numel():
            a_rep, c_rep, p_rep = a[rep], c[rep], sa[rep]
            for m in range(M_SHORT + 1, int(c_rep.max().item()) + 1):
                sel = torch.nonzero((c_rep >= m) & (cntp[p_rep + m] > 0)).squeeze(1)
                if sel.numel() == 0:
                    continue
                # the interval of the m bases starts at the last rank at or before r that shares fewer than m bases with its predecessor
                # -- itself a member of `rep` (its successor shares m with it), as is the first of `rep` (c of its predecessor <= 28)
                breaks = torch.nonzero(a_rep < m).squeeze(1)
                start = breaks[torch.searchsorted(breaks, sel, right=True) - 1]
                keys.append((rep[start] << 8) | m)
                jl.append(rep[sel])
                mu.append(cntp[p_rep[sel] + m].to(i64))
                sy.append(prev[p_rep[sel]] if m < L else torch.full((sel.numel(),), DOLLAR, dtype=torch.uint8, device=dev))
                explicit_long += int(sel.numel())
            del a_rep, c_rep, p_rep
        if keys:
            keys, perm = torch.sort(torch.cat(keys), stable=True)   # (i, m), then r as generated
            mu, sy = torch.cat(mu)[perm], torch.cat(sy)[perm]
            ins = keys >> 8
            r = keys.numel()
            del jl, perm
        else:
            mu = ins = torch.zeros(0, dtype=i64, device=dev)
            sy = torch.zeros(0, dtype=torch.uint8, device=dev)
            r = 0
        arange_n = torch.arange(n, dtype=i64, device=dev)
        pile_at = 2 * arange_n + torch.searchsorted(ins, arange_n, right=True)
        short_at = torch.arange(r, dtype=i64, device=dev) + 2 * ins
        sym = torch.empty(r + 2 * n, dtype=torch.uint8, device=dev)
        length = torch.empty(r + 2 * n, dtype=i64, device=dev)
        sym[short_at] = sy
        length[short_at] = mu
        sym[pile_at] = prev[sa]
        length[pile_at] = wlong
        sym[pile_at + 1] = DOLLAR
        length[pile_at + 1] = dollars
        out.add(sym, length)
        if grp % 8 == 7:
            say("group %d of 64 done" % (grp + 1))
    rle, totals = out.finish()
    say("%d rows above %d bases were placed one by one" % (explicit_long, M_SHORT))
    assert int(totals.sum()) == n_reads * (L + 1) and int(totals[DOLLAR]) == n_reads
    return rle, totals, n_reads
