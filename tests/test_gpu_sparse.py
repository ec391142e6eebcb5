"""The sparse suffix table (include/msbwt_hip.h, msbwt_rle_set_sparse_table; csrc/sparse_table.hpp) against the CPU oracle:
every entry the device builder wrote, the distinct counts it reports, and the counts of the kernels that look queries up in
it.  Needs an MI355X: run with `pytest -m gpu`."""
import ctypes as C

import numpy as np
import pytest

import rust_msbwt_amd as msbwt
from rust_msbwt_amd import RleBWT, _lib
from oracle import oracle as orc
from rle_random import random_kmers, random_stream

pytestmark = pytest.mark.gpu

ACGT = np.array([1, 2, 3, 5], dtype=np.uint8)


def read_set(seed, genome_len, n_reads, read_len, repeats=0, err=0.0):
    """Reads of a random genome (optionally with a few exact repeats pasted in, so that some suffixes occur often)."""
    rng = np.random.default_rng(seed)
    g = ACGT[rng.integers(0, 4, size=genome_len)]
    for _ in range(repeats):
        src, dst = rng.integers(0, genome_len - 400, size=2)
        g[dst:dst + 300] = g[src:src + 300]
    starts = rng.integers(0, genome_len - read_len, size=n_reads)
    reads = np.stack([g[s:s + read_len] for s in starts])
    if err:
        flip = rng.random(reads.shape) < err
        reads = np.where(flip, ACGT[rng.integers(0, 4, size=reads.shape)], reads)
    return reads


def bwt_of(reads):
    text = ["".join("$ACGNT"[c] for c in r) for r in reads]
    return orc.convert_to_vec(orc.naive_bwt(text))


def synth_bwt(reads):
    """the same by the suffix-sorting builder (read sets too large for the naive one)"""
    import synth
    return synth.rle_encode(synth.build_msbwt_symbols(reads))


def load_pair(rle, monkeypatch, depth, **env):
    monkeypatch.setenv("MSBWT_SEARCH", "lanes")
    monkeypatch.setenv("MSBWT_SPARSE_TABLE", str(depth))
    for k, v in env.items():
        monkeypatch.setenv(k, str(v))
    ref = orc.OracleRleBWT()
    ref.load_vector(rle)
    b = RleBWT()
    b.load_vector(rle)
    return b, ref


def oracle_ranges(ref, kmers):
    """[l, h) of every row of `kmers` by the oracle's own constrain_range, last symbol first (msbwt_core.rs:150-156)."""
    n, k = kmers.shape
    l = np.zeros(n, dtype=np.uint64)
    h = np.full(n, ref.get_total_size(), dtype=np.uint64)
    for t in range(k):
        l, h = ref.constrain_ranges(kmers[:, k - 1 - t], l, h)
    return l, h


def table_key(kmers):
    """The table index of the rows' symbols: A C G T -> 0..3, step t (the t-th symbol from the END) at bits [2t, 2t+2)."""
    codes = (kmers.astype(np.uint64) - 1 - (kmers.astype(np.uint64) >> 2))
    k = kmers.shape[1]
    key = np.zeros(len(kmers), dtype=np.uint64)
    for t in range(k):
        key |= codes[:, k - 1 - t] << np.uint64(2 * t)
    return key


def host_lookup(lines, side, info, key):
    """What the kernel does with one key, on the downloaded table: -> (l, h) or None."""
    b, tag = C.c_uint32(), C.c_uint64()
    assert _lib.lib().msbwt_sparse_hash64(int(key), info["depth"], info["buckets"], C.byref(b), C.byref(tag)) == 0
    bucket = b.value
    wide = info["depth"] >= 25           # sparse_table.hpp: 12 entries with 32-bit tags and a width byte of their own
    xwide = info["depth"] >= 30          # ... 11 entries with 40-bit tags
    nslots = 11 if xwide else 12 if wide else 14
    for dist in range(info["probe"] + 1):
        line = lines[bucket + dist]
        raw = line.view(np.uint8)
        for slot in range(nslots):
            t = int(line[slot])
            if xwide:
                width, hit = int(raw[110 + slot]), (t | (int(raw[88 + slot]) << 32)) == tag.value
                lo = int(line[11 + slot]) | (int(raw[99 + slot]) << 32)
            elif wide:
                width, hit = int(raw[108 + slot]), t == tag.value
                lo = int(line[12 + slot]) | (int(raw[96 + slot]) << 32)
            else:
                width, hit = t >> 24, (t & 0xFFFFFF) == tag.value
                lo = int(line[14 + slot]) | (int(raw[112 + slot]) << 32)
            if width != 0 and hit:
                if width == 255:
                    return int(side[lo][0]), int(side[lo][1])
                return lo, lo + width
        header = int(raw[126]) | (int(raw[127]) << 8)
        if header <= nslots:
            return None
    return None


def slots_in_use(lines, info):
    if info["depth"] >= 30:
        return int((lines.view(np.uint8).reshape(len(lines), 128)[:, 110:121] != 0).sum())
    if info["depth"] >= 25:
        return int((lines.view(np.uint8).reshape(len(lines), 128)[:, 108:120] != 0).sum())
    return int(((lines[:, :14] >> 24) != 0).sum())


@pytest.mark.parametrize("depth", [16, 17, 20, 25, 27, 28, 30, 31])
def test_every_entry_of_the_table_is_the_oracles_range(depth, monkeypatch):
    reads = read_set(11, 3000, 700, 60, repeats=4, err=0.01)
    b, ref = load_pair(bwt_of(reads), monkeypatch, depth)
    assert b.get_sparse_table() == depth and b.get_pair_index()
    info = b.sparse_table_info()
    lines, side = b.download_sparse_table()
    # the suffixes that occur = the distinct ACGT substrings of the reads
    present = np.unique(np.concatenate([np.lib.stride_tricks.sliding_window_view(reads, depth, axis=1).reshape(-1, depth)]), axis=0)
    assert info["entries"] == len(present) == info["distinct"][depth]
    l, h = oracle_ranges(ref, present)
    assert (h > l).all()
    for key, el, eh in zip(table_key(present), l, h):
        assert host_lookup(lines, side, info, key) == (int(el), int(eh))
    # the table is complete and holds nothing else: absent suffixes miss, and the slots in use are exactly the entries
    absent = random_kmers(5, 3000, depth)
    al, ah = oracle_ranges(ref, absent)
    for key, el, eh in zip(table_key(absent), al, ah):
        got = host_lookup(lines, side, info, key)
        assert got == ((int(el), int(eh)) if eh > el else None)
    assert slots_in_use(lines, info) == info["entries"]
    # distinct counts of the shallower levels the build passed through
    for d, n in info["distinct"].items():
        if 4 <= d <= depth:
            assert n == len(np.unique(np.lib.stride_tricks.sliding_window_view(reads, d, axis=1).reshape(-1, d), axis=0)), d


def test_the_automatic_depth_follows_the_declared_query_length(monkeypatch):
    """msbwt_rle_set_query_length: a table of d-mers serves k >= d only, so the AUTOMATIC depth stops at 23 while k is unknown and reaches
    min(k, 27) for a declared k; an explicit depth does not follow the hint; counts never change."""
    reads = read_set(77, 2_000_000, 500_000, 100, repeats=30, err=0.002)
    rle = synth_bwt(reads)
    b, ref = load_pair(rle, monkeypatch, "auto")
    d0 = b.get_sparse_table()
    assert b.get_query_length() == 0 and 16 <= d0 <= 23
    windows = {k: np.ascontiguousarray(np.lib.stride_tricks.sliding_window_view(reads[:300], k, axis=1).reshape(-1, k)) for k in (21, 25, 31, 59)}
    exp = {k: ref.count_kmers(q) for k, q in windows.items()}
    seen = {}
    for hint in (31, 21, 25, 59, 0):
        b.set_query_length(hint)
        assert b.get_query_length() == hint
        seen[hint] = b.get_sparse_table()
        for k, q in windows.items():
            assert np.array_equal(b.count_kmers(q), exp[k]), (hint, k)
    assert seen[0] == d0 and seen[31] == seen[59] and d0 <= seen[31] <= 31 and seen[21] <= 21 and seen[25] <= 25
    assert seen[31] >= 25, seen           # 5e7 symbols, 2e6 distinct 27-mers: deep enough for the wide layout to pay
    b.set_sparse_table(19)
    b.set_query_length(31)
    assert b.get_sparse_table() == 19
    assert np.array_equal(b.count_kmers(windows[31]), exp[31])


@pytest.mark.parametrize("stride", [96, 128])
@pytest.mark.parametrize("depth", [16, 19, 23, 24, 25, 26, 27, 28, 29, 30, 31])
def test_counts_with_the_sparse_table_equal_the_oracle(depth, stride, monkeypatch):
    if depth == 29:   # the last depth 32-bit tags reach needs 2^29 buckets whatever the index: 69 GB (+ 2 GB of slot counters while it is built)
        import torch
        if torch.cuda.mem_get_info(0)[0] < 90 * 10**9:
            pytest.skip("needs 90 GB of free HBM")
    reads = read_set(21 + depth, 5000, 900, 80, repeats=6, err=0.005)
    b, ref = load_pair(bwt_of(reads), monkeypatch, depth, MSBWT_PAIR_STRIDE=stride)
    assert b.get_sparse_table() == depth and b.get_pair_stride() == stride
    rng = np.random.default_rng(depth)
    for k in [depth, depth + 1, depth + 2, 31, 32, 33, 47, 64]:
        if k < depth or k > 80:
            continue
        windows = np.lib.stride_tricks.sliding_window_view(reads, k, axis=1).reshape(-1, k)
        q = np.concatenate([windows[rng.integers(0, len(windows), size=3000)], random_kmers(k, 1500, k)])
        # some queries differ from a present one in a single symbol, early or late
        mut = windows[rng.integers(0, len(windows), size=1500)].copy()
        pos = rng.integers(0, k, size=len(mut))
        mut[np.arange(len(mut)), pos] = ACGT[rng.integers(0, 4, size=len(mut))]
        # and some hold '$' / 'N' inside or outside the table's reach
        odd = windows[rng.integers(0, len(windows), size=600)].copy()
        odd[np.arange(len(odd)), rng.integers(0, k, size=len(odd))] = rng.choice([0, 4], size=len(odd))
        q = np.ascontiguousarray(np.concatenate([q, mut, odd]))
        rng.shuffle(q)
        exp = ref.count_kmers(q)
        assert b.search_kernel_for(k) == "lanes"
        assert np.array_equal(b.count_kmers(q), exp), k
        if k <= 64:
            plain = q[np.isin(q, ACGT).all(axis=1)]
            assert np.array_equal(b.count_kmers_packed(msbwt.rle_bwt.pack_2bit(plain), k), ref.count_kmers(plain)), k
    # the fused read windows, both strands
    k = max(depth, 25)
    fwd, rc = b.count_read_kmers(reads[:200], k, ascii=False, forward=True, revcomp=True)
    windows = np.lib.stride_tricks.sliding_window_view(reads[:200], k, axis=1)
    assert np.array_equal(fwd, ref.count_kmers(windows.reshape(-1, k)).reshape(fwd.shape))
    rcq = np.array([orc.reverse_complement_i(w) for w in windows.reshape(-1, k)], dtype=np.uint8)
    assert np.array_equal(rc, ref.count_kmers(rcq).reshape(rc.shape))
    # the cache policy of the index lines (non-temporal loads, automatic on indexes of 4 GiB and more) changes nothing either
    assert not b.get_line_streaming()
    b.set_line_streaming(1)
    assert b.get_line_streaming()
    assert np.array_equal(b.count_kmers(np.ascontiguousarray(windows.reshape(-1, k)[:6000])), ref.count_kmers(np.ascontiguousarray(windows.reshape(-1, k)[:6000])))
    b.set_line_streaming(-1)
    # switching the table off (and on again) changes nothing but the table
    b.set_sparse_table(0)
    assert b.get_sparse_table() == 0
    q = np.ascontiguousarray(windows.reshape(-1, k)[:4000])
    assert np.array_equal(b.count_kmers(q), ref.count_kmers(q))
    b.set_sparse_table(depth)
    assert b.get_sparse_table() == depth
    assert np.array_equal(b.count_kmers(q), ref.count_kmers(q))


def test_wide_entries_go_through_the_side_array_and_full_buckets_displace(monkeypatch):
    """Suffixes of a high-copy repeat (range 255 or more wide) keep their range in the side array; with enough entries some
    buckets overflow and their keys are found a bucket later -- both by the kernel's own counters."""
    rng = np.random.default_rng(3)
    genome = ACGT[rng.integers(0, 4, size=40000)]
    unit = ACGT[rng.integers(0, 4, size=40)]
    reads = [genome[s:s + 50] for s in rng.integers(0, len(genome) - 50, size=2500)]
    reads += [np.concatenate([unit, unit])[o:o + 50] for o in rng.integers(0, 30, size=600)]  # 600 copies of one 40-mer's rotations
    reads = np.stack(reads)
    b, ref = load_pair(bwt_of(reads), monkeypatch, 16)
    info = b.sparse_table_info()
    assert info["side_entries"] > 0 and info["side_entries"] == info["wide"][16]
    assert info["displaced"] > 0, info
    k = 31
    windows = np.lib.stride_tricks.sliding_window_view(reads, k, axis=1).reshape(-1, k)
    # (many more tiles than resident waves -- 3072 -- so that most tiles are prepared while another one is searched)
    q = np.ascontiguousarray(np.concatenate([windows] * 12 + [random_kmers(9, 20000, k)]))
    b.set_search_counters(True)
    got = b.count_kmers(q)
    cnt = b.search_counters(0)
    assert np.array_equal(got, ref.count_kmers(q))
    assert got.max() >= 255
    assert cnt["table_rides"] > 0.25 * 12 * len(windows)  # most lookups ride along with the search of the tile before theirs
    assert cnt["escape_queries"] > 0 and cnt["table_displaced"] > 0 and cnt["table_steps"] >= 12 * len(windows)  # (random k-mers mostly end in the presence filter)
    # every query that entered the search took exactly one lookup plus ceil((k - 16) / 2) pair steps at most
    assert cnt["lane_steps"] <= cnt["table_steps"] + cnt["searched"] * 8


def test_arbitrary_symbol_streams_and_replicas(monkeypatch):
    """Not a BWT at all: a random run stream over all six symbols -- the table is whatever backward search says, like the oracle."""
    rle = random_stream(77, 30000, "short")
    b, ref = load_pair(rle, monkeypatch, 16)
    assert b.get_sparse_table() == 16
    for k in (16, 21, 31, 40):
        q = np.ascontiguousarray(np.concatenate([random_kmers(k, 20000, k), random_kmers(k + 1, 3000, k, alphabet=(0, 1, 2, 3, 4, 5))]))
        assert np.array_equal(b.count_kmers(q), ref.count_kmers(q)), k
    twin = b.replicate(b.device_ordinal())
    assert twin.get_sparse_table() == 16 and twin.sparse_table_info()["entries"] == b.sparse_table_info()["entries"]
    q = np.ascontiguousarray(random_kmers(5, 30000, 24))
    assert np.array_equal(twin.count_kmers(q), ref.count_kmers(q))
    assert twin.device_bytes() == b.device_bytes()


def test_automatic_depth_follows_the_data_and_short_queries_use_the_direct_table(monkeypatch):
    reads = read_set(5, 20000, 6000, 100, err=0.005)
    monkeypatch.setenv("MSBWT_SEARCH", "auto")
    monkeypatch.delenv("MSBWT_SPARSE_TABLE", raising=False)
    rle = bwt_of(reads)
    ref = orc.OracleRleBWT()
    ref.load_vector(rle)
    b = RleBWT()
    b.load_vector(rle)
    depth = b.get_sparse_table()
    info = b.sparse_table_info()
    assert 16 <= depth <= 23 and info["bytes"] <= 64 << 20, info  # a toy index gets a toy table
    assert b.get_table_depth() <= 15  # the direct table beside it stays small
    for k in (8, depth - 1, depth, 31):
        windows = np.lib.stride_tricks.sliding_window_view(reads, k, axis=1).reshape(-1, k)
        q = np.ascontiguousarray(np.concatenate([windows[::7], random_kmers(k, 5000, k)]))
        assert np.array_equal(b.count_kmers(q), ref.count_kmers(q)), k
    with pytest.raises(msbwt.MsbwtError):
        b.set_sparse_table(12)


def test_when_no_depth_fits_the_direct_table_is_built_instead_and_the_counts_stay_on_record(monkeypatch):
    """A memory budget that leaves the pair blocks but no room for any sparse depth: the loader builds the direct table the plan allows,
    the handle has no sparse table, the distinct counts of the sizing pass are still reported, counts equal the oracle's."""
    monkeypatch.setenv("MSBWT_SEARCH", "auto")
    monkeypatch.delenv("MSBWT_SPARSE_TABLE", raising=False)
    reads = read_set(8, 60000, 20000, 100, err=0.005)
    rle = bwt_of(reads)
    ref = orc.OracleRleBWT()
    ref.load_vector(rle)
    b = RleBWT()
    b.load_vector(rle)
    assert b.get_sparse_table() >= 16 and b.get_pair_index()
    full = b.device_bytes()
    planes = (b.get_total_size() // 256 + 1) * 128
    found = None
    for budget in range(full, planes, -(64 << 10)):   # downwards in 64 KiB steps until the sparse table no longer fits beside the pair blocks
        b.set_memory_budget(budget)
        if b.get_pair_index() and b.get_sparse_table() == 0:
            found = budget
            break
        if not b.get_pair_index():
            break
    assert found is not None, "no budget keeps the pair blocks without a sparse table"
    info = b.sparse_table_info()
    assert info["depth"] == 0 and info["entries"] == 0 and len(info["distinct"]) >= 2 and max(info["distinct"].values()) > 10000, info
    assert b.device_bytes() <= found + (4 << 20)
    for k in (12, 25, 31):
        windows = np.lib.stride_tricks.sliding_window_view(reads, k, axis=1).reshape(-1, k)
        q = np.ascontiguousarray(np.concatenate([windows[::97], random_kmers(k, 3000, k)]))
        assert np.array_equal(b.count_kmers(q), ref.count_kmers(q)), k
    b.set_memory_budget(0)
    assert b.get_sparse_table() >= 16 and b.device_bytes() == full


# ---- the two-tier form (round 6; csrc/sparse_table.hpp): entries for the suffixes that occur at least twice, filter bits for the rest ----
def tier_lookup(lines, side, info, key):
    """What the kernel does with one key on a downloaded TWO-TIER table: -> ("entry", l, h) | ("filter",) | None (count 0)."""
    b, tag = C.c_uint32(), C.c_uint64()
    assert _lib.lib().msbwt_sparse_hash64(int(key), info["depth"], info["buckets"], C.byref(b), C.byref(tag)) == 0
    wide = info["depth"] >= 25
    nslots = 9 if wide else 10
    word, mask = C.c_uint32(), C.c_uint32()
    assert _lib.lib().msbwt_sparse_filter_bits(tag.value, C.byref(word), C.byref(mask)) == 0
    assert 0 <= word.value < 8 and bin(mask.value).count("1") <= 4 and mask.value != 0
    home = lines[b.value]
    maybe = (int(home[23 + word.value]) & mask.value) == mask.value
    for dist in range(info["probe"] + 1):
        line = lines[b.value + dist]
        raw = line.view(np.uint8)
        for slot in range(nslots):
            t = int(line[slot])
            if wide:
                width, hit, lo = int(raw[81 + slot]), t == tag.value, int(line[9 + slot]) | (int(raw[72 + slot]) << 32)
            else:
                width, hit, lo = t >> 24, (t & 0xFFFFFF) == tag.value, int(line[10 + slot]) | (int(raw[80 + slot]) << 32)
            if width != 0 and hit:
                assert width >= 2, "a suffix that occurs once has no entry"
                return ("entry",) + ((int(side[lo][0]), int(side[lo][1])) if width == 255 else (lo, lo + width))
        header = int(raw[90]) | (int(raw[91]) << 8)
        if header <= nslots:
            break
    return ("filter",) if maybe else None


def tier_slots_in_use(lines, info):
    raw = lines.view(np.uint8).reshape(len(lines), 128)
    return int((raw[:, 81:90] != 0).sum()) if info["depth"] >= 25 else int(((lines[:, :10] >> 24) != 0).sum())


@pytest.mark.parametrize("depth", [16, 17, 20, 24, 25, 27, 28])
def test_two_tier_table_holds_the_solid_suffixes_and_filters_the_ones_that_occur_once(depth, monkeypatch):
    """MSBWT_SPARSE_TIERS=1: every suffix that occurs at least twice has an entry with the oracle's range; every suffix that occurs once has
    NO entry and its four filter bits set in its own bucket; an absent suffix is a miss or (rarely) a filter false positive; the slots
    in use are exactly the solid suffixes."""
    reads = read_set(31, 3000, 700, 60, repeats=4, err=0.02)
    b, ref = load_pair(bwt_of(reads), monkeypatch, depth, MSBWT_SPARSE_TIERS=1)
    assert b.get_sparse_table() == depth and b.get_sparse_tiers() and b.get_pair_index()
    info = b.sparse_table_info()
    assert info["two_tier"]
    lines, side = b.download_sparse_table()
    present = np.unique(np.lib.stride_tricks.sliding_window_view(reads, depth, axis=1).reshape(-1, depth), axis=0)
    l, h = oracle_ranges(ref, present)
    once = (h - l) == 1
    assert once.sum() > 100 and (~once).sum() > 100          # reads with errors: both kinds
    assert info["distinct"][depth] == len(present) and info["once"][depth] == int(once.sum()) == info["filtered"]
    assert info["entries"] == int((~once).sum()) == tier_slots_in_use(lines, info)
    for key, el, eh in zip(table_key(present), l, h):
        got = tier_lookup(lines, side, info, key)
        assert got == (("filter",) if eh - el == 1 else ("entry", int(el), int(eh)))
    absent = random_kmers(5, 4000, depth)
    al, ah = oracle_ranges(ref, absent)
    false_pos = 0
    for key, el, eh in zip(table_key(absent), al, ah):
        got = tier_lookup(lines, side, info, key)
        if eh > el:
            assert got == (("filter",) if eh - el == 1 else ("entry", int(el), int(eh)))
        else:
            assert got in (None, ("filter",))
            false_pos += got is not None
    assert false_pos < 0.1 * len(absent)
    # the shallower levels' counts of suffixes that occur once (the level the expansion starts from is only seeded, not examined)
    for d, n in info["once"].items():
        if max(4, info["parent_depth"] + 1) <= d <= depth and d in info["distinct"]:
            w = np.lib.stride_tricks.sliding_window_view(reads, d, axis=1).reshape(-1, d)
            _, cnt = np.unique(w, axis=0, return_counts=True)
            assert n == int((cnt == 1).sum()), d


@pytest.mark.parametrize("direct", ["packed", "flat", "none", "deep"])
@pytest.mark.parametrize("stride", [96, 128])
@pytest.mark.parametrize("depth", [16, 19, 23, 25, 28])
def test_counts_with_the_two_tier_table_equal_the_oracle(depth, stride, direct, monkeypatch):
    """Solid, once-only, absent and mutated k-mers (counts n / 1 / 0 exact) through the two-tier table: a lookup that ends in the filter goes
    on through the direct table -- packed, flat, or none at all (from [0, total)) -- and the search."""
    if direct != "packed" and (stride == 128 or depth in (19, 28)):
        pytest.skip("the direct-table variants are covered at stride 96, depths 16 / 23 / 25")
    if direct == "deep" and depth != 23:
        pytest.skip("the deep direct table (73 GB) once")
    env = {"MSBWT_SPARSE_TIERS": 1, "MSBWT_PAIR_STRIDE": stride}
    if direct == "deep":   # packed depth 17, what a chr20-sized index keeps beside its sparse table: its index takes 34 bits
        import torch
        if torch.cuda.mem_get_info(0)[0] < 120 * 10**9:
            pytest.skip("needs 120 GB of free HBM")
        env.update({"MSBWT_TABLE_DEPTH": 15, "MSBWT_TABLE_PACKED": 1})
    if direct == "flat":
        env["MSBWT_TABLE_PACKED"] = 0
    if direct == "none":
        env["MSBWT_TABLE_DEPTH"] = 0
    reads = read_set(121 + depth, 5000, 900, 80, repeats=6, err=0.01)
    b, ref = load_pair(bwt_of(reads), monkeypatch, depth, **env)
    assert b.get_sparse_table() == depth and b.get_sparse_tiers() and b.get_pair_stride() == stride
    assert (b.get_table_depth() == 0) == (direct == "none") and (b.get_table_packed() if direct in ("packed", "deep") else not b.get_table_packed())
    assert direct != "deep" or b.get_table_depth() == 17
    rng = np.random.default_rng(depth)
    for k in [depth, depth + 1, depth + 2, 31, 32, 33, 47, 64]:
        if k < depth or k > 80:
            continue
        windows = np.lib.stride_tricks.sliding_window_view(reads, k, axis=1).reshape(-1, k)
        q = np.concatenate([windows[rng.integers(0, len(windows), size=4000)], random_kmers(k, 1500, k)])
        mut = windows[rng.integers(0, len(windows), size=2000)].copy()
        pos = rng.integers(0, k, size=len(mut))
        mut[np.arange(len(mut)), pos] = ACGT[rng.integers(0, 4, size=len(mut))]
        odd = windows[rng.integers(0, len(windows), size=600)].copy()
        odd[np.arange(len(odd)), rng.integers(0, k, size=len(odd))] = rng.choice([0, 4], size=len(odd))
        q = np.ascontiguousarray(np.concatenate([q, mut, odd]))
        rng.shuffle(q)
        exp = ref.count_kmers(q)
        assert (exp == 1).sum() > 200 and (exp == 0).sum() > 200 and (exp > 1).sum() > 200
        b.set_search_counters(True)
        got = b.count_kmers(q)
        cnt = b.search_counters(0)
        b.set_search_counters(False)
        assert np.array_equal(got, exp), k
        assert cnt["tier_fallbacks"] > 100, cnt     # the once-only suffixes went down the direct table's path
        if k <= 64:
            plain = q[np.isin(q, ACGT).all(axis=1)]
            assert np.array_equal(b.count_kmers_packed(msbwt.rle_bwt.pack_2bit(plain), k), ref.count_kmers(plain)), k
    # many more tiles than resident waves: most lookups ride along with the search of the tile before theirs (the other way into the filter)
    k = 31
    windows = np.lib.stride_tricks.sliding_window_view(reads, k, axis=1).reshape(-1, k)
    q = np.ascontiguousarray(np.concatenate([windows] * 12 + [random_kmers(9, 20000, k)]))
    b.set_search_counters(True)
    got = b.count_kmers(q)
    cnt = b.search_counters(0)
    b.set_search_counters(False)
    assert np.array_equal(got, ref.count_kmers(q))
    assert cnt["table_rides"] > 0.1 * 12 * len(windows) and cnt["tier_fallbacks"] > 1000, cnt
    # the fused read windows, both strands
    k = max(depth, 25)
    fwd, rc = b.count_read_kmers(reads[:200], k, ascii=False, forward=True, revcomp=True)
    windows = np.lib.stride_tricks.sliding_window_view(reads[:200], k, axis=1)
    assert np.array_equal(fwd, ref.count_kmers(windows.reshape(-1, k)).reshape(fwd.shape))
    rcq = np.array([orc.reverse_complement_i(w) for w in windows.reshape(-1, k)], dtype=np.uint8)
    assert np.array_equal(rc, ref.count_kmers(rcq).reshape(rc.shape))
    # the complete table of the same depth, then the automatic choice (complete: everything fits here), then two-tier again: counts never change
    q = np.ascontiguousarray(windows.reshape(-1, k)[:6000])
    exp = ref.count_kmers(q)
    for mode, tiers in ((0, False), (-1, False), (1, True)):
        b.set_sparse_tiers(mode)
        assert b.get_sparse_table() == depth and b.get_sparse_tiers() == tiers
        assert np.array_equal(b.count_kmers(q), exp), mode
    twin = b.replicate(b.device_ordinal())
    assert twin.get_sparse_tiers() and np.array_equal(twin.count_kmers(q), exp)


def test_two_tier_table_with_high_copy_suffixes_and_escape_lines_of_the_direct_table(monkeypatch):
    """Repeats: wide entries of the two-tier table (side array) AND escape lines of the packed direct table its filter sends queries to
    (a shallow packed table over 4e6 symbols: nearly every line has a delta beyond 16 bits)."""
    import synth
    genome = synth.repeat_genome(200_000, 5)
    reads = synth.reads(genome, 27_000, 150, 6, 0.005)
    rle = synth.rle_encode(synth.build_msbwt_symbols(reads))
    b, ref = load_pair(rle, monkeypatch, 16, MSBWT_SPARSE_TIERS=1, MSBWT_TABLE_DEPTH=3, MSBWT_TABLE_PACKED=1)
    info, tinfo = b.sparse_table_info(), b.table_info()
    assert b.get_sparse_tiers() and info["side_entries"] > 0 and b.get_table_packed() and b.get_table_depth() == 5
    assert tinfo["escape_lines"] >= 33 and tinfo["side_bytes"] == tinfo["escape_lines"] * 512
    for k in (16, 31, 40):
        q = np.ascontiguousarray(np.concatenate([synth.read_kmers(reads, k, limit=150_000, seed=k), random_kmers(9, 20000, k)]))
        b.set_search_counters(True)
        got = b.count_kmers(q)
        cnt = b.search_counters(0)
        b.set_search_counters(False)
        exp = ref.count_kmers(q)
        assert np.array_equal(got, exp), k
        assert (exp == 1).sum() > 1000 and exp.max() >= 255
        # lookups that ended in the filter read a line of the direct table -- an escape line more often than not -- and then its side entry
        assert cnt["tier_fallbacks"] > 1000 and cnt["escape_queries"] > 1000, cnt
    # without the direct table's side array the two-tier form is not built (an escape line could not be followed from the filter's path)
    b.set_table_side(0)
    assert not b.get_sparse_tiers() and b.get_sparse_table() == 16
    q = synth.read_kmers(reads, 31, limit=50_000, seed=3)
    assert np.array_equal(b.count_kmers(q), ref.count_kmers(q))


@pytest.mark.parametrize("tiers", [0, 1])
def test_run_blocks_behind_a_sparse_table(tiers, monkeypatch):
    """MSBWT_BLOCKS=runs (the memory-lean format) with a sparse table (round 6): the table is built at load time from temporary plane and
    pair blocks that are freed again; lookups continue with single-symbol steps over the run blocks."""
    monkeypatch.setenv("MSBWT_BLOCKS", "runs")
    reads = read_set(91, 20000, 5000, 90, repeats=10, err=0.01)
    b, ref = load_pair(bwt_of(reads), monkeypatch, 19, MSBWT_SPARSE_TIERS=tiers)
    assert b.get_block_format() == "runs" and not b.get_pair_index()
    assert b.get_sparse_table() == 19 and b.get_sparse_tiers() == bool(tiers)
    info = b.sparse_table_info()
    present = np.unique(np.lib.stride_tricks.sliding_window_view(reads, 19, axis=1).reshape(-1, 19), axis=0)
    assert info["distinct"][19] == len(present)
    # the table costs exactly its own bytes on top of the run blocks and their direct table
    monkeypatch.setenv("MSBWT_SPARSE_TABLE", "0")
    lean = RleBWT()
    lean.load_vector(bwt_of(reads))
    assert lean.get_sparse_table() == 0 and b.device_bytes() == lean.device_bytes() + info["bytes"] + info["side_bytes"]
    rng = np.random.default_rng(1)
    for k in (8, 18, 19, 20, 31, 33, 64):
        windows = np.lib.stride_tricks.sliding_window_view(reads, k, axis=1).reshape(-1, k)
        mut = windows[rng.integers(0, len(windows), size=3000)].copy()
        mut[np.arange(len(mut)), rng.integers(0, k, size=len(mut))] = ACGT[rng.integers(0, 4, size=len(mut))]
        odd = windows[rng.integers(0, len(windows), size=500)].copy()
        odd[np.arange(len(odd)), rng.integers(0, k, size=len(odd))] = rng.choice([0, 4], size=len(odd))
        q = np.ascontiguousarray(np.concatenate([windows[::3], random_kmers(k, 5000, k), mut, odd]))
        b.set_search_counters(True)
        got = b.count_kmers(q)
        cnt = b.search_counters(0)
        b.set_search_counters(False)
        assert np.array_equal(got, ref.count_kmers(q)), k
        assert np.array_equal(lean.count_kmers(q), got), k
        if k >= 19:
            assert cnt["table_steps"] > 0 and cnt["pair_steps"] == 0, cnt
            assert bool(cnt["tier_fallbacks"]) == bool(tiers), cnt
    fwd, rc = b.count_read_kmers(reads[:300], 31, ascii=False, forward=True, revcomp=True)
    windows = np.lib.stride_tricks.sliding_window_view(reads[:300], 31, axis=1)
    assert np.array_equal(fwd, ref.count_kmers(windows.reshape(-1, 31)).reshape(fwd.shape))
    rcq = np.array([orc.reverse_complement_i(w) for w in windows.reshape(-1, 31)], dtype=np.uint8)
    assert np.array_equal(rc, ref.count_kmers(rcq).reshape(rc.shape))
    twin = b.replicate(b.device_ordinal())
    q = np.ascontiguousarray(windows.reshape(-1, 31)[:5000])
    assert twin.get_sparse_table() == 19 and np.array_equal(twin.count_kmers(q), ref.count_kmers(q))
    # the table goes when asked to (it cannot come back without a load: the plane blocks it was built from are gone)
    b.set_sparse_table(0)
    assert b.get_sparse_table() == 0 and b.device_bytes() == lean.device_bytes()
    assert np.array_equal(b.count_kmers(q), ref.count_kmers(q))


def test_second_sparse_level_serves_the_queries_the_first_is_too_deep_for(monkeypatch):
    """k undeclared: the automatic table is up to 23 deep and serves k >= its depth; where the deep direct table is not kept beside it, a second
    table of the 17-symbol suffixes serves 17 <= k < depth (msbwt_rle_set_sparse_second; sparse_for in csrc/kernels.hpp).  Counts never change."""
    reads = read_set(61, 2_000_000, 400_000, 100, repeats=20, err=0.004)
    rle = synth_bwt(reads)
    # (an explicit direct-table depth: the loader then does not consider keeping the deep direct table, as it would on an index this small)
    b, ref = load_pair(rle, monkeypatch, "auto", MSBWT_TABLE_DEPTH=9)
    info = b.sparse_table_info()
    assert info["depth"] >= 19 and info["second_depth"] == 17 and info["second_bytes"] > 0 and b.get_query_length() == 0, info
    with_second = b.device_bytes()
    exp, qs = {}, {}
    for k in (12, 16, 17, 18, info["depth"] - 1, info["depth"], 31, 40):
        windows = np.lib.stride_tricks.sliding_window_view(reads[:2000], k, axis=1).reshape(-1, k)
        qs[k] = np.ascontiguousarray(np.concatenate([windows[::5], random_kmers(k, 4000, k)]))
        exp[k] = ref.count_kmers(qs[k])
        b.set_search_counters(True)
        got = b.count_kmers(qs[k])
        cnt = b.search_counters(0)
        b.set_search_counters(False)
        assert np.array_equal(got, exp[k]), k
        assert (cnt["table_steps"] > 0) == (k >= 17), (k, cnt)      # looked up in one of the two sparse tables, or in the direct table
    # a declared k gets no second level; switched off it goes; automatic again it comes back
    b.set_query_length(31)
    assert b.sparse_table_info()["second_depth"] == 0
    b.set_query_length(0)
    assert b.sparse_table_info()["second_depth"] == 17 and b.device_bytes() == with_second
    b.set_sparse_second(0)
    info0 = b.sparse_table_info()
    assert info0["second_depth"] == 0 and info0["depth"] == info["depth"] and b.device_bytes() == with_second - info["second_bytes"]
    for k, q in qs.items():
        assert np.array_equal(b.count_kmers(q), exp[k]), k
    b.set_sparse_second(-1)
    assert b.sparse_table_info()["second_depth"] == 17
    twin = b.replicate(b.device_ordinal())
    assert twin.sparse_table_info()["second_depth"] == 17 and np.array_equal(twin.count_kmers(qs[18]), exp[18])
