"""Seeded random RLE streams for parity tests (test infrastructure)."""
import numpy as np


def runs_to_bytes(syms, lens):
    """(symbol, length) runs -> RLE bytes: base-32 digits, least significant first, zero
    digits included, nothing for a zero-length run (bwt_converter.rs:151-184 semantics)."""
    syms = np.asarray(syms, dtype=np.uint8)
    lens = np.asarray(lens, dtype=np.uint64)
    ndig = 13
    shifts = (np.arange(ndig, dtype=np.uint64) * np.uint64(5))[None, :]
    rest = lens[:, None] >> shifts
    digits = (rest & np.uint64(31)).astype(np.uint8)
    keep = rest > 0
    out = (syms[:, None] | (digits << 3)).astype(np.uint8)
    return out[keep]


def random_runs(rng, nruns, kind="mixed", alphabet=(0, 1, 2, 3, 4, 5)):
    """Runs with no two neighbours of the same symbol."""
    alphabet = np.asarray(alphabet, dtype=np.uint8)
    first = rng.integers(0, len(alphabet))
    step = rng.integers(1, len(alphabet), size=nruns)  # 1..len-1: never the same symbol twice
    idx = (first + np.concatenate([[0], np.cumsum(step[1:])])) % len(alphabet)
    syms = alphabet[idx]
    if kind == "ones":
        lens = np.ones(nruns, dtype=np.uint64)
    elif kind == "short":
        lens = rng.geometric(0.2, size=nruns).astype(np.uint64)
    elif kind == "long":
        lens = rng.integers(1, 5000, size=nruns).astype(np.uint64)
    else:
        pick = rng.integers(0, 10, size=nruns)
        lens = rng.geometric(0.15, size=nruns).astype(np.uint64)
        special = np.array([32, 1024, 32768, 31, 33, 1023, 1025, 255, 256, 257], dtype=np.uint64)
        lens = np.where(pick == 0, special[rng.integers(0, len(special), size=nruns)], lens)
        lens = np.where(pick == 1, rng.integers(1, 100000, size=nruns).astype(np.uint64), lens)
    return syms, lens


def random_stream(seed, nruns, kind="mixed", alphabet=(0, 1, 2, 3, 4, 5)):
    rng = np.random.default_rng(seed)
    syms, lens = random_runs(rng, nruns, kind, alphabet)
    return runs_to_bytes(syms, lens)


def raw_byte_stream(seed, nbytes):
    """Arbitrary valid bytes: any symbol 0..5 with any digit 0..31 -- zero digits, zero-length
    runs and multi-digit runs appear by chance."""
    rng = np.random.default_rng(seed)
    sym = rng.integers(0, 6, size=nbytes).astype(np.uint8)
    # bias towards repeats so multi-byte runs are common
    rep = rng.random(nbytes) < 0.4
    chain = 1
    for i in range(1, nbytes):
        if rep[i] and chain < 3:  # at most 3 digits per run: lengths stay below 32^3
            sym[i] = sym[i - 1]
        chain = chain + 1 if sym[i] == sym[i - 1] else 1
        if chain > 3:
            sym[i] = (sym[i] + 1) % 6
            chain = 1
    digit = rng.integers(0, 32, size=nbytes).astype(np.uint8)
    digit[rng.random(nbytes) < 0.1] = 0
    return (sym | (digit << 3)).astype(np.uint8)


def random_kmers(seed, n, k, alphabet=(1, 2, 3, 5)):
    rng = np.random.default_rng(seed)
    alphabet = np.asarray(alphabet, dtype=np.uint8)
    return alphabet[rng.integers(0, len(alphabet), size=(n, k))]
