"""The CPU oracle against every known-answer vector the reference's tests hold for the
RleBWT count_kmer path.  CPU only."""
import itertools
import os

import numpy as np
import pytest

from conftest import expand_case
from oracle import oracle as orc

CODES = {"$": 0, "A": 1, "C": 2, "G": 3, "N": 4, "T": 5}


def stoi(s):
    return [CODES[c] for c in s]


def test_g1_convert_to_vec(golden):
    for case in golden["G1_convert_to_vec"]["cases"]:
        got = orc.convert_to_vec(expand_case(case))
        if "bytes" in case:
            assert got.tolist() == case["bytes"]
        else:
            assert len(got) == case["len"]


def test_convert_to_vec_rejects_other_symbols():
    with pytest.raises(orc.OracleError):
        orc.convert_to_vec("ACGX")


def test_g2_npy_bytes(golden, tmp_path):
    g = golden["G2_npy"]
    for i, case in enumerate(g["cases"]):
        path = str(tmp_path / ("g2_%d.npy" % i))
        if case["kind"] == "bytes":
            orc.save_bwt_numpy(orc.convert_to_vec(expand_case(case)), path)
        else:
            orc.save_bwt_runs_numpy(case["runs"], path)
        raw = open(path, "rb").read()
        n = len(case["payload"])
        head = bytes.fromhex(g["magic_hex"]) + (g["header_text_before_len"] + str(n) + g["header_tail"]).encode()
        head = head + b" " * (95 - len(head)) + b"\n"
        assert len(head) == g["header_total_len"]
        assert raw == head + bytes(case["payload"])
        # and it loads back
        b = orc.OracleRleBWT()
        b.load_numpy_file(path)
        assert b.get_total_size() == 3104 + (1 if case["kind"] == "runs" else 0)


def test_g3_naive_bwt(golden):
    for case in golden["G3_naive_bwt"]["cases"]:
        assert orc.naive_bwt(case["strings"]) == case["bwt"]


def test_g4_totals_via_npy_roundtrip(golden, tmp_path):
    g = golden["G4_totals"]
    path = str(tmp_path / "g4.npy")
    orc.save_bwt_numpy(orc.convert_to_vec(orc.naive_bwt(g["strings"])), path)
    b = orc.OracleRleBWT()
    b.load_numpy_file(path)
    assert [b.get_symbol_count(s) for s in range(6)] == g["symbol_counts"]


def test_g5_sampled_index(golden):
    g = golden["G5_sampled_index"]
    comp = orc.convert_to_vec(g["bwt"])
    assert len(comp) == 8
    for bp, exp in g["by_bin_power"].items():
        b = orc.OracleRleBWT(int(bp))
        b.load_vector(comp)
        n = -(-len(g["bwt"]) // (1 << int(bp))) + 1
        assert len(exp["ref"]) == n
        assert b.ref_index() == exp["ref"]
        for s in range(6):
            assert b.fm_index(s) == exp["fm"][s]


def test_g6_constrain_range_exhaustive(golden):
    g = golden["G6_constrain_range"]
    text = g["bwt"]
    ints = stoi(text)
    comp = orc.convert_to_vec(text)
    for bp in g["bin_powers"]:
        b = orc.OracleRleBWT(bp)
        b.load_vector(comp)
        start, end = b.start_index(), b.end_index()
        for sym in range(6):
            assert b.constrain_range(sym, 0, len(text)) == (start[sym], end[sym])
            cnt = 0
            for ind in range(len(text) + 1):
                assert b.constrain_range(sym, 0, ind) == (start[sym], start[sym] + cnt)
                assert b.constrain_range(sym, ind, len(text)) == (start[sym] + cnt, end[sym])
                if ind < len(text) and ints[ind] == sym:
                    cnt += 1


def _check_count_kmer_block(g):
    comp = orc.convert_to_vec(orc.naive_bwt(g["strings"]))
    for bp in g.get("bin_powers", [1, 2, 3, 4, 8]):
        b = orc.OracleRleBWT(bp)
        b.load_vector(comp)
        for c in range(6):
            assert b.count_kmer([c]) == b.get_symbol_count(c)
        for s in g["strings"]:
            assert b.count_kmer(stoi(s)) == 1
        for kmer, n in g["counts"].items():
            assert b.count_kmer(stoi(kmer)) == n


def test_g7_count_kmer(golden):
    _check_count_kmer_block(golden["G7_count_kmer"])


def test_g10_four_strings(golden):
    _check_count_kmer_block(golden["G10_load_and_add"])


def test_g8_doc_tests(golden):
    g = golden["G8_doc_tests"]
    b = orc.OracleRleBWT()
    b.load_vector(orc.convert_to_vec(g["bwt"]))
    for codes, n in g["count_codes"]:
        assert b.count_kmer(codes) == n
    for text, n in g["count_text"]:
        assert b.count_kmer(orc.convert_stoi(text)) == n
    assert b.get_symbol_count(0) == g["symbol_count_0"]
    assert b.get_total_size() == g["total_size"]


def test_g9_two_string_fixture(golden, golden_dir):
    g = golden["G9_two_string"]
    path = os.path.join(golden_dir, g["file"])
    assert list(open(path, "rb").read()[96:]) == g["payload"]
    b = orc.OracleRleBWT()
    b.load_numpy_file(path)
    for kmer, n in g["counts"].items():
        assert b.count_kmer(orc.convert_stoi(kmer)) == n
    # config C1 (BASELINE.json configs[0]): all 4-mers; only the ten rotations are present
    present = {"$ACG", "ACGT", "CGT$", "GT$A", "T$AC", "$TGC", "TGCA", "GCA$", "CA$T", "A$TG"}
    syms = "$ACGNT"
    for tup in itertools.product(range(6), repeat=4):
        text = "".join(syms[c] for c in tup)
        assert b.count_kmer(list(tup)) == (1 if text in present else 0), text


def test_string_util(golden):
    g = golden["string_util"]
    for text, codes in g["stoi"]:
        assert orc.convert_stoi(text).tolist() == codes
    for codes, text in g["itos"]:
        assert orc.convert_itos(codes) == text
    for codes, rc in g["revcomp"]:
        assert orc.reverse_complement_i(codes).tolist() == rc
    assert orc.convert_stoi("acgtnXx-").tolist() == [1, 2, 3, 5, 4, 4, 4, 4]


def test_runblock_count_doc_test(golden):
    g = golden["runblock_count"]
    data = []
    for i, v in enumerate(g["data"]):
        # runs of the symbols inserted so far (RLEBlock stores 16-bit runs: sym | len<<3)
        runs = []
        for s in data:
            if runs and (runs[-1] & 7) == s:
                runs[-1] += 8
            else:
                runs.append(s | 8)
        assert orc.runblock_count(runs, i, v) == g["expected_count_before_insert"][i]
        data.append(v)


def test_npy_loader_errors(tmp_path, golden_dir):
    good = open(os.path.join(golden_dir, "two_string.npy"), "rb").read()
    cases = {
        "missing": (None, orc.ERR_IO),
        "short_fixed": (good[:7], orc.ERR_HEADER),          # rle_bwt.rs:91-93 panic
        "short_header": (good[:50], orc.ERR_EOF),           # :102-112 read_exact
        "truncated": (good[:-1], orc.ERR_EOF),              # :129-136
        "extra": (good + b"\x08", orc.ERR_EOF),
        "not_json": (good[:10] + b"{'descr' '|u1'}".ljust(86) + good[96:], orc.ERR_HEADER),
        "no_shape": (good[:10] + b"{'descr': '|u1', }".ljust(85) + b"\n" + good[96:], orc.ERR_HEADER),
        "fortran_true": (good.replace(b"False", b"True "), orc.ERR_HEADER),
    }
    for name, (blob, code) in cases.items():
        path = str(tmp_path / (name + ".npy"))
        if blob is not None:
            open(path, "wb").write(blob)
        b = orc.OracleRleBWT()
        with pytest.raises(orc.OracleError) as e:
            b.load_numpy_file(path)
        assert e.value.code == code, name
    # a numpy-written file (64-byte aligned header, "(10,)") loads too
    path = str(tmp_path / "np.npy")
    np.save(path, np.frombuffer(good[96:], dtype=np.uint8))
    b = orc.OracleRleBWT()
    b.load_numpy_file(path)
    assert b.count_kmer(orc.convert_stoi("ACGT")) == 1


def test_empty_bwt():
    b = orc.OracleRleBWT()
    b.load_vector(np.zeros(0, dtype=np.uint8))
    assert b.get_total_size() == 0
    assert b.ref_index() == [0]
    assert b.count_kmer([1, 2]) == 0
    assert b.count_kmer([]) == 0


def test_invalid_symbol_and_range():
    b = orc.OracleRleBWT()
    b.load_vector(orc.convert_to_vec("GTN$$ACCC$G"))
    with pytest.raises(orc.OracleError) as e:
        b.count_kmer([1, 6])
    assert e.value.code == orc.ERR_SYMBOL
    with pytest.raises(orc.OracleError) as e:
        b.constrain_range(1, 0, 12)
    assert e.value.code == orc.ERR_RANGE
    assert b.count_kmer([]) == 11  # empty k-mer => total size (msbwt_core.rs:128-131,160)
