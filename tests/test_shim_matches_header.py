"""The Rust shim cannot be compiled in this image; this keeps its `extern "C"` block --
in shim/msbwt2-hip/src/lib.rs and in the copy INTEGRATION.md shows -- in step with include/msbwt_hip.h:
same names, same arity, and every parameter / return type the C type's Rust spelling."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

C_TO_RUST = {
    "uint8_t": "u8", "uint64_t": "u64", "size_t": "usize", "int": "c_int", "double": "f64",
    "const uint8_t *": "*const u8", "uint8_t *": "*mut u8", "const uint64_t *": "*const u64", "uint64_t *": "*mut u64",
    "const char *": "*const c_char", "const void *": "*const c_void", "void *": "*mut c_void", "double *": "*mut f64",
    "msbwt_rle *": "*mut MsbwtRle", "const msbwt_rle *": "*const MsbwtRle", "const msbwt_rle *const *": "*const *const MsbwtRle",
    "void **": "*mut *mut c_void",
    "void": "",
}


def c_declarations():
    text = open(os.path.join(ROOT, "include", "msbwt_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    decls = {}
    for m in re.finditer(r"([A-Za-z_][A-Za-z_0-9 \*]*?)\b(msbwt_[a-z_0-9]+)\s*\(([^)]*)\)\s*;", text):
        ret, name, params = m.group(1).strip(), m.group(2), m.group(3)
        types = []
        for p in [x.strip() for x in params.split(",")]:
            if p == "void" or not p:
                continue
            t = re.sub(r"\b[A-Za-z_][A-Za-z_0-9]*$", "", p).strip()  # drop the parameter name
            types.append(re.sub(r"\s*\*\s*", " *", re.sub(r"\s+", " ", t)).replace("* *", "**").replace(" *const *", " *const *").strip())
        decls[name] = (re.sub(r"\s*\*$", " *", ret), types)
    return decls


def rust_declarations(text):
    block = re.search(r'extern "C" \{(.*?)\n\}', text, flags=re.S).group(1)
    block = re.sub(r"//[^\n]*", "", block)
    decls = {}
    for m in re.finditer(r"fn (msbwt_[a-z_0-9]+)\s*\(([^)]*)\)\s*(?:->\s*([^;]+))?;", block, flags=re.S):
        params = [p.split(":", 1)[1].strip() for p in m.group(2).split(",") if ":" in p]
        decls[m.group(1)] = ((m.group(3) or "").strip(), [re.sub(r"\s+", " ", p) for p in params])
    return decls


def norm_c(t):
    t = re.sub(r"\s+", " ", t).strip()
    return re.sub(r" ?\* ?", " *", t).replace("* *", "**").replace(" *const *", " *const *").replace("  ", " ").strip()


SOURCES = {
    "shim/msbwt2-hip/src/lib.rs": lambda: open(os.path.join(ROOT, "shim", "msbwt2-hip", "src", "lib.rs")).read(),
    "INTEGRATION.md": lambda: open(os.path.join(ROOT, "INTEGRATION.md")).read(),
}


@pytest.mark.parametrize("where", sorted(SOURCES))
def test_rust_extern_block_matches_the_header(where):
    c = c_declarations()
    rust = rust_declarations(SOURCES[where]())
    assert len(rust) >= 15, "extern block not found or nearly empty in %s" % where
    for name, (rret, rparams) in rust.items():
        assert name in c, "%s declares %s, which include/msbwt_hip.h does not" % (where, name)
        cret, cparams = c[name]
        assert len(cparams) == len(rparams), "%s: %s takes %d parameters in the header, %d in Rust" % (where, name, len(cparams), len(rparams))
        for i, (ct, rt) in enumerate(zip(cparams, rparams)):
            assert C_TO_RUST[norm_c(ct)] == rt, "%s: %s parameter %d is `%s` in the header, `%s` in Rust" % (where, name, i, ct, rt)
        assert C_TO_RUST[norm_c(cret)] == rret, "%s: %s returns `%s` in the header, `%s` in Rust" % (where, name, cret, rret)


def test_the_two_copies_of_the_shim_declare_the_same_functions():
    a = rust_declarations(SOURCES["shim/msbwt2-hip/src/lib.rs"]())
    b = rust_declarations(SOURCES["INTEGRATION.md"]())
    assert a == b


def test_shim_covers_the_trait_surface():
    """Every method of the reference's `trait BWT` (src/msbwt_core.rs:28-162) has its C entry point bound."""
    rust = rust_declarations(SOURCES["shim/msbwt2-hip/src/lib.rs"]())
    for name in ("msbwt_rle_new", "msbwt_rle_free", "msbwt_rle_load_vector", "msbwt_rle_load_numpy_file", "msbwt_rle_get_symbol_count",
                 "msbwt_rle_get_total_size", "msbwt_rle_constrain_range", "msbwt_rle_count_kmer", "msbwt_rle_count_kmers",
                 "msbwt_rle_constrain_ranges", "msbwt_rle_last_error"):
        assert name in rust
    src = SOURCES["shim/msbwt2-hip/src/lib.rs"]()
    for method in ("fn load_vector(&mut self", "fn load_numpy_file(&mut self", "fn get_symbol_count(&self", "fn get_total_size(&self",
                   "unsafe fn constrain_range(&self", "fn count_kmer(&self"):
        assert method in src
