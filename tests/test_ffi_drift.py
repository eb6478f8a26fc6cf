"""Guards the FFI declarations that cannot be compiled in this image against drift from the header.

include/kmerhip.h is the single source of truth.  Three hand-written mirrors declare (parts of) it again:
  * bindings/rust/src/lib.rs   (`pub mod sys`: #[repr(C)] structs + extern "C" block)   -- no rustc here
  * INTEGRATION.md             (the Rust snippet a krust maintainer would paste)         -- prose
  * krust_amd/native.py        (ctypes structs)                                          -- loads, but a
                                                                                            wrong field type
                                                                                            still "works"
A stale struct there is a memory-safety bug on the caller's side (round 1: the snippet's KhStats had
lost `text_scan_ms`, so `kh_finish(&mut stats)` would have written 8 bytes past it).  This test parses
the header and checks that every mirror declares the same fields, in the same order, with the matching
type, and that every function a mirror declares exists in the header with the same argument list.
CPU only: text parsing, nothing is loaded or called."""
import ctypes as C
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _read(*parts):
    with open(os.path.join(ROOT, *parts)) as f:
        return f.read()


def _strip_c_comments(text):
    return re.sub(r"/\*.*?\*/", "", text, flags=re.S)


# ---- C side ----------------------------------------------------------------------------------------
C2RUST_SCALAR = {"uint32_t": "u32", "int32_t": "i32", "uint64_t": "u64", "uint8_t": "u8", "double": "f64",
                 "int": "c_int", "char": "c_char", "void": "c_void", "kh_ctx": "KhCtx", "kh_config": "KhConfig",
                 "kh_stats": "KhStats", "kh_unique_id": "KhUniqueId", "kh_group": "KhGroup",
                 "kh_merge_info": "KhMergeInfo"}


def c_type_to_rust(ctype):
    """'const uint8_t *' -> '*const u8'; 'kh_ctx **' -> '*mut *mut KhCtx'; 'uint64_t' -> 'u64'."""
    t = ctype.strip()
    stars = t.count("*")
    t = t.replace("*", " ").split()
    # `const T *const *` : pointer to const pointer to const T
    consts = [i for i, w in enumerate(t) if w == "const"]
    base = [w for w in t if w != "const"]
    assert len(base) == 1, ctype
    r = C2RUST_SCALAR[base[0]]
    if stars == 0:
        return r
    inner_const = 0 in consts                      # const before the base type
    outer_const = any(i > 0 for i in consts)       # `*const *`
    if stars == 1:
        return ("*const " if inner_const else "*mut ") + r
    assert stars == 2, ctype
    inner = ("*const " if inner_const else "*mut ") + r
    return ("*const " if outer_const else "*mut ") + inner


def header_structs_and_functions():
    h = _strip_c_comments(_read("include", "kmerhip.h"))
    defines = {m.group(1): int(m.group(2)) for m in re.finditer(r"#define\s+(\w+)\s+(\d+)u?\s*$", h, flags=re.M)}
    structs = {}
    for m in re.finditer(r"typedef struct (\w+) \{(.*?)\} (\w+);", h, flags=re.S):
        fields = []
        for decl in m.group(2).split(";"):
            decl = " ".join(decl.split())
            if not decl:
                continue
            fm = re.match(r"(.+?)\s*(\*?)\s*(\w+)(\[(\w+)\])?$", decl)
            ctype, star, name, _, arr = fm.groups()
            rt = c_type_to_rust(ctype + star)
            if arr:
                rt = f"[{rt}; {defines.get(arr, arr)}]"
            fields.append((name, rt))
        structs[m.group(3)] = fields
    funcs = {}
    for m in re.finditer(r"^\s*([\w ]+?[\w\*])\s*\b(kh_\w+)\s*\(([^;{]*?)\)\s*;", h, flags=re.M | re.S):
        ret, name, args = m.groups()
        ret = " ".join(ret.split())
        arglist = []
        args = " ".join(args.split())
        if args and args != "void":
            for a in args.split(","):
                a = a.strip()
                am = re.match(r"(.+?)(\w+)$", a)  # type then the parameter name
                arglist.append(c_type_to_rust(am.group(1)))
        funcs[name] = (None if ret == "void" else c_type_to_rust(ret), arglist)
    return structs, funcs


# ---- Rust side -------------------------------------------------------------------------------------
def _norm_rust_type(t):
    t = " ".join(t.replace("sys::", "").split())
    return t


def rust_structs_and_functions(src):
    src = re.sub(r"//[^\n]*", "", src)
    structs = {}
    for m in re.finditer(r"#\[repr\(C\)\](?:\s*#\[[^\]]*\])*\s*(?:pub )?struct (\w+)\s*\{(.*?)\}", src, flags=re.S):
        fields = []
        for decl in re.split(r",(?![^\[]*\])", m.group(2)):
            decl = " ".join(decl.split())
            if not decl:
                continue
            fm = re.match(r"(?:pub )?(\w+)\s*:\s*(.+)$", decl)
            fields.append((fm.group(1), _norm_rust_type(fm.group(2))))
        structs[m.group(1)] = fields
    funcs = {}
    for ext in re.finditer(r'extern "C" \{(.*?)\n\s*\}', src, flags=re.S):
        for m in re.finditer(r"(?:pub )?fn (kh_\w+)\s*\((.*?)\)\s*(?:->\s*([^;]+))?;", ext.group(1), flags=re.S):
            name, args, ret = m.groups()
            arglist = []
            for a in re.split(r",(?![^\[]*\])", args):
                a = " ".join(a.split())
                if a:
                    arglist.append(_norm_rust_type(a.split(":", 1)[1]))
            funcs[name] = (None if ret is None else _norm_rust_type(ret), arglist)
    return structs, funcs


RUST_STRUCT_OF = {"kh_config": "KhConfig", "kh_stats": "KhStats", "kh_unique_id": "KhUniqueId",
                  "kh_merge_info": "KhMergeInfo"}


def _check_rust_mirror(label, src, must_have_structs):
    hs, hf = header_structs_and_functions()
    rs, rf = rust_structs_and_functions(src)
    assert hs and hf and rf, f"{label}: parse failed"
    for cname, rname in RUST_STRUCT_OF.items():
        if rname not in rs:
            assert rname not in must_have_structs, f"{label}: struct {rname} is missing"
            continue
        assert cname in hs, f"{label}: {rname} has no counterpart in kmerhip.h"
        assert rs[rname] == hs[cname], (f"{label}: {rname} differs from {cname} in include/kmerhip.h\n"
                                        f"  header: {hs[cname]}\n  mirror: {rs[rname]}")
    for name, (ret, args) in rf.items():
        assert name in hf, f"{label}: {name} is not declared in kmerhip.h"
        assert (ret, args) == hf[name], (f"{label}: {name} differs from kmerhip.h\n  header: {hf[name]}\n"
                                         f"  mirror: {(ret, args)}")
    return rs, rf


def test_header_parser_sees_the_whole_abi():
    """The parser itself: every kh_* the header declares comes out with a signature."""
    hs, hf = header_structs_and_functions()
    text = _strip_c_comments(_read("include", "kmerhip.h"))
    declared = sorted(set(re.findall(r"\b(kh_[a-z0-9_]+)\s*\(", text)))
    assert sorted(hf) == declared
    assert [n for n, _ in hs["kh_config"]][:3] == ["struct_size", "k", "min_quality"]
    assert hs["kh_stats"][-1] == ("text_scan_ms", "f64") or hs["kh_stats"][-1][0] != "stage_ms"
    assert hf["kh_create"] == ("c_int", ["*mut *mut KhCtx", "*const KhConfig"])
    assert hf["kh_merge_regions_device"][1][3] == "*const *const u64"


def test_rust_crate_sys_block_matches_header():
    _check_rust_mirror("bindings/rust/src/lib.rs", _read("bindings", "rust", "src", "lib.rs"), {"KhConfig", "KhStats"})


def test_integration_md_rust_snippets_match_header():
    md = _read("INTEGRATION.md")
    blocks = re.findall(r"```rust\n(.*?)```", md, flags=re.S)
    assert blocks
    _check_rust_mirror("INTEGRATION.md", "\n".join(blocks), {"KhConfig", "KhStats"})


def test_ctypes_structs_match_header():
    """native.py's ctypes structs: same field names, order and widths as the header's structs."""
    from krust_amd import native
    hs, _ = header_structs_and_functions()
    width = {"u32": 4, "i32": 4, "u64": 8, "f64": 8, "c_int": 4, "u8": 1, "c_char": 1}

    def rust_size(t):
        m = re.match(r"\[(\w+); (\d+)\]", t)
        if m:
            return width[m.group(1)] * int(m.group(2))
        return C.sizeof(C.c_void_p) if t.startswith("*") else width[t]

    for cname, cls in (("kh_config", native.KhConfig), ("kh_stats", native.KhStats)):
        got = [(n, C.sizeof(t)) for n, t in cls._fields_]
        want = [(n, rust_size(t)) for n, t in hs[cname]]
        assert got == want, f"native.{cls.__name__} differs from {cname}"
    for cname, pyname in (("kh_unique_id", "KhUniqueId"), ("kh_merge_info", "KhMergeInfo")):
        if cname in hs and hasattr(native, pyname):
            cls = getattr(native, pyname)
            assert [(n, C.sizeof(t)) for n, t in cls._fields_] == [(n, rust_size(t)) for n, t in hs[cname]]
