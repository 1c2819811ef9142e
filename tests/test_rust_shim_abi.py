"""The Rust side of the boundary cannot be compiled in this image (no cargo / rustc), but it can be kept from drifting:
every `extern "C"` declaration in integration/halo2_gpu_shim/src/lib.rs and in the Rust blocks of INTEGRATION.md must
name a function that include/summa_gpu.h or include/summa_prover.h declares, with the same number of parameters and the
same kind of type in every position (pointer, 32 / 64-bit integer, C int, size, pointer-to-pointer) and the same kind of
return value -- the check `test_abi_exports_every_declared_symbol` makes for the ctypes binding, for the Rust one."""
import os
import re

from conftest import ROOT


def _strip_comments(text, rust=False):
    text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)
    return re.sub(r"//[^\n]*", " ", text)


def _c_kind(decl: str) -> str:
    """kind of a C parameter / return type"""
    d = re.sub(r"\b(const|struct|restrict)\b", " ", decl).strip()
    stars = d.count("*") + (1 if "[" in d else 0)
    base = re.sub(r"\[[^\]]*\]", "", d).replace("*", " ").split()
    if not base:
        return "?"
    if base[-1] not in ("void", "int", "float", "char", "size_t") and not base[-1].endswith("_t") and len(base) > 1:
        base = base[:-1]                       # drop the parameter's name
    t = base[0] if len(base) == 1 else " ".join(base[:-1]) if base[-1] not in ("void", "int", "float", "char", "size_t") and not base[-1].endswith("_t") else " ".join(base)
    t = t.split()[-1] if t.split() else t
    if stars >= 2:
        return "ptrptr"
    if stars == 1:
        return "ptr"
    return {"uint32_t": "u32", "uint64_t": "u64", "int": "int", "size_t": "size", "float": "f32", "void": "void",
            "sp_transcript": "int"}.get(t, t)


def c_functions():
    out = {}
    for name in ("summa_gpu.h", "summa_prover.h"):
        text = _strip_comments(open(os.path.join(ROOT, "include", name)).read())
        text = re.sub(r"#[^\n]*", " ", text)
        for m in re.finditer(r"(?:^|[;}\n])\s*((?:const\s+)?[A-Za-z_][\w\s]*?[\s\*]+)(s[gp]_\w+)\s*\(([^()]*)\)\s*;", text):
            ret, fn, params = m.group(1), m.group(2), m.group(3).strip()
            ps = [] if params in ("", "void") else [_c_kind(p) for p in params.split(",")]
            out[fn] = (_c_kind(ret + " x") if "*" in ret else _c_kind(ret), ps)
    return out


def _rust_kind(t: str) -> str:
    t = t.strip()
    if t.count("*") >= 2:
        return "ptrptr"
    if t.startswith("*"):
        return "ptr"
    return {"u32": "u32", "u64": "u64", "c_int": "int", "size_t": "size", "usize": "size", "f32": "f32"}.get(t, t)


def rust_functions(text: str):
    out = {}
    text = _strip_comments(text)
    for block in re.finditer(r'extern\s+"C"\s*\{(.*?)\n\s*\}', text, flags=re.S):
        for m in re.finditer(r"(?:pub\s+)?fn\s+(s[gp]_\w+)\s*\((.*?)\)\s*(?:->\s*([^;]+))?;", block.group(1), flags=re.S):
            fn, params, ret = m.group(1), m.group(2), (m.group(3) or "void").strip()
            if "..." in params:
                continue                          # an elided parameter list in prose: name only
            ps = []
            for p in [q for q in params.split(",") if q.strip()]:
                ps.append(_rust_kind(p.split(":", 1)[1]))
            out[fn] = (_rust_kind(ret), ps)
    return out


def test_c_header_parser_sees_the_whole_abi():
    from circuits_halo2_amd import ffi
    decl = c_functions()
    assert set(ffi.EXPORTS) <= set(decl) and set(ffi.PROVER_EXPORTS) <= set(decl)
    assert decl["sg_msm_g1"] == ("int", ["ptr", "ptr", "size", "ptr"])
    assert decl["sg_msm_g1_batch_dev"] == ("int", ["ptrptr", "ptrptr", "ptr", "size", "ptr", "ptr"])
    assert decl["sg_last_error"] == ("ptr", []) and decl["sg_shutdown"] == ("void", [])
    assert decl["sp_create_proof"][1] == ["u64", "ptrptr", "ptr", "u32", "int", "int", "ptr", "ptr", "size", "ptr"]


def test_rust_declarations_match_the_c_headers():
    decl = c_functions()
    sources = {"integration/halo2_gpu_shim/src/lib.rs": open(os.path.join(ROOT, "integration", "halo2_gpu_shim", "src", "lib.rs")).read()}
    md = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    sources["INTEGRATION.md"] = "\n".join(re.findall(r"```rust(.*?)```", md, flags=re.S))
    seen = 0
    for where, text in sources.items():
        rust = rust_functions(text)
        assert rust, where
        for fn, (ret, params) in rust.items():
            assert fn in decl, f"{where}: {fn} is not declared in include/*.h"
            c_ret, c_params = decl[fn]
            assert len(params) == len(c_params), f"{where}: {fn} takes {len(c_params)} parameters in C, {len(params)} in Rust"
            assert params == c_params, f"{where}: {fn}: C {c_params} vs Rust {params}"
            assert ret == c_ret, f"{where}: {fn} returns {c_ret} in C, {ret} in Rust"
            seen += 1
    lib = rust_functions(sources["integration/halo2_gpu_shim/src/lib.rs"])
    for fn in ("sg_msm_g1", "sg_ntt_fr", "sg_srs_upload", "sg_commit", "sp_key_create", "sp_create_proof", "sp_key_destroy", "sp_last_error",
               "sp_verify_proof"):
        assert fn in lib, fn
    assert seen >= 40
