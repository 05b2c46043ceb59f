"""The Julia binding (flight.jl_amd/julia/FlightBatch.jl) cannot be executed here (no Julia toolchain), so it is checked by
inspection: every capitalised identifier the module uses unqualified must be imported, defined in the module, or a name of
Julia's Base / Core — the class of error (`UndefVarError` at `using FlightBatch`) a missing import produces. The import
paths themselves are checked against the reference's module tree where /root/reference is present (build container only)."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SHIM = os.path.join(ROOT, "flight.jl_amd", "julia", "FlightBatch.jl")

# names every Julia session has (Base / Core exports the shim uses)
BASE = {"Cint", "Cdouble", "Cstring", "Cvoid", "Ptr", "Ref", "Int", "Int32", "Int64", "UInt8", "UInt32", "Float32", "Float64", "Bool",
        "Integer", "Real", "Symbol", "Vector", "Matrix", "Array", "Dict", "NTuple", "Any", "ENV", "C_NULL", "Base", "Union", "Nothing", "DimensionMismatch"}


def _code_only(src):
    """the source with string literals (single- and triple-quoted, possibly multi-line) emptied and comments removed"""
    out, i, n = [], 0, len(src)
    while i < n:
        if src.startswith('"""', i):
            j = src.find('"""', i + 3)
            out.append('""'); i = j + 3
        elif src[i] == '"':
            j = i + 1
            while src[j] != '"':
                j += 2 if src[j] == "\\" else 1
            out.append('""'); i = j + 1
        elif src[i] == "#":
            j = src.find("\n", i)
            i = n if j < 0 else j
        else:
            out.append(src[i]); i += 1
    return "".join(out)


def _imports(code):
    names, paths = set(), []
    for m in re.finditer(r"^\s*(?:using|import)\s+([\w.]+)\s*:\s*(.+)$", code, flags=re.M):
        mod, items = m.group(1), [s.strip() for s in m.group(2).split(",")]
        names.update(i for i in items if i)
        paths.append((mod, items))
    return names, paths


def test_every_capitalised_identifier_is_in_scope():
    code = _code_only(open(SHIM, encoding="utf-8").read())
    imported, _ = _imports(code)
    defined = set(re.findall(r"^\s*(?:mutable\s+)?struct\s+(\w+)", code, flags=re.M))
    defined |= set(re.findall(r"^\s*module\s+(\w+)", code, flags=re.M))
    for m in re.finditer(r"^\s*const\s+(.+?)\s*=", code, flags=re.M):        # const A, B, C = ...
        defined.update(s.strip() for s in m.group(1).split(","))
    for m in re.finditer(r"\(;\s*([^)]*)\)\s*=", code):                      # (; a, b) = x  destructuring
        defined.update(t.strip() for t in m.group(1).split(","))
    body = re.sub(r"^\s*(?:using|import)\s.*$", "", code, flags=re.M)
    body = re.sub(r"\b\w+::", "", body)                                       # field / argument declarations `name::Type`
    # unqualified uses: an identifier starting with an upper-case ASCII letter that is not a field / submodule access (`x.Name`),
    # not a keyword argument name (`Name = ...` inside a call is lower-case in this file) and not a type parameter
    used = set(m.group(1) for m in re.finditer(r"(?<![\w.!:])([A-Z][A-Za-z0-9_]*)\b", body))
    type_params = set(re.findall(r"where\s*\{?\s*([A-Z]\w*)", body))
    unknown = sorted(n for n in used - imported - defined - BASE - type_params)
    assert not unknown, f"FlightBatch.jl uses names that are neither imported nor defined: {unknown}"
    # the three failures this test was written for
    for name in ("WA", "ECEF", "NED", "Cessna172Sv0", "Cessna172Xv2"):
        assert name in imported, name
    assert not re.search(r"^\s*put!\(", code, flags=re.M), "the shim must not shadow Base.put!"
    assert "pathof(Geodesy" not in code   # pathof() of a submodule is `nothing`


def test_import_paths_exist_in_the_reference_module_tree():
    ref = "/root/reference"
    if not os.path.isdir(ref):
        import pytest
        pytest.skip("the reference tree is only present in the build container")
    # module name -> file that declares it, from `include(...)` + `module X` of the reference sources
    decl = {}
    for base, _, files in os.walk(ref):
        for f in files:
            if f.endswith(".jl"):
                p = os.path.join(base, f)
                txt = open(p, encoding="utf-8", errors="ignore").read()
                for m in re.finditer(r"^module\s+(\w+)", txt, flags=re.M):
                    decl.setdefault(m.group(1), []).append((p, txt))
    code = _code_only(open(SHIM, encoding="utf-8").read())
    _, paths = _imports(code)
    assert paths
    for mod, items in paths:
        parts = mod.split(".")
        assert parts[0] == "Flight", mod
        for part in parts:
            assert part in decl, f"{mod}: no `module {part}` in the reference"
        leaf = parts[-1]
        txts = [t for _, t in decl[leaf]]
        for item in items:
            ok = any(re.search(rf"(?<![\w!]){re.escape(item)}(?![\w!])", t) for t in txts)
            assert ok, f"{item} does not appear in module {leaf}"
            if leaf not in ("Flight", "FlightApps", "FlightPhysics", "Modeling"):
                # names taken with `using M: name` from a leaf module: they must be exported or defined there
                assert any(re.search(rf"^export\b.*(?<![\w!]){re.escape(item)}(?![\w!])", t, flags=re.M) or
                           re.search(rf"^(?:function|const|struct|@kwdef struct)\s+{re.escape(item)}\b", t, flags=re.M) for t in txts), (mod, item)
    # the EGM96 file is looked up next to FlightPhysics' entry file, where the reference keeps it (geodesy.jl:166)
    assert os.path.isfile(os.path.join(ref, "lib", "FlightPhysics", "src", "data", "ww15mgh_le.bin"))
    assert os.path.isfile(os.path.join(ref, "lib", "FlightPhysics", "src", "FlightPhysics.jl"))
