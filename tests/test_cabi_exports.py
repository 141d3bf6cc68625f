"""CPU: the C-ABI library loads and exports every symbol include/fvhip.h declares
(no compute calls without a GPU), and fails loudly without a device."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    txt = open(os.path.join(ROOT, "include", "fvhip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(fv_[a-z0-9_]+)\s*\(", txt)))


def test_every_declared_symbol_is_exported_and_bound(fv):
    import ctypes

    from fvamd import _lib

    lib = ctypes.CDLL(_lib.LIBPATH)
    declared = _declared_symbols()
    assert len(declared) >= 35
    for name in declared:
        assert hasattr(lib, name), "libfvhip.so does not export %s" % name
        assert name in _lib.SIGNATURES, "python binding lacks %s" % name
    assert set(_lib.SIGNATURES) == set(declared)
    assert fv.load().fv_abi_version() == _lib.ABI_VERSION == 4
    hdr = open(os.path.join(ROOT, "include", "fvhip.h")).read()
    assert "#define FVHIP_ABI_VERSION 4" in hdr
    # the experimenter's panel is exported for the tools, but it is not part of the public header
    assert "fv_tune" not in declared and hasattr(lib, "fv_tune") and set(_lib.PRIVATE_SIGNATURES) == {"fv_tune", "fv_comm_init_local"}
    assert "fv_comm_init_local" not in declared  # (round 5: the loopback transport of the rehearsals left the public header)


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "finitevolume.jl_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp", ".jl")):
                src = open(os.path.join(dirpath, f)).read()
                assert "fv_oracle" not in src and "libfvoracle" not in src, f


def test_no_device_fails_loudly(fv):
    import ctypes

    n = ctypes.c_int(0)
    try:
        hip = ctypes.CDLL("libamdhip64.so")
        have = hip.hipGetDeviceCount(ctypes.byref(n)) == 0 and n.value > 0
    except OSError:
        have = False
    if have:
        pytest.skip("a GPU is visible here")
    with pytest.raises(fv.FVError, match="no CPU fallback"):
        fv.Context(0)


def test_julia_shim_calls_only_declared_symbols_with_matching_arity():
    """The Julia binding was written without a Julia runtime: at least every `ccall((:name, libfvhip), ...)` must name a
    symbol of include/fvhip.h, with as many argument types as the C declaration has parameters."""
    hdr = open(os.path.join(ROOT, "include", "fvhip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    arity = {}
    for name, params in re.findall(r"\b(fv_[a-z0-9_]+)\s*\(([^;{]*?)\)\s*;", hdr, flags=re.S):
        params = params.strip()
        arity[name] = 0 if params in ("", "void") else params.count(",") + 1
    src = open(os.path.join(ROOT, "finitevolume.jl_amd", "julia", "FiniteVolumeHIP.jl")).read()
    calls = re.findall(r"ccall\(\(:(\w+),\s*libfvhip\),\s*\w+(?:\{[^}]*\})?,\s*\(([^)]*)\)", src)
    assert len(calls) >= 20
    for name, types in calls:
        assert name in arity, "the shim calls %s, which include/fvhip.h does not declare" % name
        ntypes = len([t for t in re.split(r",(?![^{]*\})", types) if t.strip()])
        assert ntypes == arity[name], "%s: %d argument types in the shim, %d parameters in the header" % (name, ntypes, arity[name])


def _julia_methods(src):
    """name -> list of (min, max) positional counts of every `function name(...)` / `name(...) = ...` definition at top
    level of the shim (keywords after `;` and `where` clauses dropped, defaults counted as optional)."""
    out = {}
    for m in re.finditer(r"^(?:function\s+)?([A-Za-z_][A-Za-z0-9_!]*)\(", src, flags=re.M):
        name, i = m.group(1), m.end()
        if not (m.group(0).startswith("function") or re.match(r"[^\n]*\)\s*(?:where\s*\{[^}]*\}\s*)?=[^=]", src[m.start():])):
            continue
        depth, j, args, cur = 1, i, [], ""
        while depth and j < len(src):
            c = src[j]
            if c in "([{":
                depth += 1
            elif c in ")]}":
                depth -= 1
                if depth == 0:
                    break
            if depth == 1 and c == ",":
                args.append(cur)
                cur = ""
            elif depth == 1 and c == ";":
                args.append(cur)
                cur = None
                break
            else:
                cur += c
            j += 1
        if cur is not None and cur.strip():
            args.append(cur)
        args = [a.strip() for a in args if a.strip()]
        if any(a.endswith("...") for a in args):
            continue  # a forwarding helper, not a method of the call surface
        nopt = sum(1 for a in args if re.search(r"[^=!<>]=[^=]", a))
        out.setdefault(name, []).append((len(args) - nopt, len(args)))
    return out


def test_julia_shim_defines_the_reference_call_surface_with_its_positional_arities():
    """tests/golden/reference_api.json lists every function of SURVEY.md 8b with the positional arities of its methods in
    the reference's sources; the shim (written without a Julia runtime) must define each of them so that every such call
    finds a method — `FiniteVolume.f(args...)` in the package's examples and tests then resolves after
    `const FiniteVolume = FiniteVolumeHIP`."""
    import json

    api = json.load(open(os.path.join(ROOT, "tests", "golden", "reference_api.json")))
    src = open(os.path.join(ROOT, "finitevolume.jl_amd", "julia", "FiniteVolumeHIP.jl")).read()
    have = _julia_methods(src)
    for name, spec in api.items():
        if name.startswith("_"):
            continue
        assert name in have, "the shim does not define %s (%s)" % (name, spec["where"])
        for lo, hi in spec["arity"]:
            for k in range(lo, hi + 1):
                assert any(a <= k <= b for a, b in have[name]), "%s: no method takes %d positional arguments (%s); the shim has %s" % (name, k, spec["where"], have[name])
    # the positional ORDER of the long signatures: spot checks on the names the reference uses
    for name, lead in (("integratedfdplambda", ["u2", "p", "lambdas", "ts_lambda", "tspan", "Ss", "volumes", "neighbors"]),
                       ("getadjointfunctions", ["sigma", "obsfreenodes", "uobs", "u0", "tspan", "Ss", "volumes", "neighbors"]),
                       ("adaptivebackwardeulerstep!", ["rhs", "A", "getb", "u_k", "t", "dt", "linearsolver", "atol", "callback"])):
        m = re.search(r"function\s+" + re.escape(name) + r"\(([^)]*)\)", src)
        got = [re.split(r"[:=]", a.strip())[0] for a in m.group(1).split(";")[0].split(",")][: len(lead)]
        assert got == lead, (name, got)
    for kw in ("stepper!", "linearsolver", "atol", "callback", "dt0", "keep"):
        assert re.search(r"[;,]\s*[^)]*\b" + re.escape(kw) + r"\s*=", src), "keyword %s missing" % kw


def test_julia_package_wrapper_has_the_reference_name_and_uuid():
    """`import FiniteVolume` in the reference's tests and examples must resolve to the shim without an edited line: a package
    directory with the reference's name and UUID (tests/golden/reference_api.json records them) whose module includes the shim
    and binds its names."""
    import json

    api = json.load(open(os.path.join(ROOT, "tests", "golden", "reference_api.json")))
    pkg = os.path.join(ROOT, "finitevolume.jl_amd", "julia", "FiniteVolume")
    toml = open(os.path.join(pkg, "Project.toml")).read()
    assert 'name = "FiniteVolume"' in toml and ('uuid = "%s"' % api["_package"]["uuid"]) in toml
    src = open(os.path.join(pkg, "src", "FiniteVolume.jl")).read()
    assert re.search(r"^module FiniteVolume\s*$", src, flags=re.M) and "FiniteVolumeHIP.jl" in src and "@eval const $name = FiniteVolumeHIP.$name" in src
    assert os.path.exists(os.path.normpath(os.path.join(pkg, "src", "..", "..", "FiniteVolumeHIP.jl")))
    shim = open(os.path.join(ROOT, "finitevolume.jl_amd", "julia", "FiniteVolumeHIP.jl")).read()
    assert "FVHIP_ABI_VERSION = 4" in shim and ":fv_abi_version" in shim


# ------------------------------------------------------------------ round 5: static checks beyond arity
def _split_top(s):
    """split at commas that are not inside (), [] or {}"""
    out, cur, depth = [], "", 0
    for c in s:
        if c in "([{":
            depth += 1
        elif c in ")]}":
            depth -= 1
        if c == "," and depth == 0:
            out.append(cur)
            cur = ""
        else:
            cur += c
    if cur.strip():
        out.append(cur)
    return [t.strip() for t in out if t.strip()]


def _c_class(decl):
    """A C parameter / return type of include/fvhip.h -> the class a Julia ccall type must belong to."""
    d = re.sub(r"\b(const|struct)\b", " ", decl)
    d = re.sub(r"\b[A-Za-z_][A-Za-z0-9_]*\s*(\[[^\]]*\])?\s*$", lambda m: "*" if m.group(1) else "", d.strip()) if not re.fullmatch(r"\s*(void|int|double|int32_t|int64_t|char)\s*\**\s*", d) else d
    d = d.replace(" ", "")
    if "*" in d:
        return "ptr"
    return {"int": "i32", "int32_t": "i32", "int64_t": "i64", "double": "f64", "void": "void", "uint32_t": "i32"}.get(d, "?" + d)


def _julia_class(t):
    t = t.strip()
    if re.match(r"^(Ptr|Ref)\{", t) or t in ("Cstring", "Ptr{Cvoid}"):
        return "ptr"
    return {"Cint": "i32", "Int32": "i32", "Cuint": "i32", "UInt32": "i32", "Int64": "i64", "Clonglong": "i64", "Float64": "f64", "Cdouble": "f64", "Cvoid": "void"}.get(t, "?" + t)


def test_julia_shim_ccall_types_match_the_header():
    """Every ccall of the shim against the C declaration it binds: return type and every argument type belong to the class of the
    header's C type (Int64 <-> int64_t, Cint / Int32 <-> int / int32_t, Float64 <-> double, Ptr / Ref / Cstring <-> pointers and
    arrays).  The shim has never met a Julia parser; a width mismatch here would corrupt the call frame silently."""
    hdr = open(os.path.join(ROOT, "include", "fvhip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    decls = {}
    for ret, name, params in re.findall(r"^\s*((?:const\s+)?[A-Za-z_][A-Za-z0-9_]*\s*\**)\s*\b(fv_[a-z0-9_]+)\s*\(([^;{]*?)\)\s*;", hdr, flags=re.S | re.M):
        params = params.strip()
        decls[name] = (_c_class(ret + " x") if "*" in ret else _c_class(ret), [] if params in ("", "void") else [_c_class(p) for p in _split_top(params)])
    src = open(os.path.join(ROOT, "finitevolume.jl_amd", "julia", "FiniteVolumeHIP.jl")).read()
    calls = re.findall(r"ccall\(\(:(\w+),\s*libfvhip\),\s*(\w+(?:\{[^}]*\})?),\s*\(([^)]*)\)", src)
    assert len(calls) >= 40
    bad = []
    for name, ret, types in calls:
        cret, cparams = decls[name]
        jparams = [_julia_class(t) for t in _split_top(types)]
        if _julia_class(ret) != cret:
            bad.append((name, "return", ret, cret))
        for k, (jt, ct) in enumerate(zip(jparams, cparams)):
            if jt != ct:
                bad.append((name, k, _split_top(types)[k], ct))
    assert not bad, bad
    assert all(not c.startswith("?") for _, (r, ps) in decls.items() for c in [r] + ps), [(n, d) for n, d in decls.items() if any(c.startswith("?") for c in [d[0]] + d[1])]


def test_julia_solveinfo_mirrors_fv_solve_info_field_by_field():
    hdr = open(os.path.join(ROOT, "include", "fvhip.h")).read()
    body = re.search(r"typedef struct fv_solve_info \{(.*?)\} fv_solve_info;", hdr, flags=re.S).group(1)
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    cfields = [(m.group(2), {"int32_t": "Int32", "int64_t": "Int64", "double": "Float64"}[m.group(1)]) for m in re.finditer(r"(int32_t|int64_t|double)\s+(\w+)\s*;", body)]
    src = open(os.path.join(ROOT, "finitevolume.jl_amd", "julia", "FiniteVolumeHIP.jl")).read()
    jbody = re.search(r"^struct SolveInfo[^\n]*\n(.*?)^end", src, flags=re.S | re.M).group(1)
    jfields = [(m.group(1), m.group(2)) for m in re.finditer(r"^\s*(\w+)::(\w+)", jbody, flags=re.M)]
    assert len(cfields) == 6 and jfields == cfields, (jfields, cfields)


def _julia_lex(src):
    """A small lexer for the subset of Julia the shim uses: strips comments (#, #= =#), strings ("...", \"\"\"...\"\"\", with $(...)
    interpolation skipped by bracket depth), character literals; yields (kind, text, line) for identifiers / keywords and brackets.
    Raises on an unterminated string or block comment."""
    toks, i, n, line = [], 0, len(src), 1
    while i < n:
        c = src[i]
        if c == "\n":
            line += 1
            toks.append(("nl", "\n", line))
            i += 1
        elif src.startswith("#=", i):
            j = src.find("=#", i + 2)
            assert j >= 0, "unterminated block comment at line %d" % line
            line += src.count("\n", i, j)
            i = j + 2
        elif c == "#":
            j = src.find("\n", i)
            i = n if j < 0 else j
        elif src.startswith('"""', i) or c == '"':
            q = '"""' if src.startswith('"""', i) else '"'
            j = i + len(q)
            while True:
                assert j < n, "unterminated string starting at line %d" % line
                if src[j] == "\\":
                    j += 2
                    continue
                if src.startswith("$(", j):  # interpolation: skip to the matching parenthesis
                    depth, j = 1, j + 2
                    while depth:
                        assert j < n, "unterminated interpolation at line %d" % line
                        depth += src[j] == "("
                        depth -= src[j] == ")"
                        j += 1
                    continue
                if src.startswith(q, j):
                    break
                j += 1
            line += src.count("\n", i, j)
            toks.append(("str", src[i:j + len(q)], line))
            i = j + len(q)
        elif c == "'" and re.match(r"'(\\.|[^\\'])'", src[i:i + 4]):
            i += len(re.match(r"'(\\.|[^\\'])'", src[i:i + 4]).group(0))
        elif c in "()[]{}":
            toks.append(("br", c, line))
            i += 1
        elif re.match(r"[A-Za-z_@]", c):
            m = re.match(r"@?[A-Za-z_][A-Za-z0-9_!]*", src[i:])
            if m is None:  # (a lone @, e.g. `@.`)
                i += 1
                continue
            toks.append(("id", m.group(0), line))
            i += m.end()
        elif c == ":" and i + 1 < n and re.match(r"[A-Za-z_]", src[i + 1]) and (i == 0 or not re.match(r"[A-Za-z0-9_\)\]]", src[i - 1])):
            m = re.match(r":[A-Za-z_][A-Za-z0-9_!]*", src[i:])  # a symbol (:end, :resnorm): not a keyword
            i += m.end()
        else:
            i += 1
    return toks


def test_julia_shim_blocks_brackets_and_strings_balance():
    """No Julia runtime has ever parsed the shim.  A lexer pass over it and over the package wrapper: every string and block comment
    terminates, every bracket closes in order, and the block openers (module, function, struct, if, for, while, let, do, begin, try,
    quote, macro; `end` inside [] is an index, `for` / `if` inside brackets are comprehension parts) are matched by exactly as many `end`s."""
    for rel in (("finitevolume.jl_amd", "julia", "FiniteVolumeHIP.jl"), ("finitevolume.jl_amd", "julia", "FiniteVolume", "src", "FiniteVolume.jl")):
        src = open(os.path.join(ROOT, *rel)).read()
        toks = _julia_lex(src)
        stack, blocks = [], []
        openers = {"module", "baremodule", "function", "struct", "if", "for", "while", "let", "do", "begin", "try", "quote", "macro"}
        prev = None
        for kind, text, line in toks:
            if kind == "br":
                if text in "([{":
                    stack.append((text, line))
                else:
                    assert stack, "%s: closing %s at line %d without an opener" % (rel[-1], text, line)
                    o, ol = stack.pop()
                    assert {"(": ")", "[": "]", "{": "}"}[o] == text, "%s: %s (line %d) closed by %s (line %d)" % (rel[-1], o, ol, text, line)
            elif kind == "id":
                inside = [b for b, _ in stack]
                if text == "mutable" or (text == "struct" and prev == "mutable"):
                    if text == "struct":
                        blocks.append((text, line))
                elif text in openers:
                    if text in ("for", "if") and inside and inside[-1] in "([":  # a comprehension / generator part
                        pass
                    elif not (text == "if" and prev == "else"):  # `elseif` is one word in Julia; `else if` would open a block and is not used
                        blocks.append((text, line))
                elif text == "end":
                    if inside and inside[-1] == "[":
                        pass  # a[end]
                    else:
                        assert blocks, "%s: `end` at line %d closes nothing" % (rel[-1], line)
                        blocks.pop()
            if kind != "nl":
                prev = text
        assert not stack, "%s: unclosed %s" % (rel[-1], stack[-3:])
        assert not blocks, "%s: blocks never closed: %s" % (rel[-1], blocks[-5:])


def test_legacy_tune_translates_the_historical_key_numbers_into_the_panel_s_bits(fv):
    """csrc/fv_tune.h has 18 keys since round 5: the members of a family are bits of one key (41: the fused family, 35: storage codes / z-form /
    zero row sum, 31: the re-numbering's policy and where it is computed).  Tests and tools still name the members by the numbers they had as keys
    of their own; `_lib.legacy_tune` keeps the masks and translates.  Checked against a recording stand-in for the library's entry point."""
    from fvamd import _lib

    calls = []

    def raw(k, v):
        calls.append((k, v))
        return 0 if (k, v) != (41, -1) else 1

    t = _lib.legacy_tune(raw)
    assert t(46, 0) == 0 and calls[-1] == (41, 127 & ~2)
    assert t(63, 0) == 0 and calls[-1] == (41, 127 & ~2 & ~64)
    assert t(46, 1) == 0 and calls[-1] == (41, 127 & ~64)
    assert t(41, 0) == 0 and calls[-1] == (41, 127 & ~64 & ~1)
    assert t(49, 0) == 0 and t(50, 0) == 0 and t(55, 0) == 0 and t(59, 0) == 0 and calls[-1] == (41, 2)  # (bit 2, the fused-pass loop, was switched on again above)
    assert t(36, 0) == 0 and calls[-1] == (35, 5) and t(37, 0) == 0 and calls[-1] == (35, 1) and t(35, 0) == 0 and calls[-1] == (35, 0)
    assert t(47, 0) == 0 and calls[-1] == (31, 11) and t(31, 2) == 0 and calls[-1] == (31, 12) and t(47, 1) == 0 and calls[-1] == (31, 2)
    assert t(13, 3) == 0 and calls[-1] == (13, 3)  # every other key goes through unchanged
    assert t(46, 2) == _lib.FV_ERR_ARG and t(31, 5) == _lib.FV_ERR_ARG  # members take 0 / 1, the policy 0 .. 2
    # the panel's documentation names exactly the keys the library accepts
    hdr = open(os.path.join(ROOT, "finitevolume.jl_amd", "csrc", "fv_tune.h")).read()
    documented = sorted(int(k) for k in re.findall(r"^ \* {2,3}(\d+): ", hdr, flags=re.M))
    src = open(os.path.join(ROOT, "finitevolume.jl_amd", "csrc", "fv_spmv.hip")).read()
    body = src[src.index('extern "C" int fv_tune(int key, int value)'):src.index("// lanes per row from the mean row length")]
    accepted = sorted(set(int(k) for k in re.findall(r"key == (\d+)", body)))
    assert documented == accepted and len(accepted) == 18, (documented, accepted)
