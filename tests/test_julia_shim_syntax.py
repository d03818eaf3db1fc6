"""Julia is absent from the build image, so the shim (`julia/KernelDensityEstimateHIP.jl`) and the cross-check script
(`oracle/julia_crosscheck.jl`) cannot be executed here.  What can be checked without Julia: block structure
(every `function/if/for/while/begin/let/struct/module/try/do/quote` closed by an `end`, brackets balanced, strings
terminated), and that every `ccall` in the shim names a symbol that `include/kdehip.h` declares with the same number
of arguments."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SHIM = os.path.join(ROOT, "kerneldensityestimate.jl_amd", "julia", "KernelDensityEstimateHIP.jl")
CROSS = os.path.join(ROOT, "oracle", "julia_crosscheck.jl")
HEADER = os.path.join(ROOT, "include", "kdehip.h")

OPENERS = {"function", "if", "for", "while", "begin", "let", "struct", "module", "try", "do", "quote", "macro"}


def strip_code(text):
    """Julia source with comments removed and string / char literals blanked (docstrings included)."""
    out, i, n = [], 0, len(text)
    while i < n:
        c = text[i]
        if text.startswith('"""', i):
            j = text.index('"""', i + 3)
            out.append('""' + "\n" * text.count("\n", i, j))
            i = j + 3
        elif c == '"':
            j = i + 1
            while text[j] != '"':
                if text[j] == "\\":
                    j += 1
                assert text[j] != "\n", f"unterminated string near offset {i}"
                j += 1
            out.append('""')
            i = j + 1
        elif c == "#":
            j = text.find("\n", i)
            i = n if j < 0 else j
        else:
            out.append(c)
            i += 1
    return "".join(out)


def check_blocks(path):
    code = strip_code(open(path).read())
    depth, stack = 0, []
    brackets = {"(": ")", "[": "]", "{": "}"}
    bstack = []
    for lineno, line in enumerate(code.split("\n"), 1):
        for tok in re.finditer(r"[A-Za-z_@!][A-Za-z_0-9!]*|[()\[\]{}]", line):
            t = tok.group(0)
            if t in brackets:
                bstack.append((brackets[t], lineno))
            elif t in brackets.values():
                assert bstack and bstack[-1][0] == t, f"{path}:{lineno}: unbalanced '{t}'"
                bstack.pop()
            elif t in OPENERS and not bstack:  # (a keyword inside brackets is a generator / comprehension `for`, `if`)
                stack.append((t, lineno))
            elif t == "end" and not bstack:
                assert stack, f"{path}:{lineno}: 'end' without an opener"
                stack.pop()
    assert not bstack, f"{path}: bracket opened at line {bstack[-1][1]} never closed"
    assert not stack, f"{path}: '{stack[-1][0]}' at line {stack[-1][1]} never closed"
    return depth


def header_params():
    """symbol -> list of C parameter types (normalised: no names, no const, single spaces)"""
    text = re.sub(r"/\*.*?\*/", "", open(HEADER).read(), flags=re.S)
    text = re.sub(r"//[^\n]*", "", text)
    decls = {}
    for m in re.finditer(r"\b(kdehip_[a-z0-9_]+)\s*\(([^;{]*?)\)\s*;", text, flags=re.S):
        args = m.group(2).strip()
        params = []
        if args not in ("", "void"):
            for a in args.split(","):
                a = re.sub(r"\bconst\b", "", a).strip()
                mm = re.match(r"(.*?)([A-Za-z_][A-Za-z_0-9]*)?$", a)          # drop the parameter name
                typ = mm.group(1).strip() if mm.group(1).strip() else a
                params.append(re.sub(r"\s+", " ", typ).replace(" *", "*"))
        decls[m.group(1)] = params
    return decls


def header_arity():
    return {k: len(v) for k, v in header_params().items()}


# what a Julia ccall argument type may be bound to
JULIA_TO_C = {
    "Cint": {"int"}, "Int64": {"int64_t"}, "UInt64": {"uint64_t"},
    "Ptr{Float64}": {"double*"}, "Ptr{Int64}": {"int64_t*"}, "Ptr{Int32}": {"int32_t*"}, "Ref{Int32}": {"int32_t*"},
    "Ptr{UInt8}": {"uint8_t*"}, "Ptr{CDensity}": {"kdehip_density*"}, "Ref{CDensity}": {"kdehip_density*"},
    "Cstring": {"char*"}, "Cvoid": {"void"},
    "Ptr{Cvoid}": {"kdehip_device_density*"}, "Ref{Ptr{Cvoid}}": {"kdehip_device_density**"},
    "Ptr{Ptr{Cvoid}}": {"kdehip_device_density**"},
    "Ptr{CMulItem}": {"kdehip_mul_item*"},
}


def test_julia_files_are_block_balanced():
    check_blocks(SHIM)
    check_blocks(CROSS)


def test_every_ccall_matches_the_header():
    decls = header_arity()
    code = strip_code(open(SHIM).read())
    calls = list(re.finditer(r"ccall\(\(:(kdehip_[a-z0-9_]+),\s*libkdehip\),\s*([A-Za-z0-9_{}]+),\s*\(", code))
    assert len(calls) >= 8
    for m in calls:
        name = m.group(1)
        assert name in decls, f"{name} is not declared in include/kdehip.h"
        # the argument-type tuple: from the '(' that ends the match to its closing ')'
        i, depth = m.end(), 1
        while depth:
            depth += {"(": 1, ")": -1}.get(code[i], 0)
            i += 1
        types = code[m.end():i - 1].strip()
        items, d, cur = [], 0, ""
        for ch in types:
            if ch in "({":
                d += 1
            if ch in ")}":
                d -= 1
            if ch == "," and d == 0:
                items.append(cur)
                cur = ""
            else:
                cur += ch
        if cur.strip():
            items.append(cur)
        assert len(items) == decls[name], f"{name}: ccall passes {len(items)} argument types, header declares {decls[name]}"
        cparams = header_params()[name]
        for pos, (jt, ct) in enumerate(zip((x.strip() for x in items), cparams)):
            assert jt in JULIA_TO_C, f"{name}: argument {pos}: unknown Julia type {jt}"
            assert ct in JULIA_TO_C[jt], f"{name}: argument {pos}: Julia passes {jt}, header declares {ct}"


# the keyword list of the reference's prodAppxMSGibbsS (src/MSGibbs01.jl:649-663), in its order
REFERENCE_PROD_KEYWORDS = ["Niter", "addop", "diffop", "getMu", "getLambda", "glbs", "addEntropy", "ndims", "Ndens", "Np",
                           "maxNp", "Nlevels", "randU", "randN", "partialDimMask"]


def _keyword_names(signature):
    """keyword names of a Julia `function f(a, b; k1=..., k2::T=...)` signature text (after the ';')"""
    kw = signature.split(";", 1)[1]
    names, depth, cur = [], 0, ""
    for ch in kw:
        if ch in "([{":
            depth += 1
        if ch in ")]}":
            depth -= 1
        if ch == "," and depth == 0:
            names.append(cur)
            cur = ""
        else:
            cur += ch
    names.append(cur)
    out = []
    for n in names:
        m = re.match(r"\s*([A-Za-z_][A-Za-z_0-9]*)", n)
        if m:
            out.append(m.group(1))
    return out


def _signatures(code, name):
    """every `function name(...)` signature in `code` (text between the parentheses)"""
    sigs = []
    for m in re.finditer(r"function\s+" + re.escape(name) + r"\(", code):
        i, depth = m.end(), 1
        while depth:
            depth += {"(": 1, ")": -1}.get(code[i], 0)
            i += 1
        sigs.append(code[m.end():i - 1])
    return sigs


def test_prodAppxMSGibbsS_keyword_surface_is_the_references():
    """The shim's front end and the method enable!() installs into KernelDensityEstimate take the reference's keyword
    list (src/MSGibbs01.jl:645-664) -- `maxNp` and `Nlevels` included -- and the deprecated positional-Niter method
    (:632-643) exists."""
    code = strip_code(open(SHIM).read())
    sigs = _signatures(code, "prodAppxMSGibbsS")
    kw_sigs = [s for s in sigs if ";" in s]
    pos_sigs = [s for s in sigs if ";" not in s]
    kw_sigs = [s for s in kw_sigs if "DeviceDensity" not in s]   # (the device-resident method has its own, shorter list)
    assert len(kw_sigs) == 2, "module-level front end + the override installed by enable!()"
    for s in kw_sigs:
        names = _keyword_names(s)
        assert names[:len(REFERENCE_PROD_KEYWORDS)] == REFERENCE_PROD_KEYWORDS, names
    # the override must not add keywords of its own (a caller of the reference could not have passed them)
    override = [s for s in kw_sigs if "seed" not in s]
    assert len(override) == 1 and _keyword_names(override[0]) == REFERENCE_PROD_KEYWORDS
    assert any(re.search(r"anParams\s*,\s*Niter::Int\s*$", s.strip()) for s in pos_sigs), "positional Niter method"


def _enable_body(code):
    start = re.search(r"function enable!\(", code).start()
    return code[start:code.index("overridden_methods() =")]


def test_enable_routes_default_rng_callers_to_the_device_rng():
    """enable!() overrides prodAppxMSGibbsS with `randU=nothing, randN=nothing` defaults (no host rand(...) of
    Np*Ndens*(Niter+2)*Nlevels doubles, src/MSGibbs01.jl:661-662) and the front end sends that case to
    kdehip_prod_philox; explicit streams go through the gibbs1 drop-in."""
    code = strip_code(open(SHIM).read())
    enable = _enable_body(code)
    assert "@eval KDE function prodAppxMSGibbsS" in enable and "@eval KDE function gibbs1" in enable
    override = [s for s in _signatures(enable, "prodAppxMSGibbsS")][0]
    assert re.search(r"randU\s*=\s*nothing", override) and re.search(r"randN\s*=\s*nothing", override)
    front = code[code.index("function prodAppxMSGibbsS("):re.search(r"function enable!\(", code).start()]
    assert ":kdehip_prod_philox" in front and "gibbs1(Ndens, trees, Np, Niter, points, indices, randU, randN" in front


def test_enable_overrides_the_whole_star_operator():
    """The reference's `*` is product THEN `kde!(pGM)` (src/MSGibbs01.jl:724-725 -> src/KDE01.jl:3-27 ->
    src/CrossValidation.jl:110-120), and every input of it came out of `kde!(points, ks[, weights])` (src/KDE01.jl:34-76 ->
    makeBallTreeDensity -> buildTree!, src/BallTree01.jl:415-434): enable!() must replace the product, the bandwidth search,
    the TREE CONSTRUCTORS and `evaluateDualTree` (src/DualTree01.jl:370-421), or an unchanged caller is left with a stage of
    `*` on the reference's single-threaded Julia path (missing 1).  The list of SEVEN overridden methods is
    pinned here."""
    code = strip_code(open(SHIM).read())
    enable = _enable_body(code)
    installed = [(m.group(1), enable[m.end():m.end() + 200]) for m in re.finditer(r"@eval KDE function\s+([A-Za-z_!0-9]+)\(", enable)]
    names = [n for n, _ in installed]
    assert names == ["gibbs1", "prodAppxMSGibbsS", "kde!", "kde!", "kde!", "evaluateDualTree", "evaluateDualTree"], names
    sig = [a for _, a in installed]
    # the three kde! methods carry the reference's own signatures (src/KDE01.jl:1-3, 64, 34-38): same signature = replaced
    assert "points::A" in sig[2] and "addop::Tuple" in sig[2] and "ks::" not in sig[2]
    assert "points::A" in sig[3] and "ks::Array{Float64,1}" in sig[3] and "addop::Tuple" in sig[3] and "weights" not in sig[3]
    assert "points::AbstractArray{<:Real,2}" in sig[4] and "ks::Array{Float64,1}" in sig[4] and "weights::Array{Float64,1}" in sig[4]
    assert "pos::Array{Float64,2}" in sig[5] and "pos::BallTreeDensity" in sig[6]
    # each override keeps the reference reachable: non-Euclidean operators, FORCE_EVAL_DIRECT = false, sizes beyond the limits
    for ref in ("reference_kde_auto", "reference_kde_bw", "reference_kde_bww", "reference_evaluateDualTree",
                "reference_evaluateDualTree_bd", "invoke_original"):
        assert ref in enable, ref
    assert "directEval" in enable and "isEuclidOps" in enable and "builds_here" in enable
    # ... through the world age in which the reference's methods were defined (gibbs1, prodAppxMSGibbsS, and ONE helper
    # for the five signature look-ups, each of which may miss without taking enable!() down)
    assert enable.count("Base.invoke_in_world") == 3
    assert enable.count("saved(orig") == 5 and "catch" in enable
    # the switches of the callers either side of the product
    # ... and they are OPT-IN: plain enable!() replaces only the two hot-path methods until someone has run
    # oracle/julia_crosscheck.jl --shim
    assert re.search(r"function enable!\(;\s*kde::Bool=false,\s*trees::Bool=false,\s*evaluate::Bool=false\)", code)
    # the library entries the overrides land on: ONE call for kde!(points), the pooled builder for the explicit forms
    body = code[:re.search(r"function enable!\(", code).start()]
    for sym in (":kdehip_make_density_auto", ":kdehip_make_density,", ":kdehip_auto_bandwidth", ":kdehip_evaluate",
                ":kdehip_mul_device", ":kdehip_density_download"):
        assert sym in body, sym
    # kde!(points) no longer ends in the reference's constructor
    auto = body[body.index("function kde!(points::AbstractMatrix{Float64}; device"):]
    auto = auto[:auto.index("\nend")]
    assert "KDE.kde!(" not in auto and ":kdehip_make_density_auto" in auto
    # the struct is assembled in ONE place, with `next` where buildTree! leaves it (src/BallTree01.jl:384-393,430)
    assert body.count("KDE.BallTree(") == 1 and "max(N, 2), KDE.swapDensity!" in body
    listed = re.search(r"overridden_methods\(\) = \[(.*?)\]", open(SHIM).read(), flags=re.S).group(1)
    assert listed.count('"') == 14


# ---- the overrides against the reference's ACTUAL source (skipped where /root/reference does not exist: the GPU box) ----
REFERENCE_SRC = "/root/reference/src"


def _split_top(text, sep=","):
    items, depth, cur = [], 0, ""
    for ch in text:
        if ch in "([{":
            depth += 1
        if ch in ")]}":
            depth -= 1
        if ch == sep and depth == 0:
            items.append(cur)
            cur = ""
        else:
            cur += ch
    if cur.strip():
        items.append(cur)
    return items


def _canonical(sig_text, where_text):
    """(positional 'name::Type' list without defaults or blanks, keyword NAMES, where clause without blanks)"""
    pos, _, kw = sig_text.partition(";")
    def strip_default(a):
        depth = 0
        for i, ch in enumerate(a):
            if ch in "([{":
                depth += 1
            if ch in ")]}":
                depth -= 1
            if ch == "=" and depth == 0 and a[i:i + 2] != "==" and a[i - 1] not in "<>!=":
                return a[:i]
        return a
    posl = [re.sub(r"\s+", "", strip_default(a)) for a in _split_top(pos)]
    kwl = [re.match(r"\s*([A-Za-z_][A-Za-z_0-9]*)", a).group(1) for a in _split_top(kw)] if kw.strip() else []
    # static parameters that no positional argument mentions belong to the keyword body method only (Julia's lowering keeps,
    # for the positional method and its kwcall twin, the `where` variables its positional arguments use: the reference's
    # T1..T4 of src/MSGibbs01.jl:650-653,664 annotate keywords) -- they do not take part in method replacement
    wvars = [re.sub(r"\s+", "", v) for v in _split_top(where_text.strip()[1:-1])] if where_text.strip() else []
    used = [v for v in wvars if any(re.search(r"\b" + re.match(r"[A-Za-z_][A-Za-z_0-9]*", v).group(0) + r"\b", a) for a in posl)]
    return tuple(posl), tuple(kwl), ",".join(used)


def _function_signatures(code, name, prefix=r"function\s+"):
    """[(signature text, where clause)] of every `function name(...) [where {...}]` in comment-free code"""
    out = []
    for m in re.finditer(prefix + re.escape(name) + r"\(", code):
        i, depth = m.end(), 1
        while depth:
            depth += {"(": 1, ")": -1}.get(code[i], 0)
            i += 1
        sig = code[m.end():i - 1]
        w = re.match(r"\s*where\s*(\{[^}]*\}(?:[^\n]*\})?|[A-Za-z_][^\n]*)", code[i:])
        where = ""
        if w:
            j, depth = i + code[i:].index("{"), 0
            k = j
            while True:
                depth += {"{": 1, "}": -1}.get(code[k], 0)
                k += 1
                if depth == 0:
                    break
            where = code[j:k]
        out.append((sig, where))
    return out


def test_overrides_carry_the_references_own_signatures():
    """Every method enable!() `@eval`s into KernelDensityEstimate must have, argument for argument, the positional
    signature (names, type annotations, `where` clause) of a method that exists in the reference's source -- the same
    signature is what makes the definition REPLACE the reference's method instead of adding a more or less specific one --
    and, for the two keyword methods, the reference's keyword names in its order.  Parsed from src/KDE01.jl,
    src/DualTree01.jl and src/MSGibbs01.jl themselves, not from strings in this file."""
    import pytest
    if not os.path.isdir(REFERENCE_SRC):
        pytest.skip("the reference's source is not on this machine")
    ref_code = {f: strip_code(open(os.path.join(REFERENCE_SRC, f)).read()) for f in ("KDE01.jl", "DualTree01.jl", "MSGibbs01.jl")}
    ref = {}
    for name, f in (("kde!", "KDE01.jl"), ("evaluateDualTree", "DualTree01.jl"), ("gibbs1", "MSGibbs01.jl"),
                    ("prodAppxMSGibbsS", "MSGibbs01.jl")):
        ref[name] = [_canonical(s, w) for s, w in _function_signatures(ref_code[f], name)]
        assert ref[name], name
    enable = _enable_body(strip_code(open(SHIM).read()))
    seen = []
    for name in ("gibbs1", "prodAppxMSGibbsS", "kde!", "evaluateDualTree"):
        for sig, where in _function_signatures(enable, name, prefix=r"@eval KDE function\s+"):
            pos, kw, wh = _canonical(sig, where)
            matches = [r for r in ref[name] if r[0] == pos and r[2] == wh]
            assert len(matches) == 1, f"{name}{pos} where {wh!r}: no method with this positional signature in the reference: {ref[name]}"
            if kw or matches[0][1]:
                # (the reference annotates its operator keywords with method type parameters T1..T4, :650-653, 664; names and
                # order are what a caller can observe)
                assert kw == matches[0][1], (name, kw, matches[0][1])
            seen.append((name, pos))
    assert len(seen) == 7, seen
    # the look-ups by concrete argument types name as many arguments as those methods have positional parameters
    for tup, n in ((r"Tuple\{Matrix\{Float64\},Pl,Mi\}", 3), (r"Tuple\{Matrix\{Float64\},Vector\{Float64\},Pl,Mi\}", 4),
                   (r"Tuple\{Matrix\{Float64\},Vector\{Float64\},Vector\{Float64\},Pl,Mi\}", 5),
                   (r"Tuple\{BallTreeDensity,Matrix\{Float64\},Bool,Float64,Pl,Mi\}", 6),
                   (r"Tuple\{BallTreeDensity,BallTreeDensity,Bool,Float64,Pl,Mi\}", 6)):
        assert re.search(tup, enable), tup
        assert any(len(p) == n for name, p in seen if name in ("kde!", "evaluateDualTree")), n
    # the deprecated positional-Niter front end (src/MSGibbs01.jl:632-636) exists in the reference with the shim's signature
    shim_pos = [_canonical(s, w) for s, w in _function_signatures(strip_code(open(SHIM).read()), "prodAppxMSGibbsS")
                if ";" not in s and "DeviceDensity" not in s]
    assert len(shim_pos) == 1 and shim_pos[0][0] in [r[0] for r in ref["prodAppxMSGibbsS"]]
