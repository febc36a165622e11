"""Static check of julia/BatchedSingleRoom.jl against include/rcw.h.

The Julia binding cannot be executed in this pipeline (no Julia toolchain), so it is checked mechanically:
every `ccall((:name, librcw), Ret, (ArgTypes...), ...)` in the file must name a function the header declares, with
the same arity, the same return kind and, argument by argument, the same pointer/scalar kind and element width;
every function the header declares must be bound at least once; and the Julia mirror of `struct rcw_config` must
list the same fields with the same types in the same order.  CPU only.
"""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "rcw.h")
BINDING = os.path.join(ROOT, "julia", "BatchedSingleRoom.jl")

C_SCALARS = {"int": "i32", "int32_t": "i32", "uint32_t": "u32", "int64_t": "i64", "uint64_t": "u64", "float": "f32",
             "double": "f64", "uint8_t": "u8", "char": "u8"}
JL_SCALARS = {"Cint": "i32", "Int32": "i32", "UInt32": "u32", "Int64": "i64", "UInt64": "u64", "Float32": "f32",
              "Float64": "f64", "UInt8": "u8", "Cchar": "u8"}


def strip_comments(text):
    return re.sub(r"/\*.*?\*/", " ", text, flags=re.S)


def c_kind(t):
    """'const float*' -> ('ptr', 'f32'); 'rcw_handle**' -> ('ptrptr', 'handle'); 'int32_t' -> ('scalar', 'i32')."""
    t = t.replace("const", " ").strip()
    stars = t.count("*")
    base = t.replace("*", "").strip()
    elem = {"rcw_handle": "handle", "rcw_config": "config", "void": "void"}.get(base) or C_SCALARS[base]
    return ({0: "scalar", 1: "ptr", 2: "ptrptr"}[stars], elem)


def header_prototypes():
    text = strip_comments(open(HEADER).read())
    protos = {}
    for m in re.finditer(r"RCW_API\s+(const\s+char\s*\*|int)\s+(rcw_\w+)\s*\(([^)]*)\)\s*;", text):
        ret, name, params = m.group(1), m.group(2), m.group(3).strip()
        args = []
        if params and params != "void":
            for p in params.split(","):
                p = " ".join(p.split())
                tm = re.match(r"(.*?)(\w+)$", p)          # type, then the parameter's name
                args.append(c_kind(tm.group(1)))
        protos[name] = ("cstring" if "char" in ret else "i32", args)
    return protos


def split_top_level(s):
    out, depth, cur = [], 0, ""
    for ch in s:
        if ch in "{(":
            depth += 1
        elif ch in "})":
            depth -= 1
        if ch == "," and depth == 0:
            out.append(cur.strip()); cur = ""
        else:
            cur += ch
    if cur.strip():
        out.append(cur.strip())
    return out


def jl_kind(t):
    t = t.strip()
    if t == "Cstring":
        return ("ptr", "u8")
    m = re.match(r"^(Ptr|Ref)\{(.*)\}$", t)
    if m:
        inner = m.group(2).strip()
        if re.match(r"^(Ptr|Ref)\{", inner):
            return ("ptrptr", "void" if "Cvoid" in inner else "?")
        elem = {"Cvoid": "void", "RcwConfig": "config"}.get(inner) or JL_SCALARS[inner]
        return ("ptr", elem)
    return ("scalar", JL_SCALARS[t])


def julia_ccalls():
    text = open(BINDING).read()
    calls = []
    for m in re.finditer(r"ccall\(\(:(\w+),\s*librcw\),\s*(\w+),\s*\(", text):
        name, ret = m.group(1), m.group(2)
        i, depth = m.end(), 1                       # scan to the matching ')' of the argument-type tuple
        while depth:
            depth += {"(": 1, ")": -1}.get(text[i], 0)
            i += 1
        tup = text[m.end():i - 1]
        calls.append((name, ret, [jl_kind(t) for t in split_top_level(tup)], text.count("\n", 0, m.start()) + 1))
    return calls


def compatible(c, j):
    """Is Julia argument kind j acceptable for C parameter kind c?"""
    (ck, ce), (jk, je) = c, j
    if ck != jk:
        return False
    if ck == "scalar":
        return ce == je
    if ck == "ptrptr":
        return je == "void"                                    # Ref{Ptr{Cvoid}} for rcw_handle** / void**
    if ce == "handle":
        return je == "void"                                    # the opaque handle travels as Ptr{Cvoid}
    if ce == "void":
        return je == "void"
    return ce == je                                            # typed pointers: same element type


def test_every_ccall_matches_its_prototype():
    protos = header_prototypes()
    assert len(protos) >= 60
    calls = julia_ccalls()
    assert len(calls) >= len(protos)
    for name, ret, args, line in calls:
        assert name in protos, f"BatchedSingleRoom.jl:{line}: {name} is not declared in rcw.h"
        cret, cargs = protos[name]
        assert (ret == "Cstring") == (cret == "cstring") and (ret in ("Cint", "Cstring")), f"{name} (line {line}): return type {ret}"
        assert len(args) == len(cargs), f"{name} (line {line}): {len(args)} arguments, the header has {len(cargs)}"
        for k, (c, j) in enumerate(zip(cargs, args)):
            assert compatible(c, j), f"{name} (line {line}): argument {k + 1} is {j} in Julia, {c} in C"


def test_every_export_is_bound():
    bound = {c[0] for c in julia_ccalls()}
    missing = sorted(set(header_prototypes()) - bound)
    assert not missing, f"declared in rcw.h but not bound in BatchedSingleRoom.jl: {missing}"


def test_config_struct_mirror():
    text = strip_comments(open(HEADER).read())
    body = re.search(r"typedef struct rcw_config \{(.*?)\} rcw_config;", text, flags=re.S).group(1)
    c_fields = []
    for decl in body.split(";"):
        decl = " ".join(decl.split())
        if not decl:
            continue
        m = re.match(r"(\w+) (\w+)(?:\[(\d+)\])?$", decl)
        c_fields.append((m.group(2), C_SCALARS[m.group(1)], int(m.group(3) or 1)))
    jl = open(BINDING).read()
    sbody = re.search(r"mutable struct RcwConfig\n(.*?)\nend", jl, flags=re.S).group(1)
    j_fields = []
    for line in sbody.splitlines():
        m = re.match(r"\s*(\w+)::(NTuple\{(\d+), (\w+)\}|\w+)", line)
        if m:
            if m.group(3):
                j_fields.append((m.group(1), JL_SCALARS[m.group(4)], int(m.group(3))))
            else:
                j_fields.append((m.group(1), JL_SCALARS[m.group(2)], 1))
    assert j_fields == c_fields
    size = {"i32": 4, "u32": 4, "f32": 4, "i64": 8, "u64": 8, "f64": 8}
    off = 0
    for _, t, n in c_fields:                                    # natural alignment, as both compilers lay it out
        off = (off + size[t] - 1) // size[t] * size[t] + size[t] * n
    assert off == 160
    assert re.search(r"const RCW_ABI_VERSION = (\d+)", jl).group(1) == re.search(r"#define RCW_ABI_VERSION (\d+)", text).group(1)


def test_rng_keyword_draws_in_the_reference_order():
    """The `rng` keyword (src/single_room.jl:49,265) of the binding: constructor and reset! take it, and reset_from_rng!
    makes the reference's draws with the reference's own statements in the reference's order — goal row, goal column,
    RCW.sample_empty_position on a host tile map that holds the wall ring and the NEW goal, heading — before ONE set_state!."""
    text = open(BINDING).read()
    assert re.search(r"function BatchedSingleRoom\(batch::Integer;[^)]*\brng = nothing", text)
    assert re.search(r"function RCW\.reset!\(env::BatchedSingleRoom;[^)]*\brng = env\.rng", text)
    body = text[text.index("function reset_from_rng!"):]
    body = body[:body.index("\nend\n")]
    order = [body.index(k) for k in ("rand(g, 2 : H - 1), rand(g, 2 : W - 1)", "tile_map[2, goal_position] = true",
                                     "RCW.sample_empty_position(g, tile_map)", "rand(g, 0 : nd - 1)", "set_state!(env, goal, position, direction")]
    assert order == sorted(order)
    assert "tile_map[1, :, 1] .= true" in body and "construction ? (1, 2) : (2,)" in body      # wall ring; the constructor draws twice
    fixtures = open(os.path.join(ROOT, "julia", "make_reference_fixtures.jl")).read()
    assert "Random.MersenneTwister(1)" in fixtures and "rng = rng" in fixtures and "RCW.reset!(env)" in fixtures
