"""SURVEY section 8 f1 without a Go toolchain: `integration/go/gpu_engine.go` held against what the reference's packages declare
(tests/golden/go_symbols.json: type / field / function names and types, written from /root/reference by tools/go_symbols.py) and
against include/*.h.  Fails when the shim uses a Go type, struct field, function or method that neither the reference nor the shim
declares, puts a float64 / int / string / bool conversion into a field of another type, or names a C function, type, constant or
struct field the headers do not have.  It cannot prove the file compiles; it catches renames, typos and kind mismatches."""
import importlib.util
import json
import os
import re

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
SHIM = os.path.join(ROOT, "integration", "go", "gpu_engine.go")
SYMS = os.path.join(HERE, "golden", "go_symbols.json")

_sp = importlib.util.spec_from_file_location("go_symbols", os.path.join(ROOT, "tools", "go_symbols.py"))
G = importlib.util.module_from_spec(_sp); _sp.loader.exec_module(G)

BUILTIN_FUNCS = {"len", "cap", "make", "new", "append", "copy", "delete", "panic", "recover", "print", "println", "min", "max", "close", "clear",
                 "complex", "real", "imag"}
BUILTIN_TYPES = {"bool", "string", "int", "int8", "int16", "int32", "int64", "uint", "uint8", "uint16", "uint32", "uint64", "uintptr", "byte", "rune",
                 "float32", "float64", "error", "any"}
KEYWORDS = {"if", "for", "switch", "return", "func", "go", "defer", "select", "range", "case", "else", "var", "const", "type", "struct", "map", "chan",
            "interface", "import", "package", "break", "continue", "fallthrough", "goto", "default"}
CGO_BASICS = {"int", "uint", "long", "ulong", "char", "uchar", "short", "float", "double", "size_t", "int64_t", "uint64_t", "int32_t", "uint32_t", "int16_t",
              "uint16_t", "uint8_t", "int8_t", "GoString", "GoStringN", "GoBytes", "CString", "CBytes", "free", "calloc", "malloc", "memcpy", "memset", "strlen"}


class Env:
    def __init__(self):
        ref = json.load(open(SYMS))["packages"]
        self.ref = ref["processor"]
        self.audio = ref["audio"]
        self.src = G.strip_go(open(SHIM, encoding="utf-8").read())
        self.local = G.parse_go_decls(self.src)
        self.hdr = G.header_symbols([os.path.join(ROOT, "include", "jtgpu.h"), os.path.join(ROOT, "include", "jt_host.h")])
        pre = open(SHIM, encoding="utf-8").read()
        pre = pre[pre.index("/*"):pre.index('import "C"')]
        self.preamble_funcs = set(re.findall(r"\b(\w+)\s*\([^;{)]*\)\s*\{", pre)) | set(re.findall(r"extern\s+\w+\s+(\w+)\s*\(", pre))
        self.imports = set(re.findall(r'(?m)^\s*(?:\w+\s+)?"([\w/]+)"', open(SHIM, encoding="utf-8").read().split("import (")[1].split(")")[0]))
        self.import_names = {p.split("/")[-1] for p in self.imports}

    # ---- type lookups over reference + shim declarations
    def type_def(self, name):
        for d in (self.local, self.ref):
            if name in d["types"]:
                return d["types"][name]
        return None

    def struct_of(self, name, depth=0):
        """fields of a (possibly named-alias) struct type, embedded fields promoted"""
        if name is None or depth > 4:
            return None
        for d in (self.local, self.ref):
            if name in d["structs"]:
                f = dict(d["structs"][name])
                for k, v in list(f.items()):
                    if k.startswith("<embedded>"):
                        sub = self.struct_of(base(v), depth + 1)
                        f[k[len("<embedded>"):]] = v
                        for kk, vv in (sub or {}).items():
                            f.setdefault(kk, vv)
                return f
        td = self.type_def(name)
        if td and td != "struct" and re.match(r"^\w+$", td):
            return self.struct_of(td, depth + 1)
        return None

    def methods_of(self, name):
        out = {}
        for d in (self.local, self.ref):
            out.update(d["methods"].get(name, {}))
        return out

    def func(self, name):
        for d in (self.local, self.ref):
            if name in d["funcs"]:
                return d["funcs"][name]
        return None

    def known_type(self, name):
        return name in BUILTIN_TYPES or self.type_def(name) is not None


def base(t):
    """element / pointee type name of a Go type expression"""
    t = t.strip()
    while True:
        m = re.match(r"^(\*|\[[^\]]*\])", t)
        if not m:
            break
        t = t[m.end():]
    return t


@pytest.fixture(scope="module")
def env():
    return Env()


def test_symbol_fixture_is_current():
    """Where the reference is present (this container; not the GPU box) the committed list must equal a fresh one."""
    if not os.path.isdir("/root/reference/internal/processor"):
        pytest.skip("/root/reference absent: the committed tests/golden/go_symbols.json stands")
    assert G.reference_symbols("/root/reference") == json.load(open(SYMS)), "run: python tools/go_symbols.py /root/reference > tests/golden/go_symbols.json"


def test_the_seam_functions_have_the_reference_signatures(env):
    """ProcessAudioGPU / AnalyseOnlyDetailedGPU are drop-ins for ProcessAudio / AnalyseOnlyDetailed (processor.go:29,78; the
    reflection test processor_test.go:287-356 pins those signatures)."""
    strip = lambda s: re.sub(r"\s+", " ", s).strip()
    for ours, theirs in (("ProcessAudioGPU", "ProcessAudio"), ("AnalyseOnlyDetailedGPU", "AnalyseOnlyDetailed")):
        a, b = env.local["funcs"][ours], env.ref["funcs"][theirs]
        assert [strip(p).split(" ", 1)[-1] for p in G.split_top(a["params"], ",")] == [strip(p).split(" ", 1)[-1] for p in G.split_top(b["params"], ",")], ours
        assert a["results"] == b["results"], ours


def test_every_c_name_exists_in_the_headers(env):
    bad = []
    for m in re.finditer(r"\bC\.(\w+)", env.src):
        n = m.group(1)
        if n in CGO_BASICS or n in env.preamble_funcs:
            continue
        if n in env.hdr["funcs"] or n in env.hdr["types"] or n in env.hdr["consts"] or n in env.hdr["structs"]:
            continue
        bad.append(n)
    assert not bad, f"C names used by the shim but absent from include/*.h: {sorted(set(bad))}"


def _functions(env):
    """(name, receiver type, params text, results, body) of every function in the shim"""
    out = []
    for m in re.finditer(r"(?m)^func\s+(\((?P<recv>[^)]*)\)\s*)?(?P<name>\w+)\s*\(", env.src):
        p0 = m.end() - 1
        p1 = G.match_brace(env.src, p0, "(", ")")
        b0 = env.src.find("{", p1)
        # results may contain parentheses but no braces
        b1 = G.match_brace(env.src, b0)
        sig = (env.local["methods"].get(base(m.group("recv").split()[-1]), {}) if m.group("recv") else env.local["funcs"]).get(m.group("name"))
        out.append((m.group("name"), m.group("recv"), env.src[p0 + 1:p1], sig["results"] if sig else [], env.src[b0 + 1:b1], env.src[p1 + 1:b0]))
    return out


def _params(text):
    """{name: type} of a Go parameter list ("a, b T, c *U")"""
    vars_, pending = {}, []
    for p in G.split_top(text, ","):
        m = re.match(r"^(\w+)\s+(.+)$", p.strip())
        if m:
            for q in pending:
                vars_[q] = m.group(2).strip()
            pending = []
            vars_[m.group(1)] = m.group(2).strip()
        else:
            pending.append(p.strip())
    return vars_


def _local_vars(env, recv, params, results_text, body):
    v = {}
    if recv:
        v.update(_params(recv))
    v.update(_params(params))
    rt = results_text.strip()
    if rt.startswith("(") and re.search(r"\w+\s+[\*\[\w]", rt):
        v.update({k: t for k, t in _params(rt[1:G.match_brace(rt, 0, "(", ")")]).items() if k not in BUILTIN_TYPES})
    for m in re.finditer(r"\bvar\s+((?:\w+\s*,\s*)*\w+)\s+((?:\[[^\]]*\]|\*)*[\w.]+)", body):
        for nm in m.group(1).split(","):
            v[nm.strip()] = m.group(2)
    for m in re.finditer(r"\b(\w+)\s*:=\s*&?((?:C\.)?\w+)\{", body):
        v[m.group(1)] = m.group(2)
    for m in re.finditer(r"\b(\w+)\s*:=\s*new\(((?:C\.)?\w+)\)", body):
        v[m.group(1)] = "*" + m.group(2)
    for m in re.finditer(r"\b(\w+)\s*:=\s*\(\*(C\.\w+)\)\(", body):
        v[m.group(1)] = "*" + m.group(2)
    for m in re.finditer(r"\b(\w+)\s*:=\s*make\((\[[^\]]*\](?:C\.)?[\w.]+)", body):
        v[m.group(1)] = m.group(2)
    for m in re.finditer(r"\b((?:\w+\s*,\s*)*\w+)\s*:=\s*(\w+)\(", body):
        f = env.func(m.group(2))
        if f:
            for nm, t in zip([q.strip() for q in m.group(1).split(",")], f["results"]):
                if nm != "_":
                    v[nm] = t
    for m in re.finditer(r"\b((?:\w+\s*,\s*)*\w+)\s*:=\s*(\w+)\.(\w+)\(", body):
        t = v.get(m.group(2))
        meth = env.methods_of(base(t)).get(m.group(3)) if t else None
        if meth:
            for nm, tt in zip([q.strip() for q in m.group(1).split(",")], meth["results"]):
                if nm != "_":
                    v[nm] = tt
    # no scopes here: a name declared more than once in the function (or re-declared over a parameter) is left untyped
    first = dict(_params(recv) if recv else {}); first.update(_params(params))
    for nm in list(v):
        decls = len(re.findall(r"(?<![\w.])%s\s*(?:,\s*\w+\s*)*:=" % re.escape(nm), body)) + len(re.findall(r"(?:,\s*)%s\s*(?:,\s*\w+\s*)*:=" % re.escape(nm), body)) \
            + len(re.findall(r"\bvar\s+(?:\w+\s*,\s*)*%s\b" % re.escape(nm), body)) + (1 if nm in first else 0)
        if decls > 1:
            del v[nm]
    return v


def _walk(env, typ, chain, where, errors):
    """follow `.a.b[i].c` from a value of type `typ`; report a field / method the type does not have"""
    for part in re.findall(r"\.\w+|\[[^\]]*\]", chain):
        if typ is None:
            return
        if part.startswith("["):
            m = re.match(r"^\[[^\]]*\](.*)$", typ.lstrip("*"))
            typ = m.group(1) if m else None
            continue
        name = part[1:]
        b = base(typ) if not typ.lstrip("*").startswith("[") else None
        if b is None:
            return
        if b.startswith("C."):
            fields = env.hdr["structs"].get(b[2:])
            if fields is None:
                return
            key = name[1:] if name.startswith("_") and name[1:] in fields else name       # cgo spells C keywords / clashes with a leading underscore
            if key not in fields:
                errors.append(f"{where}: C struct {b[2:]} has no field '{name}'")
                return
            ct = fields[key]
            arr = ct.endswith("[]")
            ct = ct[:-2] if arr else ct
            ptr = ct.count("*")
            core = ct.replace("*", "").replace("const ", "").replace("struct ", "").strip()
            typ = ("[]" if arr else "") + "*" * ptr + ("C." + core if core in env.hdr["structs"] else core)
            continue
        fields = env.struct_of(b)
        meths = env.methods_of(b)
        if fields is None and not meths:
            return
        if fields and name in fields:
            typ = fields[name]
        elif name in meths:
            r = meths[name]["results"]
            typ = r[0] if r else None
        else:
            if fields is not None:
                errors.append(f"{where}: type {b} has no field or method '{name}'")
            return


def test_selectors_and_literals_name_existing_fields(env):
    errors = []
    for name, recv, params, results, body, results_text in _functions(env):
        v = _local_vars(env, recv, params, results_text, body)
        # selector chains rooted at a typed local
        for m in re.finditer(r"(?<![\w.])([A-Za-z_]\w*)((?:\.\w+|\[[^\]\n]*\])+)", body):
            root = m.group(1)
            if root in v and root not in env.import_names and root != "C":
                _walk(env, v[root], m.group(2), f"{name}: {root}{m.group(2)}", errors)
        # composite literals of known struct types
        for m in re.finditer(r"(?<![\w.])((?:C\.)?\w+)\{", body):
            t = m.group(1)
            fields = env.hdr["structs"].get(t[2:]) if t.startswith("C.") else env.struct_of(t)
            if fields is None:
                continue
            b0 = m.end() - 1
            inner = body[b0 + 1:G.match_brace(body, b0)]
            for ent in G.split_top(inner, ","):
                km = re.match(r"^(\w+)\s*:\s*(.+)$", ent, re.S)
                if not km:
                    continue
                key, val = km.group(1), km.group(2).strip()
                if key not in fields:
                    errors.append(f"{name}: {t}{{...}} has no field '{key}'")
                    continue
                ft = fields[key]
                if t.startswith("C."):
                    continue
                kind = None
                cm = re.match(r"^(float64|float32|int|int64|int32|string|bool|time\.Duration)\((.*)\)$", val, re.S)
                if cm and G.match_brace(val, val.index("("), "(", ")") == len(val) - 1:
                    kind = cm.group(1)
                elif re.match(r"^C\.GoString\(", val) and G.match_brace(val, val.index("("), "(", ")") == len(val) - 1:
                    kind = "string"
                elif re.search(r"(!=|==|>|<|>=|<=)\s*[\w.]+$", val) and "(" not in val:
                    kind = "bool"
                elif val.startswith('"') and val.endswith('"'):
                    kind = "string?"
                if kind == "string?":
                    if ft not in ("string",) and env.type_def(ft) not in ("string",):
                        errors.append(f"{name}: {t}.{key} is {ft}, assigned a string literal")
                elif kind and ft != kind:
                    errors.append(f"{name}: {t}.{key} is {ft}, assigned {kind}(...)")
    assert not errors, "\n".join(errors)


def test_every_called_function_and_named_type_is_declared(env):
    errors = []
    for name, recv, params, results, body, results_text in _functions(env):
        v = _local_vars(env, recv, params, results_text, body)
        for m in re.finditer(r"(?<![\w.])([A-Za-z_]\w*)\s*\(", body):
            f = m.group(1)
            if f in KEYWORDS or f in BUILTIN_FUNCS or f in BUILTIN_TYPES or f in v:
                continue
            if env.func(f) or env.known_type(f):
                continue
            if re.search(r"\b%s\s*:?=\s*func\b" % re.escape(f), body):
                continue
            errors.append(f"{name}: calls '{f}', which neither the reference package nor the shim declares")
        # `&T{`, `*T`, `[]T`, `T{` with a capitalised or package-local type name
        for m in re.finditer(r"(?:&|\*|\]|\bnew\()\s*([A-Z]\w*)\b(?!\.)", body + " " + params + " " + " ".join(results)):
            t = m.group(1)
            if not env.known_type(t) and t not in v:
                errors.append(f"{name}: names type '{t}', which neither the reference package nor the shim declares")
    # package-qualified reference types the shim mentions (audio.X)
    for m in re.finditer(r"\baudio\.(\w+)", env.src):
        if m.group(1) not in env.audio["types"] and m.group(1) not in env.audio["funcs"]:
            errors.append(f"audio.{m.group(1)} is not declared by internal/audio")
    assert not errors, "\n".join(sorted(set(errors)))


def test_the_checker_catches_a_renamed_field_and_a_wrong_kind(env, monkeypatch):
    """The checks above are not vacuous: a renamed reference field, a float put into an int field and an unknown C function are each
    reported."""
    src = open(SHIM, encoding="utf-8").read()
    broken = src.replace("InputLRA:    parsedFloat", "InputLoudnessRange: parsedFloat", 1)
    assert broken != src
    broken = broken.replace("C.jt_pass3(", "C.jt_pass_three(", 1)
    broken = re.sub(r"Progress:\s*float64\(u\.progress\)", "Progress: int(u.progress)", broken, count=1)
    tmp = SHIM + ".broken-selftest"
    try:
        open(tmp, "w", encoding="utf-8").write(broken)
        monkeypatch.setattr(__import__(__name__), "SHIM", tmp, raising=False)
        import sys
        monkeypatch.setattr(sys.modules[__name__], "SHIM", tmp)
        e2 = Env()
        with pytest.raises(AssertionError) as a1:
            test_selectors_and_literals_name_existing_fields(e2)
        assert "InputLoudnessRange" in str(a1.value) and "Progress is float64, assigned int" in str(a1.value)
        with pytest.raises(AssertionError) as a2:
            test_every_c_name_exists_in_the_headers(e2)
        assert "jt_pass_three" in str(a2.value)
    finally:
        os.unlink(tmp)
