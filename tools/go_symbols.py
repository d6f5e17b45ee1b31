"""A small Go reader for the cgo shim check (the image has no Go toolchain).

Two uses:
  * `python tools/go_symbols.py /root/reference > tests/golden/go_symbols.json` lists what the reference's packages
    `internal/processor` and `internal/audio` DECLARE (type names, struct field names with their types, function and method names
    with result types, constants, variables).  Names and types only: data about the reference, not its source.
  * tests/test_go_shim_symbols.py holds `integration/go/gpu_engine.go` against that list and against include/*.h: every Go
    identifier the shim uses must be declared by the reference or by the shim itself, every `T{Field: ...}` key and every
    `x.Field` it can type must exist, a float / int / string / bool conversion must land in a field of that kind, and every
    `C.name` and C struct field must exist in the headers.
No Go semantics beyond declarations, composite literals and selector chains: enough to catch a renamed field or function, a wrong
kind, a typo.  It cannot prove the file compiles.
"""
import json
import os
import re
import sys


def strip_go(src):
    """Comments and string / rune literal CONTENTS blanked out (delimiters kept, length preserved)."""
    out = []
    i, n = 0, len(src)
    while i < n:
        c = src[i]
        if src.startswith("//", i):
            j = src.find("\n", i); j = n if j < 0 else j
            out.append(" " * (j - i)); i = j
        elif src.startswith("/*", i):
            j = src.find("*/", i + 2); j = n if j < 0 else j + 2
            out.append(re.sub(r"[^\n]", " ", src[i:j])); i = j
        elif c == '"':
            j = i + 1
            while j < n and src[j] != '"':
                j += 2 if src[j] == "\\" else 1
            out.append('"' + "_" * (j - i - 1) + '"'); i = j + 1
        elif c == "`":
            j = src.find("`", i + 1); j = n if j < 0 else j
            out.append("`" + re.sub(r"[^\n]", "_", src[i + 1:j]) + "`"); i = j + 1
        elif c == "'":
            j = i + 1
            while j < n and src[j] != "'":
                j += 2 if src[j] == "\\" else 1
            out.append("'" + "_" * (j - i - 1) + "'"); i = j + 1
        else:
            out.append(c); i += 1
    return "".join(out)


def match_brace(s, i, open_c="{", close_c="}"):
    """index of the brace closing the one at s[i]"""
    d = 0
    for j in range(i, len(s)):
        if s[j] == open_c:
            d += 1
        elif s[j] == close_c:
            d -= 1
            if d == 0:
                return j
    return len(s) - 1


def split_top(s, seps=",\n"):
    """split on separators at bracket depth 0"""
    parts, d, cur = [], 0, []
    for ch in s:
        if ch in "([{":
            d += 1
        elif ch in ")]}":
            d -= 1
        if ch in seps and d == 0:
            parts.append("".join(cur)); cur = []
        else:
            cur.append(ch)
    parts.append("".join(cur))
    return [p.strip() for p in parts if p.strip()]


def parse_struct_body(body):
    """{field: type} of a struct body (embedded fields as {"<embedded>TypeName": type})"""
    fields = {}
    for line in split_top(body, "\n;"):
        line = re.sub(r"`[^`]*`", "", line).strip()
        if not line:
            continue
        m = re.match(r"^((?:\w+\s*,\s*)*\w+)\s+(.+)$", line)
        if m and not re.match(r"^\*?[\w.]+$", line):
            typ = m.group(2).strip()
            if typ.startswith("struct"):
                typ = "struct"
            for nm in m.group(1).split(","):
                fields[nm.strip()] = typ
        else:
            t = line.lstrip("*")
            fields["<embedded>" + t.split(".")[-1]] = line
    return fields


def parse_go_decls(src):
    """Declarations of one Go file (already stripped)."""
    d = {"structs": {}, "types": {}, "funcs": {}, "methods": {}, "consts": [], "vars": {}}
    # types (single and grouped)
    def one_type(text):
        m = re.match(r"^(\w+)(\[[^\]]*\])?\s+(.*)$", text, re.S)
        if not m:
            return
        name, rest = m.group(1), m.group(3).strip()
        if rest.startswith("struct"):
            b = rest.find("{")
            d["structs"][name] = parse_struct_body(rest[b + 1:match_brace(rest, b)])
            d["types"][name] = "struct"
        else:
            d["types"][name] = rest.split("\n")[0].strip().lstrip("= ").strip()
    for m in re.finditer(r"(?m)^type\s+", src):
        i = m.end()
        if src[i] == "(":
            j = match_brace(src, i, "(", ")")
            body = src[i + 1:j]
            k = 0
            while k < len(body):
                mm = re.compile(r"\s*(\w+)").match(body, k)
                if not mm:
                    break
                # extent of this declaration: up to newline at depth 0
                depth, e = 0, mm.start()
                while e < len(body):
                    if body[e] in "{(":
                        depth += 1
                    elif body[e] in "})":
                        depth -= 1
                    elif body[e] == "\n" and depth == 0 and e > mm.end():
                        break
                    e += 1
                one_type(body[mm.start():e].strip()); k = e + 1
        else:
            depth, e = 0, i
            while e < len(src):
                if src[e] in "{(":
                    depth += 1
                elif src[e] in "})":
                    depth -= 1
                elif src[e] == "\n" and depth == 0:
                    break
                e += 1
            one_type(src[i:e].strip())
    # funcs and methods
    for m in re.finditer(r"(?m)^func\s+(\((?P<recv>[^)]*)\)\s*)?(?P<name>\w+)\s*(\[[^\]]*\])?\(", src):
        p0 = m.end() - 1
        p1 = match_brace(src, p0, "(", ")")
        params = src[p0 + 1:p1]
        rest = src[p1 + 1:src.find("\n", p1) if src.find("{", p1) < 0 else src.find("{", p1)]
        res = rest.strip()
        if res.startswith("("):
            res = res[1:match_brace(res, 0, "(", ")")]
        results = [re.sub(r"^\w+\s+(?=[\*\[\w])", "", r).strip() if re.match(r"^\w+\s+[\*\[\w]", r) else r for r in split_top(res, ",")] if res else []
        sig = {"params": params.strip(), "results": results}
        if m.group("recv"):
            rt = m.group("recv").split()[-1].lstrip("*")
            rt = re.sub(r"\[.*\]$", "", rt)
            d["methods"].setdefault(rt, {})[m.group("name")] = sig
        else:
            d["funcs"][m.group("name")] = sig
    # consts / vars (package level)
    for kw in ("const", "var"):
        for m in re.finditer(r"(?m)^%s\s+" % kw, src):
            i = m.end()
            if src[i] == "(":
                body = src[i + 1:match_brace(src, i, "(", ")")]
                lines = split_top(body, "\n")
            else:
                lines = [src[i:src.find("\n", i)]]
            for ln in lines:
                mm = re.match(r"^((?:\w+\s*,\s*)*\w+)\s*([^=]*)(=|$)", ln.strip())
                if not mm:
                    continue
                for nm in mm.group(1).split(","):
                    if kw == "const":
                        d["consts"].append(nm.strip())
                    else:
                        d["vars"][nm.strip()] = mm.group(2).strip()
    return d


def merge(a, b):
    for k in ("structs", "types", "funcs", "vars"):
        a[k].update(b[k])
    for t, ms in b["methods"].items():
        a["methods"].setdefault(t, {}).update(ms)
    a["consts"] = sorted(set(a["consts"]) | set(b["consts"]))
    return a


def package_symbols(directory):
    acc = {"structs": {}, "types": {}, "funcs": {}, "methods": {}, "consts": [], "vars": {}}
    for fn in sorted(os.listdir(directory)):
        if fn.endswith(".go") and not fn.endswith("_test.go"):
            merge(acc, parse_go_decls(strip_go(open(os.path.join(directory, fn), encoding="utf-8").read())))
    return acc


def reference_symbols(ref_root):
    out = {"source": "declarations of the reference's Go packages (names and types only), written by tools/go_symbols.py",
           "packages": {}}
    for pkg in ("internal/processor", "internal/audio"):
        out["packages"][pkg.split("/")[-1]] = package_symbols(os.path.join(ref_root, pkg))
    return out


# ---------------------------------------------------------------------------------------------------------- C headers
def strip_c(src):
    src = re.sub(r"/\*.*?\*/", lambda m: re.sub(r"[^\n]", " ", m.group(0)), src, flags=re.S)
    return re.sub(r"//[^\n]*", "", src)


def header_symbols(paths):
    """{"funcs": set, "structs": {name: {field: ctype}}, "consts": set, "types": set} of the C headers"""
    h = {"funcs": set(), "structs": {}, "consts": set(), "types": set()}
    for p in paths:
        s = strip_c(open(p).read())
        for m in re.finditer(r"#define\s+(\w+)", s):
            h["consts"].add(m.group(1))
        for m in re.finditer(r"\benum\b[^{;]*\{([^}]*)\}", s):
            for it in m.group(1).split(","):
                nm = it.split("=")[0].strip()
                if nm:
                    h["consts"].add(nm)
        for m in re.finditer(r"typedef\s+struct\s*(\w*)\s*\{", s):
            b = m.end() - 1
            e = match_brace(s, b)
            tail = re.match(r"\s*(\w+)\s*;", s[e + 1:])
            name = tail.group(1) if tail else m.group(1)
            fields = {}
            for decl in s[b + 1:e].split(";"):
                decl = " ".join(decl.split())
                if not decl:
                    continue
                mm = re.match(r"^(.*?[\s\*])((?:\**\w+(?:\[[^\]]*\])*\s*,\s*)*\**\w+(?:\[[^\]]*\])*)$", decl)
                if not mm:
                    continue
                for nm in mm.group(2).split(","):
                    nm = nm.strip()
                    ptr = nm.count("*")
                    base = re.sub(r"\[.*", "", nm.lstrip("*"))
                    arr = "[]" if "[" in nm else ""
                    fields[base] = mm.group(1).strip() + "*" * ptr + arr
            h["structs"][name] = fields; h["types"].add(name)
        for m in re.finditer(r"typedef\s+struct\s+(\w+)\s+(\w+)\s*;", s):
            h["types"].add(m.group(2))
        for m in re.finditer(r"typedef\s+[^;{]*?\(\s*\*\s*(\w+)\s*\)\s*\(", s):
            h["types"].add(m.group(1))
        for m in re.finditer(r"(?m)^\s*(?:extern\s+\"C\"\s+)?[\w\s\*]+?\b(\w+)\s*\([^;{]*\)\s*;", s):
            h["funcs"].add(m.group(1))
    return h


if __name__ == "__main__":
    root = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
    json.dump(reference_symbols(root), sys.stdout, indent=1, sort_keys=True)
    sys.stdout.write("\n")
