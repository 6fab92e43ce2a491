"""The oracle is test infrastructure: only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import or
execute anything under oracle/ (the round prompt's rule; a product path that routes through the oracle voids every parity
claim).  Checked statically: no file of the package, of tools/ or of julia/ mentions the oracle package in an import, and
the two root files that do keep it inside the functions that are allowed to."""
import ast
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _imports_oracle(path):
    tree = ast.parse(open(path, encoding="utf-8").read())
    hits = []
    for node in ast.walk(tree):
        if isinstance(node, ast.Import):
            hits += [(node.lineno, a.name) for a in node.names if a.name.split(".")[0] == "oracle"]
        elif isinstance(node, ast.ImportFrom) and node.module and node.module.split(".")[0] == "oracle":
            hits.append((node.lineno, node.module))
    return hits


def _py_files(d):
    for base, dirs, files in os.walk(os.path.join(ROOT, d)):
        dirs[:] = [x for x in dirs if x != "__pycache__"]
        for f in files:
            if f.endswith(".py"):
                yield os.path.join(base, f)


def test_package_and_tools_never_import_the_oracle():
    bad = {}
    for d in ("stheno.jl_amd", "tools"):
        for p in _py_files(d):
            h = _imports_oracle(p)
            if h:
                bad[os.path.relpath(p, ROOT)] = h
            src = open(p, encoding="utf-8").read()
            if re.search(r"importlib\.import_module\(\s*['\"]oracle", src) or re.search(r"__import__\(\s*['\"]oracle", src):
                bad[os.path.relpath(p, ROOT)] = "dynamic import"
    assert not bad, bad


def _enclosing_functions(path):
    """{line of an oracle import: name of the top-level function it sits in (None: module level)}"""
    tree = ast.parse(open(path, encoding="utf-8").read())
    out = {}
    for top in tree.body:
        for node in ast.walk(top):
            mod = None
            if isinstance(node, ast.Import):
                mod = [a.name for a in node.names if a.name.split(".")[0] == "oracle"]
            elif isinstance(node, ast.ImportFrom) and node.module and node.module.split(".")[0] == "oracle":
                mod = [node.module]
            if mod:
                out[node.lineno] = top.name if isinstance(top, (ast.FunctionDef, ast.AsyncFunctionDef)) else None
    return out


def test_root_files_keep_the_oracle_inside_the_allowed_functions():
    where = _enclosing_functions(os.path.join(ROOT, "bench.py"))
    assert where and set(where.values()) <= {"cpu_baseline"}, where          # the untimed CPU-baseline leg only
    where = _enclosing_functions(os.path.join(ROOT, "__graft_entry__.py"))
    assert set(where.values()) <= {"smoke", "build"}, where                  # smoke() checks against it; build() imports it
    # (build(): "building the checker is not using it")


def test_c_sources_do_not_know_the_oracle():
    csrc = os.path.join(ROOT, "stheno.jl_amd", "csrc")
    for f in os.listdir(csrc):
        if f.endswith((".hip", ".h", ".cpp", "Makefile")):
            for n, line in enumerate(open(os.path.join(csrc, f), encoding="utf-8", errors="ignore"), 1):
                code = line.split("//")[0]          # a comment may cite the file a formula is restated in
                assert "oracle" not in code, (f, n, line)
