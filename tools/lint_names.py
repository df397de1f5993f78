"""Names that a function reads but nothing defines (no linter in this image): python tools/lint_names.py [files...].
For every function scope: a symbol the compiler resolved as global must be assigned / imported / defined at module level, or be a builtin."""
import builtins, glob, os, symtable, sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")


def check(path):
    src = open(path).read()
    top = symtable.symtable(src, path, "exec")
    module_names = {s.get_name() for s in top.get_symbols() if s.is_assigned() or s.is_imported() or s.is_namespace()}
    module_names |= set(dir(builtins)) | {"__file__", "__name__", "__doc__", "__builtins__", "__spec__", "__package__"}
    star = "import *" in src
    bad = []

    def walk(tab):
        for child in tab.get_children():
            if child.get_type() in ("function", "class"):
                for s in child.get_symbols():
                    if s.is_global() and not s.is_local() and s.is_referenced() and not s.is_declared_global() and s.get_name() not in module_names and not star:
                        bad.append((child.get_lineno(), child.get_name(), s.get_name()))
                    if s.is_declared_global() and s.get_name() not in module_names and not s.is_assigned():
                        bad.append((child.get_lineno(), child.get_name(), s.get_name()))
            walk(child)
    walk(top)
    return bad


if __name__ == "__main__":
    files = sys.argv[1:] or sorted(glob.glob(os.path.join(ROOT, "seal_amd", "**", "*.py"), recursive=True) + glob.glob(os.path.join(ROOT, "*.py")) +
                                   glob.glob(os.path.join(ROOT, "oracle", "*.py")) + glob.glob(os.path.join(ROOT, "tools", "*.py")) +
                                   glob.glob(os.path.join(ROOT, "tests", "*.py")))
    n = 0
    for f in files:
        for line, fn, name in check(f):
            print(f"{os.path.relpath(f, ROOT)}:{line}: in {fn}: undefined name {name!r}")
            n += 1
    print(f"{len(files)} files, {n} undefined names")
    sys.exit(1 if n else 0)
