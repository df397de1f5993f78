"""No function of the product, the oracle, bench.py or the tools reads a name that nothing defines (tools/lint_names.py): the GPU-only
branches are never executed by the CPU tests, a typo there would otherwise first show on an MI355X."""
import glob
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_no_undefined_names():
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    try:
        import lint_names
    finally:
        sys.path.pop(0)
    files = sorted(glob.glob(os.path.join(ROOT, "seal_amd", "**", "*.py"), recursive=True) + glob.glob(os.path.join(ROOT, "*.py")) +
                   glob.glob(os.path.join(ROOT, "oracle", "*.py")) + glob.glob(os.path.join(ROOT, "tools", "*.py")) +
                   glob.glob(os.path.join(ROOT, "tests", "*.py")))
    assert len(files) > 50
    bad = [(os.path.relpath(f, ROOT),) + b for f in files for b in lint_names.check(f)]
    assert not bad, bad
    # the checker itself notices a typo
    import tempfile
    with tempfile.NamedTemporaryFile("w", suffix=".py", delete=False) as t:
        t.write("import os\n\ndef top(a):\n    b = a + 1\n    return [x for x in range(b)]\n\ndef f(x):\n    if x:\n        return undefined_thing(x)\n    return os.sep\n")
    try:
        assert lint_names.check(t.name) == [(7, "f", "undefined_thing")]
    finally:
        os.unlink(t.name)
