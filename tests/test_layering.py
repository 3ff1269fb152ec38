"""The oracle is test infrastructure: nothing under the package may import it, bench.py may only inside its
cpu_baseline leg, __graft_entry__ only inside smoke(); and the product loader has no CPU fallback."""
import ast
import glob
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _oracle_imports(path):
    """[(enclosing top-level function or '<module>', line)] of every import that names the oracle package."""
    tree = ast.parse(open(path).read(), path)
    found = []

    def visit(node, owner):
        for child in ast.iter_child_nodes(node):
            inner = owner
            if isinstance(child, (ast.FunctionDef, ast.AsyncFunctionDef)) and owner == '<module>':
                inner = child.name
            if isinstance(child, ast.Import) and any(a.name.split('.')[0] == 'oracle' for a in child.names):
                found.append((owner, child.lineno))
            if isinstance(child, ast.ImportFrom) and (child.module or '').split('.')[0] == 'oracle':
                found.append((owner, child.lineno))
            visit(child, inner)
    visit(tree, '<module>')
    return found


def test_package_never_imports_the_oracle():
    for path in glob.glob(os.path.join(ROOT, 'augmentedautoencoder_amd', '**', '*.py'), recursive=True):
        assert _oracle_imports(path) == [], path
        assert 'oracle' not in [w for line in open(path) for w in line.split() if line.lstrip().startswith(('import ', 'from '))], path


def test_bench_and_entry_use_the_oracle_only_as_the_checker():
    assert {o for o, _ in _oracle_imports(os.path.join(ROOT, 'bench.py'))} <= {'cpu_baseline'}
    assert {o for o, _ in _oracle_imports(os.path.join(ROOT, '__graft_entry__.py'))} <= {'smoke'}


def test_product_loader_has_no_fallback_library():
    src = open(os.path.join(ROOT, 'augmentedautoencoder_amd', '_lib.py')).read()
    assert 'libaae_emu' not in src and 'oracle' not in src and 'libaae_oracle' not in src
