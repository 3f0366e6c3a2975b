"""Code-object metadata of every kernel in the built translation units (monoforce_amd/csrc/*.o): registers, LDS, scratch.

    python tools/kernel_metadata.py [--scratch]        (--scratch: only the kernels with a non-zero private segment)

Each object carries one gfx950 code object in its `.hip_fatbin` section: llvm-objcopy dumps the section, clang-offload-bundler
unbundles the code object, llvm-readelf prints its AMDGPU metadata notes."""
import glob
import os
import re
import subprocess
import sys
import tempfile

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = '/opt/rocm/lib/llvm/bin'


def kernels(csrc=os.path.join(REPO, 'monoforce_amd', 'csrc')):
    """[(object file, demangled kernel name, dict(vgpr, agpr, sgpr, lds, scratch))] over all objects."""
    rows = []
    for o in sorted(glob.glob(os.path.join(csrc, '*.o'))):
        with tempfile.TemporaryDirectory() as td:
            fb, co = os.path.join(td, 'fb'), os.path.join(td, 'co')
            if subprocess.run([f'{LLVM}/llvm-objcopy', '--dump-section', f'.hip_fatbin={fb}', o], capture_output=True).returncode:
                continue      # (a host-only object)
            r = subprocess.run([f'{LLVM}/clang-offload-bundler', '--unbundle', '--type=o', '--targets=hipv4-amdgcn-amd-amdhsa--gfx950',
                                f'--input={fb}', f'--output={co}'], capture_output=True, text=True)
            if r.returncode:
                raise RuntimeError(f'{o}: {r.stderr}')
            notes = subprocess.run([f'{LLVM}/llvm-readelf', '--notes', co], capture_output=True, text=True, check=True).stdout
        for blk in notes.split('  - .agpr_count:')[1:]:
            f = lambda k: int(re.search(r'\.' + k + r':\s+(\d+)', blk).group(1))  # noqa: E731
            rows.append((os.path.basename(o), re.search(r'\.name:\s+(\S+)', blk).group(1),
                         dict(agpr=int(re.match(r'\s*(\d+)', blk).group(1)), vgpr=f('vgpr_count'), sgpr=f('sgpr_count'), lds=f('group_segment_fixed_size'),
                              scratch=f('private_segment_fixed_size'))))
    names = subprocess.run(['c++filt'], input='\n'.join(n for _, n, _ in rows), capture_output=True, text=True).stdout.split('\n')
    return [(o, re.sub(r'\(.*', '', n).replace('void ', ''), m) for (o, _, m), n in zip(rows, names)]


if __name__ == '__main__':
    only = '--scratch' in sys.argv
    for o, n, m in kernels():
        if not only or m['scratch']:
            print(f'{o:36s} vgpr {m["vgpr"]:4d} agpr {m["agpr"]:3d} sgpr {m["sgpr"]:3d} lds {m["lds"]:6d} scratch {m["scratch"]:5d}  {n[:120]}')
