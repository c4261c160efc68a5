"""Build libnrx.so (the C-ABI HIP library) in-tree for gfx950 with hipcc.

    python -m neoradium_amd.build [--force]          (NRX_FORCE_BUILD=1 does the same for __graft_entry__.build())

One translation unit per csrc/*.hip, linked into neoradium_amd/libnrx.so.  No torch, no cmake: plain hipcc.
`-ffp-contract=off` keeps float64 paths bit-identical to the NumPy reference (no FMA contraction).
"""
import hashlib
import os
import subprocess
import sys
import glob
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
OBJ = os.path.join(CSRC, 'obj')
LIB = os.path.join(HERE, 'libnrx.so')
STAMP = os.path.join(HERE, 'libnrx.stamp')
ISA_JSON = os.path.join(HERE, 'libnrx.isa.json')      # instruction counts of the decoder kernels of THIS binary (tools/isa_mix.py), untracked
FLAGS = ['-std=c++20', '-O3', '--offload-arch=gfx950', '-ffp-contract=off', '-fno-slp-vectorize', '-mllvm', '-amdgpu-sdwa-peephole=0', '-fPIC', '-Wno-comment',
         '-Wno-unused-value']


def _newer(src, dst, deps):
    if not os.path.exists(dst):
        return True
    t = os.path.getmtime(dst)
    return any(os.path.getmtime(d) > t for d in [src] + deps)


def _hipcc():
    return os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')


_compiler_id = None


def compiler_id():
    """`hipcc --version` (the compiler is part of what produces the binary)."""
    global _compiler_id
    if _compiler_id is None:
        try:
            _compiler_id = subprocess.run([_hipcc(), '--version'], capture_output=True, text=True).stdout.strip() or 'unknown'
        except OSError:
            _compiler_id = 'unknown'
    return _compiler_id


def flags_hash():
    return hashlib.sha256((' '.join(FLAGS) + '\n' + compiler_id()).encode()).hexdigest()


def lib_hash():
    return hashlib.sha256(open(LIB, 'rb').read()).hexdigest() if os.path.exists(LIB) else ''


def source_hash(with_compiler=True):
    """SHA-256 over the compile flags, the compiler's version and every source / header the library is built from."""
    h = hashlib.sha256((' '.join(FLAGS) + ('\n' + compiler_id() if with_compiler else '')).encode())
    for f in sorted(glob.glob(os.path.join(CSRC, '*.hip')) + glob.glob(os.path.join(CSRC, '*.h'))
                    + glob.glob(os.path.join(HERE, '..', 'include', '*.h'))):
        h.update(os.path.basename(f).encode())
        h.update(open(f, 'rb').read())
    return h.hexdigest()


def build(force=False, verbose=True):
    """Compile what is out of date (by modification time; everything when the flags or the compiler changed) and link.  `force` (or
    NRX_FORCE_BUILD=1 in the environment) recompiles everything.  neoradium_amd/libnrx.stamp records the hash of the sources + flags
    + compiler the library was linked from AND the SHA-256 of libnrx.so itself: `stamp_ok()` tells whether the library in the tree
    is the file that build produced from exactly these sources."""
    force = force or os.environ.get('NRX_FORCE_BUILD', '') not in ('', '0')
    hipcc = _hipcc()
    os.makedirs(OBJ, exist_ok=True)
    fh_path = os.path.join(OBJ, 'flags.hash')
    if not force and stamp_ok():
        return LIB          # the library in the tree was linked from exactly these sources, flags and compiler (objects may be absent: GPU box)
    if not (os.path.exists(fh_path) and open(fh_path).read().strip() == flags_hash()):
        force = True        # objects of other flags / another compiler (or of unknown origin) are not reused

    srcs = sorted(glob.glob(os.path.join(CSRC, '*.hip')))
    deps = glob.glob(os.path.join(CSRC, '*.h')) + glob.glob(os.path.join(HERE, '..', 'include', '*.h'))
    jobs = []
    objs = []
    for s in srcs:
        o = os.path.join(OBJ, os.path.basename(s)[:-4] + '.o')
        objs.append(o)
        if force or _newer(s, o, deps):
            jobs.append([hipcc] + FLAGS + ['-c', s, '-o', o])

    def run(cmd):
        if verbose:
            print('[nrx build]', ' '.join(os.path.relpath(c) if os.path.exists(c) else c for c in cmd[-3:]), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError('hipcc failed:\n' + ' '.join(cmd) + '\n' + r.stdout + r.stderr)

    with ThreadPoolExecutor(max_workers=4) as ex:
        list(ex.map(run, jobs))
    open(fh_path, 'w').write(flags_hash() + '\n')
    if jobs or force or not os.path.exists(LIB) or not stamp_ok():
        run([hipcc, '--offload-arch=gfx950', '-shared', '-fPIC'] + objs + ['-o', LIB])
        # sources + flags + compiler | SHA-256 of the library | sources + flags alone (for a box where the compiler cannot be asked)
        open(STAMP, 'w').write(source_hash() + '\n' + lib_hash() + '\n' + source_hash(with_compiler=False) + '\n')
        # the decoder's instruction counts of THIS binary, for bench.py's VALU-issue roofline (best effort: needs llvm-objdump);
        # written next to the library, not into profiles/ (a build must not touch tracked evidence)
        tool = os.path.join(HERE, '..', 'tools', 'isa_mix.py')
        if os.path.exists(tool):
            subprocess.run([sys.executable, tool, LIB, 'ldpc_dec', '--json', ISA_JSON], capture_output=True)
    return LIB


def read_stamp(full=False):
    """(source hash, library SHA-256) recorded by the last link, or ('', ''); ``full``: also the compiler-less source hash."""
    lines = (open(STAMP).read().split() if os.path.exists(STAMP) else []) + ['', '', '']
    return (lines[0], lines[1], lines[2]) if full else (lines[0], lines[1])


def stamp_ok():
    """The library in the tree is the file the last link wrote (SHA-256) AND that link used exactly these sources, flags and
    compiler.  On a box where `hipcc --version` cannot be run the compiler part of the hash cannot be formed: the check then uses
    the stamp's third line, the hash of sources + flags alone (still together with the library's own SHA-256)."""
    if not (os.path.exists(LIB) and os.path.exists(STAMP)):
        return False
    src, so, src_nc = read_stamp(full=True)
    if so != lib_hash():
        return False
    if compiler_id() in ('', 'unknown'):
        return src_nc != '' and src_nc == source_hash(with_compiler=False)
    return src == source_hash()


if __name__ == '__main__':
    print(build(force='--force' in sys.argv))
