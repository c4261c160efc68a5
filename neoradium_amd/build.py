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
FLAGS = ['-std=c++20', '-O3', '--offload-arch=gfx950', '-ffp-contract=off', '-fno-slp-vectorize', '-mllvm', '-amdgpu-sdwa-peephole=0', '-fPIC', '-Wno-comment',
         '-Wno-unused-value']


def _newer(src, dst, deps):
    if not os.path.exists(dst):
        return True
    t = os.path.getmtime(dst)
    return any(os.path.getmtime(d) > t for d in [src] + deps)


def source_hash():
    """SHA-256 over the compile flags and every source / header the library is built from."""
    h = hashlib.sha256(' '.join(FLAGS).encode())
    for f in sorted(glob.glob(os.path.join(CSRC, '*.hip')) + glob.glob(os.path.join(CSRC, '*.h'))
                    + glob.glob(os.path.join(HERE, '..', 'include', '*.h'))):
        h.update(os.path.basename(f).encode())
        h.update(open(f, 'rb').read())
    return h.hexdigest()


def build(force=False, verbose=True):
    """Compile what is out of date (by modification time) and link.  `force` (or NRX_FORCE_BUILD=1 in the environment) recompiles
    everything.  neoradium_amd/libnrx.stamp records the hash of the sources + flags the library was linked from: `stamp_ok()`
    tells whether the library in the tree is the one these sources produce (a shipped binary can be checked against it)."""
    force = force or os.environ.get('NRX_FORCE_BUILD', '') not in ('', '0')
    hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
    os.makedirs(OBJ, exist_ok=True)
    srcs = sorted(glob.glob(os.path.join(CSRC, '*.hip')))
    deps = glob.glob(os.path.join(CSRC, '*.h')) + glob.glob(os.path.join(HERE, '..', 'include', '*.h'))
    if not force and stamp_ok():
        return LIB          # the library in the tree was linked from exactly these sources and flags (objects may be absent: GPU box)
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
    if jobs or force or not os.path.exists(LIB) or not stamp_ok():
        run([hipcc, '--offload-arch=gfx950', '-shared', '-fPIC'] + objs + ['-o', LIB])
        open(STAMP, 'w').write(source_hash() + '\n')
        # the decoder's instruction counts of THIS binary, for bench.py's VALU-issue roofline (best effort: needs llvm-objdump)
        tool = os.path.join(HERE, '..', 'tools', 'isa_mix.py')
        out = os.path.join(HERE, '..', 'profiles', 'r3_decoder_isa.json')
        if os.path.exists(tool) and os.path.isdir(os.path.dirname(out)):
            subprocess.run([sys.executable, tool, LIB, 'ldpc_dec', '--json', out], capture_output=True)
    return LIB


def stamp_ok():
    return os.path.exists(LIB) and os.path.exists(STAMP) and open(STAMP).read().strip() == source_hash()


if __name__ == '__main__':
    print(build(force='--force' in sys.argv))
