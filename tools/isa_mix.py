#!/usr/bin/env python3
"""Developer tool: instruction mix of the iteration loop of a decoder kernel in a compiled object / shared library.

    python tools/isa_mix.py <file.o|libnrx.so> <kernel-name-substring> [--json out.json]

--json writes, for every matching kernel, the register metadata and the loop's instruction counts by issue class (VALU / SALU /
LDS / waits) together with the SHA-256 of the inspected file: bench.py prices the decoder's VALU-issue bound from
neoradium_amd/libnrx.isa.json (written by the build next to the library; a copy per round under profiles/) and checks that hash
against the library it actually loaded.

Prints register / scratch metadata, the opcode histogram of the innermost long backward-branch loop, and the number of
back-to-back VOP2 v_cndmask_b32 pairs in it (each costs the issuing wave ~19 cycles on gfx950)."""
import collections
import hashlib
import json
import re
import subprocess
import sys
import tempfile

LLVM = '/opt/rocm/lib/llvm/bin/'


def code_objects(path):
    data = open(path, 'rb').read()
    idx = [m.start() for m in re.finditer(b'\x7fELF', data)]
    for n, i in enumerate(idx):
        end = idx[n + 1] if n + 1 < len(idx) else len(data)
        f = tempfile.NamedTemporaryFile(suffix='.elf', delete=False)
        f.write(data[i:end])
        f.close()
        yield f.name


def classify(op):
    if op.startswith('v_'):
        return 'valu'
    if op.startswith('ds_'):
        return 'lds'
    if op in ('s_waitcnt', 's_nop'):
        return 'wait'
    if op.startswith('s_load') or op.startswith('s_store') or op.startswith('s_buffer') or op in ('s_memtime', 's_memrealtime'):
        return 'smem'
    if op.startswith('s_'):
        return 'salu'
    return 'vmem'


def main():
    path, want = sys.argv[1], sys.argv[2]
    out_json = sys.argv[sys.argv.index('--json') + 1] if '--json' in sys.argv else None
    report = {'file': path, 'sha256': hashlib.sha256(open(path, 'rb').read()).hexdigest(), 'kernels': {}}
    for co in code_objects(path):
        sym = subprocess.run([LLVM + 'llvm-readelf', '-s', co], capture_output=True, text=True).stdout
        if want not in sym:
            continue
        notes = subprocess.run([LLVM + 'llvm-readelf', '--notes', co], capture_output=True, text=True).stdout
        dis = subprocess.run([LLVM + 'llvm-objdump', '-d', '--mcpu=gfx950', co], capture_output=True, text=True).stdout.split('\n')
        for k, l in enumerate(dis):
            m = re.match(r'^[0-9a-f]{16} <(.*)>:', l)
            if not m or want not in m.group(1) or m.group(1).endswith('.kd'):
                continue
            name = m.group(1)
            end = next((j for j in range(k + 1, len(dis)) if re.match(r'^[0-9a-f]{16} <', dis[j])), len(dis))
            ins = []
            for l2 in dis[k:end]:
                mm = re.match(r'\s+(\S+)\s+(.*?)\s+//\s+([0-9A-F]+):', l2)
                if mm:
                    ins.append((int(mm.group(3), 16), mm.group(1), mm.group(2)))
            at = {a: i for i, (a, _, _) in enumerate(ins)}
            loops = []
            for i, (a, op, args) in enumerate(ins):
                if op.startswith('s_cbranch') or op == 's_branch':
                    off = int(args.split()[0])
                    off -= 65536 if off >= 32768 else 0
                    if off < 0:      # (the iteration loop may be closed by an unconditional branch behind its exit test)
                        loops.append((at.get(a + 4 + 4 * off), i))
            loops = [lp for lp in loops if lp[0] is not None and lp[1] - lp[0] > 500]
            meta = re.search(re.escape(name) + r'.*?\.private_segment_fixed_size:\s+(\d+).*?\.sgpr_count:\s+(\d+).*?\.vgpr_count:\s+(\d+)',
                             notes, re.S)
            print(name[:110])
            if meta:
                print(f"  scratch {meta.group(1)} B, sgpr {meta.group(2)}, vgpr {meta.group(3)}")
            for lo, hi in loops[:1]:
                body = ins[lo:hi + 1]
                c = collections.Counter(op for _, op, _ in body)
                pairs = sum(1 for i in range(len(body) - 1) if body[i][1] == 'v_cndmask_b32_e32' and body[i + 1][1] == 'v_cndmask_b32_e32')
                scr = sum(n for op, n in c.items() if op.startswith('scratch_') or op.startswith('buffer_'))
                print(f"  loop of {len(body)} instructions; back-to-back VOP2 cndmask pairs {pairs}; scratch/buffer ops {scr}")
                print("  " + ", ".join(f"{op} {n}" for op, n in c.most_common(24)))
                cls = collections.Counter()
                for op, n in c.items():
                    cls[classify(op)] += n
                # further loops of the same size class behind the first (a second copy of the iteration loop: NRX_DEC3_SKIPZ)
                more, last_hi = [], hi
                for lo2, hi2 in loops[1:]:
                    if lo2 > last_hi and hi2 - lo2 > (hi - lo) // 2:
                        b2 = ins[lo2:hi2 + 1]
                        more.append({'loop_instructions': len(b2), 'valu': sum(1 for _, op, _ in b2 if classify(op) == 'valu')})
                        last_hi = hi2
                report['kernels'][name] = {'other_loops': more, 'scratch_bytes': int(meta.group(1)) if meta else None, 'sgpr': int(meta.group(2)) if meta else None,
                                           'vgpr': int(meta.group(3)) if meta else None, 'loop_instructions': len(body),
                                           'by_class': dict(cls), 'scratch_ops_in_loop': scr, 'by_opcode': dict(c.most_common())}
    if out_json:
        json.dump(report, open(out_json, 'w'), indent=1)


if __name__ == '__main__':
    main()
