"""Scan the gfx950 code of libciaosr_hip.so for packed-fp32 VALU instructions with a CROSSED half selection.

Why (tools/ubench/pk_mfma_corun.hip): on gfx950 a v_pk_mul_f32 / v_pk_fma_f32 / v_pk_add_f32 whose operand pair is read
crossed -- op_sel:[0,1] op_sel_hi:[1,0] and the like: the low result lane takes the HIGH register of the pair, the high lane the LOW
one -- returns wrong values in lanes 48-63 while another wave of the same SIMD issues 16-bit MFMAs.  hipcc's SLP vectoriser emits that
form; the library is built with -fno-slp-vectorize, and this scan (a CPU test) holds the shipped binary to "no such instruction".

    python tools/isa_scan.py [path/to/lib.so]   ->   prints offending instructions, exit code = their number (capped at 255)
"""
import os
import re
import subprocess
import sys
import tempfile

LLVM = '/opt/rocm/lib/llvm/bin'
MAGIC = b'__CLANG_OFFLOAD_BUNDLE__'
PK = re.compile(r'\b(v_pk_(?:mul|fma|add)_f32)\b(.*)')


def crossed(ops):
    """True if some source operand reads (hi, lo): op_sel bit 1 and op_sel_hi bit 0 for that operand (defaults: op_sel 0, op_sel_hi 1)."""
    sel = re.search(r'op_sel:\[([01,]+)\]', ops)
    sel_hi = re.search(r'op_sel_hi:\[([01,]+)\]', ops)
    lo = [int(v) for v in sel.group(1).split(',')] if sel else []
    hi = [int(v) for v in sel_hi.group(1).split(',')] if sel_hi else []
    n = max(len(lo), len(hi), 2)
    lo += [0] * (n - len(lo))
    hi += [1] * (n - len(hi))
    return any(a == 1 and b == 0 for a, b in zip(lo, hi))


def device_objects(lib, workdir):
    fat = os.path.join(workdir, 'fatbin.bin')
    subprocess.run([os.path.join(LLVM, 'llvm-objcopy'), '--dump-section', f'.hip_fatbin={fat}', lib, os.path.join(workdir, 'copy.so')], check=True)
    blob = open(fat, 'rb').read()
    starts = [m.start() for m in re.finditer(re.escape(MAGIC), blob)]
    for i, s0 in enumerate(starts):
        piece = os.path.join(workdir, f'bundle{i}.bin')
        open(piece, 'wb').write(blob[s0:starts[i + 1] if i + 1 < len(starts) else len(blob)])
        out = os.path.join(workdir, f'dev{i}.co')
        r = subprocess.run([os.path.join(LLVM, 'clang-offload-bundler'), '--unbundle', '--type=o', f'--input={piece}',
                            '--targets=hipv4-amdgcn-amd-amdhsa--gfx950', f'--output={out}'], capture_output=True, text=True)
        if r.returncode == 0 and os.path.exists(out) and os.path.getsize(out) > 0:
            yield out


def scan(lib):
    hits, n_pk, n_obj = [], 0, 0
    with tempfile.TemporaryDirectory() as wd:
        for co in device_objects(lib, wd):
            n_obj += 1
            dis = subprocess.run([os.path.join(LLVM, 'llvm-objdump'), '-d', '--no-show-raw-insn', co], capture_output=True, text=True).stdout
            func = '?'
            for line in dis.splitlines():
                m = re.match(r'^[0-9a-f]+ <(.+)>:', line)
                if m:
                    func = m.group(1)
                    continue
                m = PK.search(line)
                if m:
                    n_pk += 1
                    if crossed(m.group(2)):
                        hits.append((func, line.strip()))
    return hits, n_pk, n_obj


if __name__ == '__main__':
    here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(here, 'ciaosr_amd', 'csrc', 'libciaosr_hip.so')
    hits, n_pk, n_obj = scan(lib)
    print(f'{lib}: {n_obj} device code objects, {n_pk} packed fp32 mul/fma/add instructions, {len(hits)} with a crossed half selection')
    for func, line in hits[:40]:
        print(f'  {func}: {line}')
    sys.exit(min(len(hits), 255))
