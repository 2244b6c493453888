#!/usr/bin/env python3
"""Per-basic-block instruction table of one kernel from the compiler's ISA (hipcc -save-temps), as committed under profiles/.

    python3 tools/isa_sections.py <mangled-name fragment> [out.md]      e.g.  consistency_step_q32_kernelILi10ELi2E

Compiles csrc/dc_consistency.hip for gfx950 with -save-temps into a scratch directory, cuts the kernel out of the .s and
prints, per basic block of at least `--min` instructions, the number of VALU instructions by class (fp64 / fp32 / integer-and-
move), LDS, vector-memory and scalar instructions, plus the register / LDS footprint from the kernel descriptor."""
import argparse
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def cls(i):
    op = i.split()[0]
    if op.startswith('v_'):
        return 'valu_f64' if 'f64' in op else ('valu_f32' if 'f32' in op else 'valu_int')
    if op.startswith('s_'):
        return 'salu'
    if op.startswith('ds_'):
        return 'lds'
    return 'vmem'


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('kernel')
    ap.add_argument('out', nargs='?')
    ap.add_argument('--min', type=int, default=8)
    ap.add_argument('--source', default='dc_consistency.hip')
    args = ap.parse_args()
    tmp = tempfile.mkdtemp()
    src = os.path.join(ROOT, 'depth_correction_amd', 'csrc', args.source)
    subprocess.run(['/opt/rocm/bin/hipcc', '-O3', '-std=c++17', '-fPIC', '--offload-arch=gfx950', '-I' + os.path.join(ROOT, 'include'),
                    '-I' + os.path.dirname(src), '-save-temps', '-c', src, '-o', os.path.join(tmp, 'o.o')], check=True, cwd=tmp,
                   stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    asm = [f for f in os.listdir(tmp) if f.endswith('gfx950.s')][0]
    lines = open(os.path.join(tmp, asm)).read().split('\n')
    start = next(i for i, l in enumerate(lines) if re.match(r'^_Z\w*' + re.escape(args.kernel) + r'\w*:', l))
    end = next(i for i in range(start, len(lines)) if '.amdhsa_kernel' in lines[i])
    name = lines[start].split(':')[0]
    dstart = next(i for i in range(start, len(lines)) if re.match(r'\s*\.amdhsa_kernel\s+' + re.escape(name) + r'\s*$', lines[i]))
    desc = {}
    for l in lines[dstart:dstart + 60]:
        m = re.match(r'\s*\.amdhsa_(next_free_vgpr|next_free_sgpr|group_segment_fixed_size|private_segment_fixed_size)\s+(\d+)', l)
        if m:
            desc[m.group(1)] = int(m.group(2))
    blocks, cur = [], ['(entry)', []]
    for l in lines[start + 1:end]:
        t = l.strip()
        if not t or t.startswith(';'):
            continue
        if re.match(r'^\.LBB\d+_\d+:', t):
            blocks.append(cur)
            cur = [t.split(':')[0], []]
            continue
        if t.startswith('.') or re.match(r'^[A-Za-z_$][\w.$]*:', t):
            continue
        cur[1].append(t)
    blocks.append(cur)
    out = ['kernel `%s` (%s): VGPRs %s, SGPRs %s, static LDS %s B, scratch %s B' % (
        lines[start].split(':')[0], args.source, desc.get('next_free_vgpr'), desc.get('next_free_sgpr'),
        desc.get('group_segment_fixed_size'), desc.get('private_segment_fixed_size')), '',
        '| block | instructions | VALU fp64 | VALU fp32 | VALU int / move | LDS | vector memory | scalar |', '|---|---|---|---|---|---|---|---|']
    tot = {}
    for name, ins in blocks:
        c = {}
        for i in ins:
            c[cls(i)] = c.get(cls(i), 0) + 1
            tot[cls(i)] = tot.get(cls(i), 0) + 1
        if len(ins) >= args.min:
            out.append('| %s | %d | %d | %d | %d | %d | %d | %d |' % (name, len(ins), c.get('valu_f64', 0), c.get('valu_f32', 0), c.get('valu_int', 0),
                                                                     c.get('lds', 0), c.get('vmem', 0), c.get('salu', 0)))
    out.append('| (all blocks, static) | %d | %d | %d | %d | %d | %d | %d |' % (sum(tot.values()), tot.get('valu_f64', 0), tot.get('valu_f32', 0),
                                                                               tot.get('valu_int', 0), tot.get('lds', 0), tot.get('vmem', 0), tot.get('salu', 0)))
    text = '\n'.join(out)
    if args.out:
        with open(args.out, 'a') as f:
            f.write(text + '\n')
    print(text)


if __name__ == '__main__':
    main()
