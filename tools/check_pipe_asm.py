#!/usr/bin/env python3
"""Static check of the hand-scheduled kernels (wave_pipe.hpp) in a gfx950 assembly file:
no instruction may touch a register whose asm-issued load is still in flight, i.e. between the
load (inside an ASMSTART/ASMEND block) and the `; pin` statement that releases it after a
counted wait.  A forward "may be in flight" dataflow over the kernel's control-flow graph (basic
blocks from the labels and branches of the assembly text), so loop back-edges and the branches
around the border-aware sampler are followed.

A kernel that carries such loops must not spill vector registers either, nor contain flat / scratch
accesses (they count in vmcnt) (exit 1).

The LDS form of the same construct (tile_warp.hpp, ring_remap.hpp: `ds_read_b64` / `ds_read2_b32`
issued from asm statements the compiler does not see as memory accesses, released by
`s_waitcnt lgkmcnt(N)`): LDS operations of a wave return in order and count in lgkmcnt together
with scalar memory reads, which return out of order.  Every kernel with such reads is walked block
by block with the queue of its outstanding lgkm operations: a wait with N > 0 retires all but the
N youngest only while no scalar read is outstanding; no instruction may touch the destination of
an asm-issued read that is still in the queue; none may be outstanding at a block boundary; and
the kernel must not use scratch memory (`.private_segment_fixed_size` 0, no spilled vector
register): a spill the compiler puts between a read group and its counted wait would be invisible
here and in the source alike.

    tools/check_pipe_asm.py file.s [name filter ...]       exit 1 on a violation
"""
import re
import sys

REG = re.compile(r'\bv(\d+)\b|\bv\[(\d+):(\d+)\]')


def regs_of(text):
    out = set()
    for m in REG.finditer(text):
        if m.group(1) is not None:
            out.add(int(m.group(1)))
        else:
            out.update(range(int(m.group(2)), int(m.group(3)) + 1))
    return out


def check(name, lines):
    """forward may-be-in-flight dataflow over the kernel's control-flow graph"""
    # --- instructions: (line no, kind, text, regs) with kind in load/pin/use/label/branch/end
    ins, in_asm = [], False
    cur_loop, parent, pending_label = None, {}, None   # innermost loop header of the current block
    loop_of = []
    asm_first = None   # first instruction of the current asm block
    unguarded = []
    for ln, l in lines:
        t = l.strip()
        # every asm block that carries a vector-memory instruction opens with `s_nop 4`: the
        # VALU-writes-SGPR -> VMEM-reads-SGPR hazard (5 wait states) is invisible to the compiler
        if t.startswith(';;#ASMSTART'):
            asm_first = None
        elif in_asm and t and not t.startswith(';'):
            if asm_first is None:
                asm_first = t
            if t.split()[0].startswith(('buffer_', 'global_')) and asm_first.split(';')[0].strip() != 's_nop 4':
                unguarded.append((ln, t))
        # LLVM's block comments: "in Loop: Header=BB0_729 Depth=1", "=>This Loop Header: Depth=1",
        # "Parent Loop BB0_729 Depth=1" + "=> This Inner Loop Header: Depth=2"
        if re.match(r'^(\.LBB\d+_\d+):|^; %bb\.\d+:', t):
            cur_loop = None
            pending_label = re.match(r'^\.L(BB\d+_\d+):', t).group(1) if t.startswith('.') else None
        mm = re.search(r'in Loop: Header=(BB\d+_\d+) Depth=(\d+)', t)
        if mm:
            cur_loop = mm.group(1)
        if 'Loop Header: Depth=' in t and pending_label:
            cur_loop = pending_label
        mm = re.search(r'Parent Loop (BB\d+_\d+) Depth=(\d+)', t)
        if mm and pending_label:
            parent.setdefault(pending_label, mm.group(1))
        if t.startswith(';;#ASMSTART'):
            in_asm = True
            continue
        if t.startswith(';;#ASMEND'):
            in_asm = False
            continue
        m = re.match(r'^(\.LBB\d+_\d+):', t)
        if m:
            ins.append((ln, 'label', m.group(1), None)); loop_of.append(None)
            continue
        if in_asm and t.startswith('; pin'):
            ins.append((ln, 'pin', t, regs_of(t))); loop_of.append(cur_loop)
            continue
        if not t or t.startswith((';', '.')):
            continue
        code = t.split(';')[0].strip()
        op = code.split()[0]
        # flat and scratch accesses count in vmcnt too (and complete out of order with it):
        # neither may appear in a kernel whose waits are counted by hand
        if op.startswith(('flat_', 'scratch_')):
            unguarded.append((ln, 'counts in vmcnt behind the hand-counted waits: ' + code))
        if in_asm and op.startswith(('buffer_load', 'global_load')):
            dst = code.split()[1].rstrip(',')
            ins.append((ln, 'load', code, (regs_of(dst), regs_of(' '.join(code.split()[2:]))))); loop_of.append(cur_loop)
        elif op == 's_endpgm':
            ins.append((ln, 'end', code, None)); loop_of.append(cur_loop)
        elif op == 's_branch' or op.startswith('s_cbranch'):
            ins.append((ln, 'branch', code, (op == 's_branch', code.split()[-1]))); loop_of.append(cur_loop)
        else:
            ins.append((ln, 'use', code, regs_of(code))); loop_of.append(cur_loop)
    # --- basic blocks
    starts = {0}
    for i, (ln, k, t, r) in enumerate(ins):
        if k == 'label':
            starts.add(i)
        if k in ('branch', 'end') and i + 1 < len(ins):
            starts.add(i + 1)
    starts = sorted(starts)
    block_of = {}
    for bi, st in enumerate(starts):
        en = starts[bi + 1] if bi + 1 < len(starts) else len(ins)
        for i in range(st, en):
            block_of[i] = bi
    label_block = {ins[st][2]: bi for bi, st in enumerate(starts) if ins[st][1] == 'label'}
    nb = len(starts)

    def succ(bi):
        st = starts[bi]
        en = starts[bi + 1] if bi + 1 < nb else len(ins)
        last = ins[en - 1]
        out = []
        if last[1] == 'end':
            return out
        if last[1] == 'branch':
            uncond, tgt = last[3]
            if tgt in label_block:
                out.append(label_block[tgt])
            if not uncond and bi + 1 < nb:
                out.append(bi + 1)
            return out
        if bi + 1 < nb:
            out.append(bi + 1)
        return out

    def transfer(bi, inset, report):
        cur = dict(inset)
        st = starts[bi]
        en = starts[bi + 1] if bi + 1 < nb else len(ins)
        for i in range(st, en):
            ln, k, t, r = ins[i]
            if k == 'load':
                dst, addr = r
                for x in addr:
                    if x in cur and report is not None:
                        report.append((ln, t, x, cur[x]))
                for x in dst:
                    cur[x] = ln
            elif k == 'pin':
                for x in r:
                    cur.pop(x, None)
            elif k == 'use':
                if report is not None:
                    for x in r:
                        if x in cur:
                            report.append((ln, t, x, cur[x]))
        return cur

    def outer(h):
        seen_h = set()
        while h in parent and h not in seen_h:
            seen_h.add(h)
            h = parent[h]
        return h

    def block_loop(bi):
        st = starts[bi]
        en = starts[bi + 1] if bi + 1 < nb else len(ins)
        for i in range(st, en):
            if loop_of[i] is not None:
                return outer(loop_of[i])
            if ins[i][1] == 'label' and i + 1 < en and loop_of[i + 1] is not None:
                return outer(loop_of[i + 1])
        return None

    bloop = [block_loop(bi) for bi in range(nb)]
    inn = [dict() for _ in range(nb)]
    work = [0]
    seen = {0}
    while work:
        bi = work.pop()
        out = transfer(bi, inn[bi], None)
        for sb in succ(bi):
            # leaving the strip loop: the wave is done with its strip, nothing is used afterwards
            # (the uniform if/else of the two coordinate rules is laid out as loop A -> Flow -> loop B)
            if bloop[bi] is not None and bloop[sb] != bloop[bi]:
                continue
            merged = dict(inn[sb])
            changed = sb not in seen
            for x, ln in out.items():
                if x not in merged:
                    merged[x] = ln
                    changed = True
            if changed:
                inn[sb] = merged
                seen.add(sb)
                work.append(sb)
    bad = []
    for bi in sorted(seen):
        transfer(bi, inn[bi], bad)
    nload = sum(1 for x in ins if x[1] == 'load')
    npin = sum(1 for x in ins if x[1] == 'pin')
    for ln, t in unguarded:
        bad.append((ln, t if t.startswith('counts in') else
                    'vector-memory asm statement without the s_nop 4 hazard guard: ' + t, -1, ln))
    return bad, nload, npin


LGKM = re.compile(r'lgkmcnt\((\d+)\)')


def check_lds(name, lines):
    """the asm-issued LDS reads of a kernel against its lgkmcnt waits (see the module docstring)"""
    queue = []      # outstanding lgkm operations, oldest first: (kind, destination registers, line)
    bad = []
    in_asm = False
    nread = nwait = 0
    for ln, l in lines:
        t = l.strip()
        if t.startswith(';;#ASMSTART'):
            in_asm = True
            continue
        if t.startswith(';;#ASMEND'):
            in_asm = False
            continue
        if re.match(r'^(\.LBB\d+_\d+):', t):
            for k, d, src in queue:
                if k == 'asm':
                    bad.append((ln, 'block boundary (label) with an asm-issued LDS read in flight', min(d), src))
            queue = []
            continue
        if not t or t.startswith((';', '.')):
            continue
        code = t.split(';')[0].strip()
        op = code.split()[0]
        regs = regs_of(code)
        for k, d, src in queue:
            if k == 'asm' and regs & d:
                bad.append((ln, code, min(regs & d), src))
        if op == 's_waitcnt':
            m = LGKM.search(code)
            if m:
                n = int(m.group(1))
                nwait += 1 if in_asm else 0
                if n == 0:
                    queue = []
                elif not any(k == 'smem' for k, _, _ in queue):
                    queue = queue[len(queue) - n:] if n < len(queue) else queue
        elif op.startswith('ds_'):
            is_read = op.startswith(('ds_read', 'ds_bpermute', 'ds_permute', 'ds_swizzle'))
            dst = regs_of(code.split()[1].rstrip(',')) if is_read and len(code.split()) > 1 else set()
            if in_asm and is_read:
                nread += 1
                queue.append(('asm', dst, ln))
            else:
                queue.append(('lds', set(), ln))
        elif op.startswith(('s_load', 's_buffer_load', 's_memtime', 's_memrealtime', 's_dcache', 's_store')):
            queue.append(('smem', set(), ln))
        elif op == 's_endpgm' or op == 's_branch' or op.startswith('s_cbranch') or op == 's_barrier':
            for k, d, src in queue:
                if k == 'asm' and op != 's_barrier':
                    bad.append((ln, 'block boundary (%s) with an asm-issued LDS read in flight' % op, min(d), src))
            if op != 's_barrier':
                queue = []
    return bad, nread, nwait


def private_segments(text):
    """kernel name -> .private_segment_fixed_size of the code object metadata"""
    out = {}
    for blk in text.split('  - .agpr_count:')[1:]:
        n = re.search(r'\.name:\s+(\S+)', blk)
        v = re.search(r'\.private_segment_fixed_size:\s+(\d+)', blk)
        if n and v:
            out[n.group(1)] = int(v.group(1))
    return out


def spills(text):
    """kernel name -> .vgpr_spill_count of the code object metadata"""
    out = {}
    for blk in text.split('  - .agpr_count:')[1:]:
        n = re.search(r'\.name:\s+(\S+)', blk)
        v = re.search(r'\.vgpr_spill_count:\s+(\d+)', blk)
        if n and v:
            out[n.group(1)] = int(v.group(1))
    return out


def main():
    text = open(sys.argv[1]).read()
    spill = spills(text)
    private = private_segments(text)
    s = text.split('\n')
    flts = sys.argv[2:]
    rc = 0
    i = 0
    while i < len(s):
        l = s[i]
        if l.startswith('_Z') and l.split(';')[0].rstrip().endswith(':'):
            name = l.split(':')[0]
            j = i
            while 's_endpgm' not in s[j]:
                j += 1
            if not flts or any(f in name for f in flts):
                body = list(enumerate(s[i:j], i + 1))
                if any('; pin' in x for _, x in body):
                    bad, nload, npin = check(name, body)
                    nsp = spill.get(name, 0)
                    print('%s: %d asm loads, %d pins, %d violations, %d spilled VGPRs'
                          % (name[:100], nload, npin, len(bad), nsp))
                    # a kernel on the hand-scheduled loops must not spill vector registers: the
                    # 7x7 kernel capped at 128 registers (7 spills) gave wrong first rows of
                    # strips on the GPU although no statement here flags it (round 3)
                    rc |= 1 if nsp else 0
                    for ln, t, r, src in bad[:20]:
                        print('   line %d: v%d (loaded at line %d) may still be in flight: %s'
                              % (ln, r, src, t))
                    rc |= 1 if bad else 0
                in_asm_ds = False
                flag = False
                for _, x in body:
                    xs = x.strip()
                    if xs.startswith(';;#ASMSTART'):
                        flag = True
                    elif xs.startswith(';;#ASMEND'):
                        flag = False
                    elif flag and xs.startswith('ds_read'):
                        in_asm_ds = True
                        break
                if in_asm_ds:
                    bad, nread, nwait = check_lds(name, body)
                    nsp, prv = spill.get(name, 0), private.get(name, 0)
                    print('%s: %d asm LDS reads, %d asm lgkmcnt waits, %d violations, %d spilled VGPRs, '
                          '%d bytes of scratch per lane' % (name[:100], nread, nwait, len(bad), nsp, prv))
                    for ln, t, r, src in bad[:20]:
                        print('   line %d: v%d (LDS read issued at line %d) may still be in flight: %s'
                              % (ln, r, src, t))
                    rc |= 1 if (bad or nsp or prv) else 0
            i = j
        i += 1
    sys.exit(rc)


if __name__ == '__main__':
    main()
