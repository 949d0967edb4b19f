#!/usr/bin/env python3
"""Per-kernel resource report and build-time ISA checks over hipcc -save-temps .s files (gfx950).

    isa_report.py <file.s> ...            register / LDS / scratch usage per kernel
    isa_report.py --check <file.s> ...    the checks below; exit status 1 and one line per finding if any fails

Checks (run by the Makefile on every build):

 1. wide-store data hazard.  CDNA ISA, "manually inserted wait states": a VMEM store of more than 64 bits followed by a VALU
    write of the VGPRs that hold its data needs one wait state — "unless the store uses an SGPR for soffset".  LLVM's
    GCNHazardRecognizer::createsVALUHazard encodes that exemption, so hipcc pads the sequence only for soffset = 0 / off.
    dwconv_roll.h saw stale dwords with exactly the exempted form (buffer_store_dwordx4, SGPR soffset, data registers
    rewritten right behind it; profiles/micro/store_hazard.hip reproduces the sequence in isolation), so this build does
    not rely on the exemption: any buffer_store_dwordx3 / x4 with an SGPR soffset whose data registers are written by a
    VALU instruction within the next two instructions is an error.  (Kernels avoid the form: 8-byte stores, or all data
    registers finished before the first store.)

 2. registers with a load in flight behind the compiler's back.  Inline-asm ds_read_* / global_load_* results are not
    tracked by SIInsertWaitcnts; correctness rests on no instruction touching the destination registers until the
    inline-asm s_waitcnt that retires them (lgkmcnt for ds_read, vmcnt for global_load).  A register-allocator copy,
    spill or any other use in between would read a value that has not landed: flagged as an error.  Loop bodies are scanned a
    second time with the loads that are in flight at their back-edge.

 3. the register cliff.  Several kernels sit at their register budget by design (dwconv_mfma.h: 42 tap operands + 56
    accumulators + 12 gathered operands at the 128-register budget of a 16-wave workgroup; pw2f_kernel: 192 accumulators at
    256): a compiler bump or an innocent edit can push one over, and what comes out is a kernel that is still correct and
    several times slower (the two dw variants of DESIGN.md section 4.0 item 9 compiled to 220+ spilled registers).  Every
    kernel's scratch (.private_segment_fixed_size) must stay within SCRATCH_BUDGET: 0 bytes unless listed, and the listed
    ones are the few dwords hipcc 7.2 spills outside the steady-state loops today.  Raising an entry is a decision, not a
    side effect.
"""
import re
import subprocess
import sys


def demangle(n):
    for tool in ("/opt/rocm/lib/llvm/bin/llvm-cxxfilt", "c++filt"):
        try:
            out = subprocess.run([tool, n], capture_output=True, text=True).stdout.strip()
            if out:
                return out
        except Exception:
            pass
    return n


# bytes of scratch per lane a kernel may use: (regex on the MANGLED name — the image has no demangler that knows DF16_ —,
# bytes); first match wins, default 0
SCRATCH_BUDGET = [
    (r"pw2f_kernelI\w+?Li384ELi384ELi3ELi8EE", 28),     # 6 dwords outside the steady-state loop
    (r"xs_mlp_kernelI\w+?Li192ELb[01]EE", 20),          # 4 dwords in the pass prologue
    (r"dwconv7_ln_roll_kernelI\w+?Li96ELi4EE", 8),      # 1 dword
    (r"dwconv7_ln_kernelI\w+?Li768EE", 88),             # generic kernel of the 3x3 maps of the 112-pixel pass
]


def kernel_meta(path):
    """[(mangled name, {vgpr, agpr, sgpr, spill, scratch, lds})] from the .amdgpu_metadata block of one .s file"""
    out = []
    s = open(path).read()
    for b in re.split(r'^\s+- \.agpr_count:', s, flags=re.M)[1:]:
        g = lambda k: int(re.search(r'\.%s:\s+(\S+)' % k, b).group(1))
        out.append((re.search(r'\.name:\s+(\S+)', b).group(1),
                    dict(agpr=int(b.split()[0]), vgpr=g('vgpr_count'), sgpr=g('sgpr_count'), spill=g('vgpr_spill_count'),
                         scratch=g('private_segment_fixed_size'), lds=g('group_segment_fixed_size'))))
    return out


def report(paths):
    for path in paths:
        for name, m in kernel_meta(path):
            nm = re.sub(r'\(.*', '', demangle(name))[:110]
            print(f"{nm:110s} vgpr {m['vgpr']:>4} agpr {m['agpr']:>4} sgpr {m['sgpr']:>4} spill {m['spill']:>3} "
                  f"scratch {m['scratch']:>4} lds {m['lds']}")


def check_scratch(path):
    findings = []
    for name, m in kernel_meta(path):
        nm = re.sub(r'\(.*', '', demangle(name))
        budget = next((b for pat, b in SCRATCH_BUDGET if re.search(pat, name)), 0)
        if m['scratch'] > budget:
            findings.append(f"{path}: {nm[:100]}: {m['scratch']} bytes of scratch per lane ({m['spill']} spilled VGPRs), budget "
                            f"{budget} — register cliff (isa_report.py check 3)")
    return findings


REG = re.compile(r'\b([va])(\d+)\b|\b([va])\[(\d+):(\d+)\]')


def regs_of(operand):
    """set of ('v'|'a', n) named by one operand string"""
    out = set()
    for m in REG.finditer(operand):
        if m.group(1):
            out.add((m.group(1), int(m.group(2))))
        else:
            out.update((m.group(3), i) for i in range(int(m.group(4)), int(m.group(5)) + 1))
    return out


def split_ops(rest):
    return [o.strip() for o in rest.split(',')] if rest else []


def is_valu(op):
    return op.startswith('v_') and not op.startswith('v_mfma') and not op.startswith('v_smfma')


def kernels(text):
    """yield (kernel symbol, [(line number, instruction text, in_inline_asm)])"""
    name, body, in_asm = None, [], False
    for ln, raw in enumerate(text.split('\n'), 1):
        line = raw.split(';')[0].rstrip() if not raw.lstrip().startswith(';;#') else raw.strip()
        m = re.match(r'^(_Z\w+):', raw)
        if m:
            name, body, in_asm = m.group(1), [], False
            continue
        if name is None:
            continue
        if raw.lstrip().startswith(';;#ASMSTART'):
            in_asm = True
            continue
        if raw.lstrip().startswith(';;#ASMEND'):
            in_asm = False
            continue
        t = line.strip()
        if not t or t.startswith('.') or t.endswith(':'):
            if t.startswith('.Lfunc_end'):
                yield name, body
                name = None
            elif re.match(r'^\.LBB\w+:$', t):
                body.append((ln, t, False))               # a label: check 2 follows backward branches to it
            continue
        body.append((ln, t, in_asm))
        if t.startswith('s_endpgm'):
            pass


def check_file(path):
    findings = []
    text = open(path).read()
    for kname, body in kernels(text):
        short = re.sub(r'\(.*', '', demangle(kname))[:90]
        # ---- check 1
        for i, (ln, t, _) in enumerate(body):
            m = re.match(r'buffer_store_dwordx[34]\s+(.*)', t)
            if not m:
                continue
            ops = split_ops(m.group(1))
            if len(ops) < 4:
                continue
            data = regs_of(ops[0])
            soff = ops[3].split()[0]
            if not re.match(r'^s\d+$|^s\[\d+', soff):
                continue                                  # soffset 0 / off / literal: hipcc's hazard recogniser covers it
            for ln2, t2, _ in body[i + 1:i + 3]:
                op2 = t2.split()[0]
                if not is_valu(op2):
                    continue
                dst = split_ops(t2[len(op2):])[:1]
                if dst and regs_of(dst[0]) & data:
                    findings.append(f"{path}:{ln}: {short}: `{t}` (SGPR soffset) has its data registers written by "
                                    f"`{t2}` {ln2 - ln} line(s) later — wide-store hazard (isa_report.py check 1)")
        # ---- check 2: a linear scan, plus one more pass over every loop body with the pending set its back-edge carries
        # (a load issued at the end of a loop body and retired at its head is otherwise only checked through the copy of
        # the same load in front of the loop)
        labels = {t[:-1]: i for i, (_, t, _) in enumerate(body) if t.endswith(':')}

        def scan(lo, hi, pending, carried):
            for idx in range(lo, hi):
                ln, t, in_asm = body[idx]
                if t.endswith(':'):
                    continue
                op = t.split()[0]
                rest = t[len(op):]
                ops = split_ops(rest)
                if in_asm and (op.startswith('ds_read') or op.startswith('global_load_dword') or op.startswith('buffer_load_dword')) \
                        and 'lds' not in t.split():
                    kind = 'lgkm' if op.startswith('ds_read') else 'vm'
                    for r in regs_of(ops[0]) if ops else ():
                        pending[r] = (kind, ln)
                    continue
                if in_asm and op == 's_waitcnt':
                    # a counted vmcnt(N) wait in this code base retires everything older than the N youngest operations; the
                    # inline-asm loads it guards are always older (see the kernels), so any asm vmcnt / lgkmcnt wait clears
                    if 'lgkmcnt' in t:
                        pending = {r: v for r, v in pending.items() if v[0] != 'lgkm'}
                    if 'vmcnt' in t:
                        pending = {r: v for r, v in pending.items() if v[0] != 'vm'}
                    continue
                if not carried and pending and op.startswith('s_cbranch') or (not carried and pending and op == 's_branch'):
                    tgt = ops[0] if ops else ''
                    if tgt in labels and labels[tgt] < idx:          # back-edge: the loop body again, with what is in flight now
                        scan(labels[tgt], idx, dict(pending), True)
                if not pending or in_asm:
                    continue
                used = set()
                for o in ops:
                    used |= regs_of(o)
                hit = used & set(pending)
                if hit:
                    r = sorted(hit)[0]
                    how = " (carried over the loop's back-edge)" if carried else ""
                    findings.append(f"{path}:{ln}: {short}: `{t}` touches {r[0]}{r[1]} while the inline-asm load of line "
                                    f"{pending[r][1]} is still in flight{how} (no asm s_waitcnt between) — isa_report.py check 2")
                    for h in hit:
                        pending.pop(h, None)
            return pending

        scan(0, len(body), {}, False)
    return findings


if __name__ == "__main__":
    args = sys.argv[1:]
    if args and args[0] == "--check":
        bad = []
        for p in args[1:]:
            bad += check_file(p)
            bad += check_scratch(p)
        for b in bad:
            print(b)
        print(f"isa_report.py --check: {len(args) - 1} file(s), {len(bad)} finding(s)")
        sys.exit(1 if bad else 0)
    report(args)
