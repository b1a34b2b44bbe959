#!/usr/bin/env python3
"""Static checks on the gfx950 machine code of lib/libqarig_hip.so (no GPU needed).

Several kernels issue MFMAs, LDS reads, buffer loads and LDS-DMAs from inline asm, where hipcc
neither pads hazards nor counts memory operations (csrc/ring_common.h, bmu.hip, conv.hip, gemm_lp.hip).
Two such slips were met on the GPU in round 2 ("wrong indices in some builds", a memory fault at
Cin = 256); this pass makes the invariants a property of the BUILD.  It disassembles every code
object of the library (llvm-objdump) and walks each kernel:

  R1  a VALU write of a VGPR that an MFMA reads as A / B / C needs 2 wait states in between;
  R2  the D registers of an MFMA may not be read or written by anything but an MFMA taking exactly
      that range as C until the MFMA's passes have drained (4 / 6 / 10 / 18 wait states for
      2 / 4 / 8 / 16-pass fp32 MFMAs, one more for the bf16 / fp8 ones);
  R3  the destination VGPRs of a vector-memory or LDS load may not be read or overwritten before an
      s_waitcnt has retired that load (vmcnt / lgkmcnt modelled as the in-order queues they are);
  R4  in a kernel that sets M0 by hand for an LDS-DMA (s_mov_b32 m0 / s_nop 0 / load ... lds), every
      LDS-DMA has its own M0 write right in front of it (no mixing with compiler-managed M0).

Wait states: one per instruction issued, N + 1 for `s_nop N`.  Control flow: hazards (R1, R2) are
followed through every branch edge for as long as a window is open; the wait-count model (R3) runs
linearly through the kernel and once more around every loop (backward branch) with the state at
the branch.  Findings are returned as text lines; tests/test_isa_lint.py requires none.
"""
import os
import re
import subprocess
import sys
import tempfile

OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"

_REG = re.compile(r"\b([va])(?:(\d+)|\[(\d+):(\d+)\])")
_LABEL = re.compile(r"^<(L\d+)>:")
_FUNC = re.compile(r"^[0-9a-f]+ <(.+)>:$")

# passes of an MFMA from its shape and input type (gfx950): cycles = flop / (flop per cycle per SIMD
# of the type's dense peak), a pass = 4 cycles
_MFMA_NAME = re.compile(r"v_mfma(?:_scale)?_f32_(\d+)x(\d+)x(\d+)(?:_(\d+)b)?_?(\w+)$")


def _mfma_passes(mn):
    m = _MFMA_NAME.match(mn)
    if not m:
        raise ValueError(f"isa_lint: unknown MFMA {mn}: teach _mfma_passes its shape")
    M, N, K, blocks, typ = int(m.group(1)), int(m.group(2)), int(m.group(3)), int(m.group(4) or 1), m.group(5)
    flop = 2 * M * N * K * blocks
    if typ == "f32":
        rate = 64               # 157.3 TF dense
    elif typ in ("bf16", "f16"):
        rate = 1024             # 2.5 PF
    elif "f8" in typ or "bf8" in typ or typ in ("i8", "f8f6f4"):
        rate = 2048             # 5 PF
    else:
        raise ValueError(f"isa_lint: unknown MFMA input type in {mn}")
    cycles = max(8, flop // rate)
    return {8: 2, 16: 4, 32: 8, 64: 16}.get(cycles, 16), typ == "f32"


_NEED_F32 = {2: 4, 4: 6, 8: 10, 16: 18}
_NEED_XDL = {2: 5, 4: 7, 8: 11, 16: 19}


def _regs(text):
    out = []
    for m in _REG.finditer(text):
        kind = m.group(1)
        if m.group(2) is not None:
            out.append((kind, int(m.group(2))))
        else:
            out += [(kind, r) for r in range(int(m.group(3)), int(m.group(4)) + 1)]
    return out


class Ins:
    __slots__ = ("idx", "mn", "ops", "text", "defs", "uses", "ws", "mfma", "need", "srcc", "target",
                 "vm", "lgkm", "is_valu", "addr")

    def __init__(self, idx, text, addr):
        self.idx, self.text, self.addr = idx, text, addr
        parts = text.split(None, 1)
        self.mn = parts[0]
        ops = parts[1] if len(parts) > 1 else ""
        self.ops = [o.strip() for o in ops.split(",")] if ops else []
        mn = self.mn
        self.ws = 1
        if mn == "s_nop":
            self.ws = int(self.ops[0], 0) + 1
        self.target = None
        if mn.startswith("s_cbranch") or mn == "s_branch":
            self.target = self.ops[0] if self.ops else None
        self.mfma = mn.startswith("v_mfma") or mn.startswith("v_smfma")
        self.need = 0
        self.srcc = ()
        self.vm = self.lgkm = False
        self.is_valu = mn.startswith("v_") and not self.mfma and mn != "v_nop"
        first_is_dst = False
        if mn.startswith("v_"):
            first_is_dst = mn not in ("v_nop",)
        elif mn.startswith(("global_load", "buffer_load", "flat_load", "scratch_load", "tbuffer_load")):
            self.vm = True
            lds_form = "_lds_" in mn or (self.ops and self.ops[-1].split()[-1] == "lds")
            first_is_dst = not lds_form
            if mn.startswith("flat_"):
                self.lgkm = True
        elif mn.startswith(("global_store", "buffer_store", "flat_store", "scratch_store", "tbuffer_store")):
            self.vm = True
        elif mn.startswith(("global_atomic", "buffer_atomic", "flat_atomic")):
            self.vm = True
            first_is_dst = any(t in ("sc0", "glc") for o in self.ops for t in o.split())
        elif mn.startswith("ds_"):
            self.lgkm = True
            first_is_dst = (mn.startswith(("ds_read", "ds_bpermute", "ds_permute", "ds_swizzle", "ds_consume",
                                           "ds_append", "ds_load")) or "_rtn" in mn)
        elif mn.startswith(("s_load", "s_buffer_load", "s_scratch_load")):
            self.lgkm = True
        defs, uses = [], []
        for i, o in enumerate(self.ops):
            rs = _regs(o)
            if i == 0 and first_is_dst:
                defs += rs
                if mn.startswith("v_swap"):
                    uses += rs
            else:
                uses += rs
        if mn.startswith("v_swap") and len(self.ops) > 1:
            defs += _regs(self.ops[1])
        self.defs, self.uses = defs, uses
        if self.mfma:
            passes, f32 = _mfma_passes(mn)
            self.need = (_NEED_F32 if f32 else _NEED_XDL)[passes]
            # operands: D, A, B, C [, modifiers]
            self.srcc = tuple(_regs(self.ops[3])) if len(self.ops) > 3 else ()


def parse_waitcnt(ins):
    """(vmcnt, lgkmcnt) limits of an s_waitcnt (None = not waited on)."""
    vm = lg = None
    txt = " ".join(ins.ops)
    m = re.search(r"vmcnt\((\d+)\)", txt)
    if m:
        vm = int(m.group(1))
    m = re.search(r"lgkmcnt\((\d+)\)", txt)
    if m:
        lg = int(m.group(1))
    if vm is None and lg is None and "cnt" not in txt and txt:
        imm = int(txt.split()[0], 0)
        vm = (imm & 0xF) | ((imm >> 14) & 0x3) << 4
        lg = (imm >> 8) & 0xF
        vm = None if vm == 63 else vm
        lg = None if lg == 15 else lg
    return vm, lg


def disassemble(so_path, workdir):
    """{kernel name: [Ins]} with `labels` {name: {label: index}} for every code object in the library."""
    local = os.path.join(workdir, "lib.so")
    with open(so_path, "rb") as f, open(local, "wb") as g:
        g.write(f.read())
    subprocess.run([OBJDUMP, "--offloading", local], cwd=workdir, capture_output=True, check=True)
    funcs, labels = {}, {}
    for name in sorted(os.listdir(workdir)):
        if "amdgcn" not in name:
            continue
        out = subprocess.run([OBJDUMP, "-d", "--no-show-raw-insn", "--symbolize-operands",
                              os.path.join(workdir, name)], capture_output=True, text=True, check=True).stdout
        cur = None
        for line in out.splitlines():
            if not line:
                continue
            if line[0] != "\t" and line[0] != " ":
                m = _FUNC.match(line)
                if m and re.fullmatch(r"L\d+", m.group(1)) and cur is not None:
                    labels[cur][m.group(1)] = len(funcs[cur])       # `addr <L12>:` = a branch target
                elif m:
                    cur = m.group(1)
                    funcs[cur], labels[cur] = [], {}
                continue
            if cur is None:
                continue
            body = line.strip()
            addr = ""
            if "//" in body:
                body, tail = body.split("//", 1)
                addr = tail.strip().split(":")[0]
                body = body.strip()
            if not body:
                continue
            m = _LABEL.match(body)
            if m:
                labels[cur][m.group(1)] = len(funcs[cur])
                continue
            funcs[cur].append(Ins(len(funcs[cur]), body, addr))
    return funcs, labels


def parse_listing(text):
    """(code, labels) from a hand-written listing: one instruction per line, `L3:` lines are labels
    (what the self-tests of tests/test_isa_lint.py feed the checks)."""
    code, labels = [], {}
    for line in text.strip().splitlines():
        line = line.strip()
        if not line:
            continue
        if re.fullmatch(r"L\d+:", line):
            labels[line[:-1]] = len(code)
            continue
        code.append(Ins(len(code), line, f"line{len(code)}"))
    return code, labels


def _succ(ins, i, labels, n):
    """Successor indices of instruction i."""
    out = []
    if ins.mn in ("s_endpgm", "s_setpc_b64", "s_swappc_b64"):
        return out
    if ins.target is not None and ins.target in labels:
        out.append(labels[ins.target])
    if ins.mn != "s_branch" and i + 1 < n:
        out.append(i + 1)
    return out


def check_hazards(name, code, labels):
    """R1 and R2 along every path while a window is open."""
    findings = []
    n = len(code)
    seen = set()

    def walk(start, opened, budget, kind):
        # follow straight-line code and branch edges until `budget` wait states have passed
        stack = [(start, budget)]
        while stack:
            i, left = stack.pop()
            while left > 0 and i < n:
                ins = code[i]
                if kind == "valu":
                    if ins.mfma and any(r in opened.regs for r in ins.uses):
                        findings.append(f"R1 {name}: `{opened.text}` @{opened.addr} writes a source of "
                                        f"`{ins.text}` @{ins.addr} {budget - left} wait states earlier (needs 2)")
                        return
                    if any(r in opened.regs for r in ins.defs):
                        break              # overwritten: this writer is no longer the producer
                else:
                    touched = [r for r in ins.uses + ins.defs if r in opened.regs]
                    if touched:
                        ok = False
                        if ins.mfma:
                            c = set(ins.srcc)
                            ab = [r for o in ins.ops[1:3] for r in _regs(o)]
                            ok = (c == opened.regs or not (c & opened.regs)) and not any(r in opened.regs for r in ab)
                            if ok and c != opened.regs and any(r in opened.regs for r in ins.defs):
                                ok = set(ins.defs) == opened.regs and not c & opened.regs
                        if not ok:
                            findings.append(f"R2 {name}: `{ins.text}` @{ins.addr} touches the result of "
                                            f"`{opened.text}` @{opened.addr} after {budget - left} wait states "
                                            f"(needs {budget})")
                            return
                        if ins.mfma and set(ins.defs) == opened.regs:
                            break          # the accumulate chain continues: the later MFMA opens its own window
                left -= ins.ws
                if left <= 0:
                    break
                if not _succ(ins, i, labels, n):
                    break
                if ins.target is not None and ins.target in labels:
                    key = (opened.idx, labels[ins.target], left)
                    if key not in seen:
                        seen.add(key)
                        stack.append((labels[ins.target], left))
                    if ins.mn == "s_branch":
                        break
                i += 1

    class Open:
        __slots__ = ("regs", "text", "addr", "idx")

    for i, ins in enumerate(code):
        if ins.is_valu and ins.defs:
            o = Open()
            o.regs, o.text, o.addr, o.idx = set(ins.defs), ins.text, ins.addr, i
            for j in _succ(ins, i, labels, n):
                walk(j, o, 2, "valu")
        elif ins.mfma:
            o = Open()
            o.regs, o.text, o.addr, o.idx = set(ins.defs), ins.text, ins.addr, i
            for j in _succ(ins, i, labels, n):
                walk(j, o, ins.need, "mfma")
    return findings


_SREG = re.compile(r"\bs(?:(\d+)|\[(\d+):(\d+)\])")


def _sregs(text):
    out = set()
    for m in _SREG.finditer(text):
        if m.group(1) is not None:
            out.add(int(m.group(1)))
        else:
            out.update(range(int(m.group(2)), int(m.group(3)) + 1))
    return out


_SCC_KEEP = ("s_mov_", "s_cmov", "s_nop", "s_waitcnt", "s_barrier", "s_branch", "s_cbranch", "s_sleep", "s_setprio",
             "s_load", "s_buffer_load", "s_store", "s_setreg", "s_getreg", "s_endpgm", "s_sendmsg", "s_movk",
             "s_cselect", "s_mul_i32", "s_mul_hi", "s_bitset", "s_sext", "s_getpc", "s_dcache", "s_icache", "s_pack")


def check_waitcnt(name, code, labels):
    """R3 as a forward data-flow problem over basic blocks.  State: for every load still outstanding
    (by instruction index) the number of YOUNGER operations of its counter issued since -- vmcnt(N) /
    lgkmcnt(N) retire exactly the entries with at least N younger ones (both counters return in
    order; with a scalar load pending only lgkmcnt(0) is trusted).  At a join the states are united,
    an entry keeping its smallest count (the path on which it is hardest to retire).
    Branch correlation: macro-unrolled loop bodies re-test one scalar compare several times
    (`s_cmp X; s_cbranch_scc0 ..; s_cmp X; s_cbranch_scc1 ..`); a node of the analysis is therefore a
    (block, known compare outcome) pair -- the text of the last s_cmp whose outcome the taken edge
    fixed, valid until one of its registers is rewritten -- and a branch on a compare whose outcome
    is known has one successor.  Other correlations are not modelled: a report is a path of the
    control-flow graph, not necessarily of the program."""
    n = len(code)
    if n == 0:
        return []
    leaders = {0} | set(labels.values())
    for i, ins in enumerate(code):
        if ins.target is not None or ins.mn in ("s_endpgm", "s_setpc_b64", "s_swappc_b64"):
            leaders.add(i + 1)
    leaders = sorted(x for x in leaders if x < n)
    block_of = {st: bi for bi, st in enumerate(leaders)}
    ends = leaders[1:] + [n]
    found = {}

    def transfer(bi, vm, lg, fact, report):
        """-> (vm, lg, [(successor block, fact)])"""
        vm, lg = dict(vm), dict(lg)
        scc = None                      # (cmp text, outcome or None) of the compare SCC currently holds
        for i in range(leaders[bi], ends[bi]):
            ins = code[i]
            mn = ins.mn
            if mn == "s_waitcnt":
                v, l = parse_waitcnt(ins)
                if v is not None:
                    vm = {e: c for e, c in vm.items() if c < v}
                if l is not None:
                    if l == 0:
                        lg = {}
                    elif not any(code[e].mn.startswith("s_") for e in lg):
                        lg = {e: c for e, c in lg.items() if c < l}
                continue
            if mn.startswith("s_cmp"):
                scc = (ins.text, fact[1] if fact is not None and fact[0] == ins.text else None)
            elif mn.startswith("s_") and not mn.startswith(_SCC_KEEP):
                scc = None              # another SALU instruction rewrote SCC
            if fact is not None and (mn.startswith("s_") and not mn.startswith(("s_cmp", "s_cbranch", "s_branch",
                                                                                  "s_nop", "s_waitcnt", "s_barrier"))
                                     or mn.startswith("v_readfirstlane") or mn.startswith("v_readlane")
                                     or mn.startswith("v_cmp")):
                if ins.ops and _sregs(ins.ops[0]) & fact[2] or "vcc" in ins.ops[0:1] and False:
                    fact = None
            if report and (vm or lg):
                touched = ins.uses + ins.defs
                if touched:
                    for q in (vm, lg):
                        for e in q:
                            p = code[e]
                            if not p.defs or p.mn.startswith("s_") or not any(r in p.defs for r in touched):
                                continue
                            # a later load into the same registers is ordered behind the earlier one
                            if (ins.vm and p.vm or ins.lgkm and p.lgkm) and not any(r in p.defs for r in ins.uses):
                                continue
                            found.setdefault((i, e), f"R3 {name}: `{ins.text}` @{ins.addr} touches the destination "
                                                     f"of `{p.text}` @{p.addr} before an s_waitcnt retired it")
            # (counts saturate at 64: s_waitcnt cannot name more, and a bounded count bounds the iteration)
            if ins.vm:
                vm = {e: min(c + 1, 64) for e, c in vm.items()}
                vm[i] = 0
            if ins.lgkm:
                lg = {e: min(c + 1, 64) for e, c in lg.items()}
                lg[i] = 0
        last = code[ends[bi] - 1]
        out = []
        if last.mn in ("s_endpgm", "s_setpc_b64", "s_swappc_b64"):
            return vm, lg, out
        tgt = block_of[labels[last.target]] if last.target is not None and last.target in labels \
            and labels[last.target] < n else None
        fall = block_of[ends[bi]] if last.mn != "s_branch" and ends[bi] < n else None
        if last.mn in ("s_cbranch_scc0", "s_cbranch_scc1") and scc is not None:
            taken_if = last.mn.endswith("1")
            regs = frozenset(_sregs(scc[0]))
            if scc[1] is not None:      # outcome known: one successor
                if scc[1] == taken_if:
                    fall = None
                else:
                    tgt = None
                f_t = f_f = (scc[0], scc[1], regs)
            else:
                f_t, f_f = (scc[0], taken_if, regs), (scc[0], not taken_if, regs)
            if tgt is not None:
                out.append((tgt, f_t))
            if fall is not None:
                out.append((fall, f_f))
            return vm, lg, out
        if tgt is not None:
            out.append((tgt, fact))
        if fall is not None:
            out.append((fall, fact))
        return vm, lg, out

    state = {(0, None): ({}, {})}
    work = [(0, None)]
    queued = {(0, None)}
    rounds = 0
    while work:
        node = work.pop()
        queued.discard(node)
        rounds += 1
        if rounds > 4000 * len(leaders) + 2000:
            return [f"R3 {name}: the data-flow iteration did not converge"]
        vm, lg, out = transfer(node[0], state[node][0], state[node][1], node[1], False)
        for sj, f in out:
            key = (sj, f)
            changed = False
            if key not in state:
                state[key] = (dict(vm), dict(lg))
                changed = True
            else:
                for src, dst in ((vm, state[key][0]), (lg, state[key][1])):
                    for e, c in src.items():
                        if e not in dst or c < dst[e]:
                            dst[e] = c
                            changed = True
            if changed and key not in queued:
                queued.add(key)
                work.append(key)
    for node, (vm, lg) in state.items():
        transfer(node[0], vm, lg, node[1], True)
    return [found[k] for k in sorted(found)][:5]


def _is_dma(ins):
    return ins.vm and ("_lds_" in ins.mn or bool(ins.ops and ins.ops[-1].split()[-1] == "lds"))


def _asm_form_dma(ins):
    """The LDS-DMA form only the inline asm of this library issues (csrc/ring_common.h dma16_saddr):
    scalar base + 32-bit lane offset, `global_load_lds_dwordx4 v1, s[2:3]`.  The builtins
    (__builtin_amdgcn_global_load_lds: 64-bit vector address, `v[2:3], off`;
    __builtin_amdgcn_raw_ptr_buffer_load_lds: `buffer_load_dwordx4 ... offen lds`) leave M0 to hipcc."""
    return _is_dma(ins) and ins.mn.startswith("global_") and any(o.startswith("s[") for o in ins.ops[1:])


def check_m0(name, code):
    """R4."""
    findings = []
    dma = [i for i, ins in enumerate(code) if _is_dma(ins)]
    hand = [i for i in dma if _asm_form_dma(code[i])]
    if not hand:
        return findings
    if len(hand) != len(dma):
        bad = [code[i] for i in dma if i not in hand][0]
        findings.append(f"R4 {name}: {len(hand)} LDS-DMAs with a hand-set M0 share the kernel with the "
                        f"compiler-managed `{bad.text}` @{bad.addr}")
    for i in hand:
        ok = False
        for j in range(i - 1, max(-1, i - 4), -1):
            p = code[j]
            if p.mn in ("s_mov_b32", "s_add_u32", "s_add_i32") and p.ops and p.ops[0] == "m0":
                ok = True
                break
            if p.mn != "s_nop":
                break
        if not ok:
            findings.append(f"R4 {name}: `{code[i].text}` @{code[i].addr} has no M0 write of its own in front of it")
            break
    return findings


def _lint_one(job):
    name, code, labels = job
    return check_hazards(name, code, labels) + check_waitcnt(name, code, labels) + check_m0(name, code)


def lint(so_path, only=None, workers=None):
    with tempfile.TemporaryDirectory() as wd:
        funcs, labels = disassemble(so_path, wd)
    findings, stats = [], {"kernels": 0, "instructions": 0, "mfma": 0, "lds_dma": 0, "asm_style_loads": 0}
    jobs = []
    for name, code in funcs.items():
        if only and only not in name:
            continue
        stats["kernels"] += 1
        stats["instructions"] += len(code)
        stats["mfma"] += sum(1 for c in code if c.mfma)
        stats["lds_dma"] += sum(1 for c in code if _is_dma(c))
        stats["asm_style_loads"] += sum(1 for c in code if _asm_form_dma(c))
        jobs.append((name, code, labels[name]))
    # the kernels are independent and the walk is pure Python: one process per core, largest kernels first
    jobs.sort(key=lambda j: -len(j[1]))
    workers = workers or min(len(jobs), os.cpu_count() or 1, 8)
    if workers > 1 and len(jobs) > 1:
        import multiprocessing as mp
        with mp.get_context("fork").Pool(workers) as pool:
            for f in pool.imap_unordered(_lint_one, jobs):
                findings += f
    else:
        for j in jobs:
            findings += _lint_one(j)
    return sorted(findings), stats


if __name__ == "__main__":
    here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    so = sys.argv[1] if len(sys.argv) > 1 else os.path.join(
        here, "quantized-autoregression-image-generator_amd", "lib", "libqarig_hip.so")
    f, s = lint(so, sys.argv[2] if len(sys.argv) > 2 else None)
    for line in f[:200]:
        print(line)
    print(s, "findings:", len(f))
