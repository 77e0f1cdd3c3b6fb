#!/usr/bin/env python
"""Static guard for the hand-scheduled GEMM kernels: checks, on the gfx950 assembly the compiler emits (hipcc -S
--cuda-device-only), the two properties their hand-counted `s_waitcnt vmcnt(N)` rely on and that the register / scratch audit
of __graft_entry__.py cannot see:

  1. between a `global_load_dword*` and the `s_waitcnt` that retires it, NO instruction reads or overwrites its destination
     registers -- the class of the intermittent fault of round 2 (a load into a register nobody read was a dead register to the
     compiler, which reused it while the load was in flight), and of a compiler-inserted `v_mov` of an in-flight destination;
  2. inside the K loops (from the first to the last MFMA of the kernel) every vector-memory instruction comes from one of the
     kernel's own asm statements (`;;#ASMSTART` ... `;;#ASMEND`): a compiler-generated load, store or spill there would shift
     every count behind it.  (A few kernels are allowed a small, named number of compiler scratch operations: an extra
     operation is YOUNGER than the loads a counted wait protects, so it can only make that wait stricter; it is reported.)

The check simulates the in-order vmcnt queue along every control-flow path from the kernel's entry to its last MFMA: loads, stores
and LDS-DMA enter the queue in program order (they retire in issue order under the one counter: tools/vmcnt_order_test.hip), a
wait with vmcnt(N) pops all but the N youngest; both arms of every branch are followed and loops are walked until the queue
contents repeat, so operations in flight across a back edge are seen by the next iteration.  Usage: asm_guard.py file.s [kernel-name-substring ...]"""
import re
import sys

_VREG = re.compile(r"\bv\[(\d+):(\d+)\]|\bv(\d+)\b")
_VMCNT = re.compile(r"vmcnt\((\d+)\)")
_LABEL = re.compile(r"^(\.LBB[0-9_]+):")


def _vregs(text):
    out = set()
    for m in _VREG.finditer(text):
        if m.group(1) is not None:
            out.update(range(int(m.group(1)), int(m.group(2)) + 1))
        else:
            out.add(int(m.group(3)))
    return out


def split_kernels(asm_text):
    """-> {mangled name: [lines]} for every .globl function that ends in s_endpgm"""
    kernels, cur, name = {}, None, None
    for line in asm_text.splitlines():
        m = re.match(r"^(_Z\w+):", line)
        if m:
            name, cur = m.group(1), []
            continue
        if cur is not None:
            cur.append(line)
            if line.strip().startswith("s_endpgm"):
                kernels[name] = cur
                cur, name = None, None
    return kernels


def _parse(lines):
    """instruction list: dicts with op, text, in_asm, label (for label lines op is None)"""
    out, in_asm = [], False
    for raw in lines:
        s = raw.strip()
        if not s or s.startswith(";") and "ASMSTART" not in s and "ASMEND" not in s:
            continue
        if "ASMSTART" in s:
            in_asm = True
            continue
        if "ASMEND" in s:
            in_asm = False
            continue
        m = _LABEL.match(s)
        if m:
            out.append({"op": None, "label": m.group(1)})
            continue
        if s.startswith("."):
            continue
        code = s.split(";")[0].strip()
        if not code:
            continue
        op = code.split()[0]
        out.append({"op": op, "text": code, "in_asm": in_asm})
    return out


def _is_vmem(op):
    return (op.startswith("global_load") or op.startswith("global_store") or op.startswith("global_atomic") or
            op.startswith("scratch_") or op.startswith("buffer_load") or op.startswith("buffer_store") or
            op.startswith("flat_load") or op.startswith("flat_store"))


def check_kernel(lines, allowed_compiler_vmem=0, max_states=400000):
    """-> (violations [str], stats dict).  Explores the control-flow graph of the instructions up to the kernel's last MFMA: a state
    is (instruction index, queue of outstanding vector-memory operations); both successors of a conditional branch are followed and
    a state seen before is not expanded again, so loops are walked until the queue contents repeat."""
    ins = _parse(lines)
    mf = [i for i, x in enumerate(ins) if x["op"] and x["op"].startswith("v_mfma")]
    if not mf:
        return [], {"mfma": 0}
    first, last = mf[0], mf[-1]
    labels = {x["label"]: i for i, x in enumerate(ins) if x["op"] is None}
    stats = {"mfma": len(mf), "asm_loads": 0, "asm_lds_dma": 0, "waits": 0, "compiler_vmem_in_loop": 0, "max_in_flight": 0, "states": 0}
    viol = {}
    dests, uses = {}, {}
    for i, x in enumerate(ins[:last + 1]):
        if x["op"] is None:
            continue
        uses[i] = frozenset(_vregs(x["text"]))
        if _is_vmem(x["op"]):
            d = set()
            if x["op"].startswith(("global_load", "scratch_load", "buffer_load", "flat_load")) and "_lds_" not in x["op"]:
                d = _vregs(x["text"].split(",")[0])
            dests[i] = frozenset(d)
        elif x["op"] == "s_waitcnt" and _VMCNT.search(x["text"]):
            stats["waits"] += 1
    seen = set()
    work = [(0, ())]
    QMAX = 64                          # vmcnt is a 6-bit counter: more than 63 operations are never in flight

    def busy_of(queue):
        b = set()
        for q in queue:
            b |= dests[q]
        return b

    while work:
        i, queue = work.pop()
        busy = busy_of(queue)
        while i <= last:
            x = ins[i]
            op = x["op"]
            if op is None or op[0] == "s" and (op.startswith("s_cbranch") or op == "s_branch"):
                key = (i, queue)
                if key in seen:
                    break
                seen.add(key)
                stats["states"] += 1
                if stats["states"] > max_states:
                    viol["analysis"] = "state space larger than %d: analysis incomplete" % max_states
                    work = []
                    break
                if op is None:
                    i += 1
                    continue
                tgt = labels.get(x["text"].split()[-1])
                if op == "s_branch":
                    if tgt is None:
                        break
                    i = tgt
                    continue
                if tgt is not None:
                    work.append((tgt, queue))
                i += 1
                continue
            if op == "s_endpgm":
                break
            if op == "s_waitcnt":
                m = _VMCNT.search(x["text"])
                if m:
                    n = int(m.group(1))
                    if len(queue) > n:
                        queue = queue[len(queue) - n:] if n else ()
                        busy = busy_of(queue)
                i += 1
                continue
            if busy:
                hit = uses[i] & busy
                if hit and i not in viol:
                    viol[i] = "`%s` touches v%s while `%s` is in flight" % (x["text"], sorted(hit)[:4],
                                                                             [ins[q]["text"] for q in queue if dests[q] & hit][0])
            if i in dests:
                queue = (queue + (i,))[-QMAX:]
                busy = busy_of(queue)
                if len(queue) > stats["max_in_flight"]:
                    stats["max_in_flight"] = len(queue)
            i += 1
    out = [viol[k] for k in sorted(viol, key=str)]
    # property 2: vector-memory instructions between the first and the last MFMA come from asm statements
    for x in ins[first:last + 1]:
        if x["op"] and _is_vmem(x["op"]):
            if x["in_asm"]:
                if "_lds_" in x["op"]:
                    stats["asm_lds_dma"] += 1
                else:
                    stats["asm_loads"] += 1
            else:
                stats["compiler_vmem_in_loop"] += 1
                if stats["compiler_vmem_in_loop"] > allowed_compiler_vmem:
                    out.append("compiler-generated vector-memory instruction inside the K loop: `%s`" % x["text"])
    return out, stats


def check_kernel_local(lines, allowed_compiler_vmem=0):
    """The block-local form of property 1 for kernels whose unrolled K steps are guarded by trip-count branches (the CFG walk above
    follows infeasible paths through them: `step s+2 skipped, loop taken again`): from each asm load onwards, in layout order up
    to the next `s_waitcnt vmcnt`, branch or label, no instruction may touch its destination registers.  This is the window in
    which the round-2 fault happened (a destination nobody read was handed out right behind the load).  Property 2 as above."""
    ins = _parse(lines)
    mf = [i for i, x in enumerate(ins) if x["op"] and x["op"].startswith("v_mfma")]
    if not mf:
        return [], {"mfma": 0}
    first, last = mf[0], mf[-1]
    stats = {"mfma": len(mf), "asm_loads": 0, "asm_lds_dma": 0, "waits": 0, "compiler_vmem_in_loop": 0, "max_in_flight": 0, "states": 0}
    out = []
    for i, x in enumerate(ins[:last + 1]):
        if not x["op"] or not x["in_asm"] or not x["op"].startswith("global_load") or "_lds_" in x["op"]:
            continue
        dest = _vregs(x["text"].split(",")[0])
        for y in ins[i + 1:last + 1]:
            if y["op"] is None or y["op"].startswith(("s_cbranch", "s_branch", "s_endpgm")):
                break
            if y["op"] == "s_waitcnt" and _VMCNT.search(y["text"]):
                break
            hit = _vregs(y["text"]) & dest
            if hit:
                out.append("`%s` touches v%s right behind `%s`" % (y["text"], sorted(hit)[:4], x["text"]))
                break
    for x in ins[:last + 1]:
        if x["op"] == "s_waitcnt" and _VMCNT.search(x["text"]):
            stats["waits"] += 1
    for x in ins[first:last + 1]:
        if x["op"] and _is_vmem(x["op"]):
            if x["in_asm"]:
                stats["asm_lds_dma" if "_lds_" in x["op"] else "asm_loads"] += 1
            else:
                stats["compiler_vmem_in_loop"] += 1
                if stats["compiler_vmem_in_loop"] > allowed_compiler_vmem:
                    out.append("compiler-generated vector-memory instruction inside the K loop: `%s`" % x["text"])
    return out, stats


def check_file(path, want=(), allowed=None, local=()):
    """Check every kernel of `path` whose mangled name contains one of `want` (all when empty).  `allowed`: {substring: n} compiler
    scratch operations tolerated inside the MFMA region; `local`: name substrings of the kernels checked in the block-local form.
    -> (report lines, violations)"""
    kernels = split_kernels(open(path).read())
    rows, bad = [], []
    for name, lines in sorted(kernels.items()):
        if want and not any(w in name for w in want):
            continue
        n_ok = 0
        for sub, n in (allowed or {}).items():
            if sub in name:
                n_ok = n
        is_local = any(w in name for w in local)
        viol, st = (check_kernel_local if is_local else check_kernel)(lines, n_ok)
        if not st.get("mfma"):
            continue
        rows.append("%s %s mfma %d asm_loads %d asm_lds_dma %d waits %d max_in_flight %d states %d compiler_vmem_in_loop %d violations %d"
                    % (name, "block-local" if is_local else "all-paths", st["mfma"], st["asm_loads"], st["asm_lds_dma"], st["waits"], st["max_in_flight"], st["states"], st["compiler_vmem_in_loop"], len(viol)))
        bad += ["%s: %s" % (name, v) for v in viol]
    return rows, bad


if __name__ == "__main__":
    rows, bad = check_file(sys.argv[1], [a for a in sys.argv[2:] if not a.startswith("local=")],
                           local=[a[6:] for a in sys.argv[2:] if a.startswith("local=")])
    print("\n".join(rows))
    if bad:
        print("\n".join(bad[:40]))
        sys.exit(1)
